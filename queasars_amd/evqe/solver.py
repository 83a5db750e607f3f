"""EVQE driver over a circuit evaluator.

Restates the control flow of the reference's solver so that the evaluator can be exercised end to end
(BASELINE config 4), without Qiskit, Dask or a thread pool:

* operator order and seeding chain: queasars/minimum_eigensolvers/evqe/evqe.py:188-229;
* generation loop, callbacks, termination: queasars/minimum_eigensolvers/base/evolving_ansatz_minimum_eigensolver.py:331-433;
* mutation: queasars/minimum_eigensolvers/evqe/evolutionary_algorithm/mutation.py:28-132 (layer optimisation),
  :194-235 (who is mutated, with which seed), :328-334 / :347-353 / :383-395 (the four mutation operators);
* speciation: .../speciation.py:34-90;  selection: .../selection.py:64-175;
* termination criterion: queasars/minimum_eigensolvers/base/termination_criteria.py:90-144;
* SPSA stopping rule: queasars/utility/spsa_termination.py:48-96.

The optimiser itself lives in qiskit-algorithms (absent here); :class:`SPSA` restates its published first-order
algorithm for the settings the reference's examples use (constant learning rate and perturbation, Bernoulli +-1
directions from ``numpy.random.default_rng(seed)``, one resampling, optional trust region, ``last_avg=1``).

MI355X-first difference: the reference farms one optimiser run per individual to a thread pool and lets a
"batching mutex" glue their two-circuit requests together inside a 0.1 s window
(queasars/circuit_evaluation/mutex_primitives.py:67-199).  Here all runs advance in lock-step and every SPSA
iteration of the whole population is ONE ``evaluate_circuits`` call.  Each run still sees exactly the sequence of
function values an independent run with its seed would see.
"""

from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from random import Random
from statistics import mean, median
from typing import Callable, Optional, Sequence

import numpy as np

from queasars_amd.evqe.genome import EVQEIndividual, EVQEPopulation, new_random_seed


# ---- optimiser ------------------------------------------------------------------------------------------------


class SPSATerminationChecker:
    """Stop when |f_k - f_{k-1}| / f_{k-1} stayed below ``minimum_relative_change`` for
    ``allowed_consecutive_violations + 1`` consecutive accepted iterations, or when ``maxfev`` evaluations are reached
    (reference: queasars/utility/spsa_termination.py:9-134; held to sequences the reference's class itself answered,
    tests/golden/spsa_termination_reference.json).  One object may serve one optimisation after another: a call after it has
    said "stop", or with fewer evaluations than the call before, starts a new history."""

    def __init__(self, minimum_relative_change: float, allowed_consecutive_violations: int, maxfev: Optional[int] = None):
        self.minimum_relative_change = minimum_relative_change
        self.allowed_consecutive_violations = allowed_consecutive_violations
        self.maxfev = maxfev
        self._start_over()

    def _start_over(self) -> None:
        self.function_value_history: list[float] = []
        self.n_function_evaluation_history: list[int] = []
        self._changes: list[float] = []
        self.n_function_evaluations = 0
        self.best_function_value = float("inf")
        self._best_parameter_values = None
        self._done = False

    @property
    def best_parameter_values(self):
        if self._best_parameter_values is None:
            raise ValueError("The termination checker has stored no parameter values (no accepted step since its history began)!")
        return self._best_parameter_values

    def fresh(self) -> "SPSATerminationChecker":
        return SPSATerminationChecker(self.minimum_relative_change, self.allowed_consecutive_violations, self.maxfev)

    def termination_check(self, n_function_evaluations: int, parameter_values, function_value: float, step_size: float, accepted: bool) -> bool:
        if self._done or n_function_evaluations < self.n_function_evaluations:
            self._start_over()
        self.n_function_evaluations = n_function_evaluations
        if self.maxfev is not None and n_function_evaluations >= self.maxfev:
            return True
        if not accepted:
            return False
        self.function_value_history.append(function_value)
        self.n_function_evaluation_history.append(n_function_evaluations)
        if function_value < self.best_function_value:
            self.best_function_value = function_value
            self._best_parameter_values = parameter_values
        if len(self.function_value_history) < 2:
            return False
        previous = self.function_value_history[-2]
        self._changes.append(abs(function_value - previous) / previous)
        window = self.allowed_consecutive_violations + 1
        if len(self._changes) < window:
            return False
        if max(self._changes[-window:]) < self.minimum_relative_change:
            self._done = True
            return True
        return False


@dataclass
class SPSA:
    """First-order SPSA with constant gains (the configuration of examples/evqe_jssp_optimization.ipynb)."""

    maxiter: int = 33
    learning_rate: float = 0.43
    perturbation: float = 0.35
    trust_region: bool = True
    termination_checker: Optional[SPSATerminationChecker] = None

    @property
    def n_circuit_evaluations(self) -> int:
        return 2 * self.maxiter

    def new_run(self, x0: Sequence[float], seed: Optional[int]) -> "_SPSARun":
        return _SPSARun(self, x0, seed)


class _SPSARun:
    """State of one SPSA minimisation; the driver advances many of them in lock-step."""

    def __init__(self, config: SPSA, x0: Sequence[float], seed: Optional[int]):
        self.config = config
        self.x = np.asarray(x0, dtype=np.float64).copy()
        self.rng = np.random.default_rng(seed)
        self.checker = config.termination_checker.fresh() if config.termination_checker else None
        self.nfev = 0
        self.iteration = 0
        self.done = self.x.size == 0 or config.maxiter <= 0
        self._delta = None
        # (base vector, positions): the run's variables are entries `positions` of a longer parameter vector whose other
        # entries stay at `base` -- a layer of an individual inside the individual's fully parameterised circuit
        self.embed = None

    def propose(self) -> list[np.ndarray]:
        """The two points the next iteration evaluates."""
        self._delta = 1 - 2 * self.rng.binomial(1, 0.5, size=self.x.size)
        eps = self.config.perturbation
        return [self.x + eps * self._delta, self.x - eps * self._delta]

    def accept(self, f_plus: float, f_minus: float) -> None:
        cfg = self.config
        self.nfev += 2
        self.iteration += 1
        update = (f_plus - f_minus) / (2 * cfg.perturbation) * self._delta
        if cfg.trust_region:
            norm = float(np.linalg.norm(update))
            if norm > 1:
                update = update / norm
        update = update * cfg.learning_rate
        self.x = self.x - update
        stop = self.iteration >= cfg.maxiter
        if self.checker is not None and self.checker.termination_check(
            self.nfev, self.x, 0.5 * (f_plus + f_minus), float(np.linalg.norm(update)), True
        ):
            stop = True
        self.done = stop


@dataclass
class NFT:
    """Nakanishi-Fujii-Todo sequential minimisation (Phys. Rev. Research 2, 043158): with every other angle fixed, the
    expectation value is a sinusoid of period 2 pi in one gate angle, f(t) = c + a cos(t - b), so three values --
    f(t0), f(t0 + pi/2), f(t0 - pi/2) -- determine it and the parameter jumps to the exact minimiser t = b + pi.
    Parameters are visited cyclically; the fitted minimum c - a is reused as the next f(t0) and re-evaluated every
    ``reset_interval`` iterations.  This is the optimiser of the reference's own test harness
    (test/minimum_eigensolvers/evqe/solver.py:28-36: ``NFT(maxfev=40)``); restated from the paper, the reference takes
    it from qiskit-algorithms."""

    maxfev: int = 40
    maxiter: Optional[int] = None
    reset_interval: int = 32

    @property
    def n_circuit_evaluations(self) -> int:
        return self.maxfev

    def new_run(self, x0: Sequence[float], seed: Optional[int]) -> "_NFTRun":
        return _NFTRun(self, x0)


class _NFTRun:
    """State of one NFT minimisation, advanced in lock-step with the others like :class:`_SPSARun`."""

    def __init__(self, config: NFT, x0: Sequence[float]):
        self.config = config
        self.x = np.asarray(x0, dtype=np.float64).copy()
        self.nfev = 0
        self.iteration = 0
        self.done = self.x.size == 0 or config.maxfev <= 0
        self._recycled: Optional[float] = None
        self._needs_base = True
        self.embed = None  # (as _SPSARun.embed)

    def propose(self) -> list[np.ndarray]:
        cfg = self.config
        j = self.iteration % self.x.size
        if cfg.reset_interval > 0 and self.iteration % cfg.reset_interval == 0:
            self._recycled = None
        self._needs_base = self._recycled is None
        plus, minus = self.x.copy(), self.x.copy()
        plus[j] += 0.5 * math.pi
        minus[j] -= 0.5 * math.pi
        return ([self.x.copy()] if self._needs_base else []) + [plus, minus]

    def accept(self, *values: float) -> None:
        cfg = self.config
        self.nfev += len(values)
        z0 = values[0] if self._needs_base else self._recycled
        z1, z3 = values[-2], values[-1]
        j = self.iteration % self.x.size
        # f(t) = c + a cos(t - b):  z0 - c = a cos(t0 - b),  (z3 - z1) / 2 = a sin(t0 - b)
        c = 0.5 * (z1 + z3)
        cos_part, sin_part = z0 - c, 0.5 * (z3 - z1)
        a = math.hypot(cos_part, sin_part)
        if a > 0.0:
            self.x[j] = self.x[j] - math.atan2(sin_part, cos_part) + math.pi
        self._recycled = c - a
        self.iteration += 1
        self.done = self.nfev >= cfg.maxfev or (cfg.maxiter is not None and self.iteration >= cfg.maxiter)


def _full_point(run, point: np.ndarray) -> np.ndarray:
    """The parameter vector the evaluator gets for a run's point (run.embed)."""
    if run.embed is None:
        return point
    base, positions = run.embed
    full = base.copy()
    full[positions] = point
    return full


def _minimize_spsa_vectorised(evaluator, jobs: list) -> None:
    """:func:`_minimize_batched` for SPSA runs of one configuration, with the arithmetic of all runs in whole-array NumPy
    operations: as written per run (propose / accept of :class:`_SPSARun`: a binomial draw, two array expressions and two
    ``tolist`` per proposal, a dozen small NumPy calls per acceptance) an iteration of 64 runs costs 0.9 ms of Python around
    0.1 ms of GPU work.  Here every run's sign vectors are drawn once (the same draws, in the same order, as one call per
    iteration makes), the iterates live in one zero-padded matrix, and the evaluator receives rows of it (NumPy views: the
    packer copies them as buffers).  Every run still gets exactly the iterates its own ``propose`` / ``accept`` would have
    produced -- bit for bit: the same expressions element by element, the norm of an update taken over the run's own length
    (``tests/test_evqe_solver.py``)."""
    runs = [run for _, run in jobs]
    cfg = runs[0].config
    eps, lr = cfg.perturbation, cfg.learning_rate
    sizes = np.array([run.x.size for run in runs])
    width = int(sizes.max())
    n_runs = len(runs)
    x = np.zeros((n_runs, width))
    signs = np.zeros((n_runs, cfg.maxiter, width))
    for i, run in enumerate(runs):
        x[i, : sizes[i]] = run.x
        left = cfg.maxiter - run.iteration
        if left > 0 and sizes[i] > 0:  # (what propose() would draw call by call: one draw of the lot gives the same numbers)
            signs[i, run.iteration :, : sizes[i]] = 1 - 2 * run.rng.binomial(1, 0.5, size=(left, int(sizes[i])))
    iteration = np.array([run.iteration for run in runs])
    nfev = np.array([run.nfev for run in runs])
    any_checker = any(run.checker is not None for run in runs)
    active = np.array([i for i, run in enumerate(runs) if not run.done], dtype=np.int64)
    embedded = any(run.embed is not None for run in runs)
    if embedded:
        # (every run's variables are entries of a longer vector: the rows the evaluator gets are the base vectors with the
        # points scattered into them)
        full_sizes = np.array([run.embed[0].size if run.embed is not None else run.x.size for run in runs])
        full = np.zeros((n_runs, int(full_sizes.max())))
        for i, run in enumerate(runs):
            if run.embed is not None:
                full[i, : full_sizes[i]] = run.embed[0]
    circuits_of = {}
    while active.size:
        key = active.tobytes()
        cached = circuits_of.get(key)
        if cached is None:  # (the same list object while the same runs are active: the evaluator's caches key on it)
            scatter = None
            if embedded:
                rows, cols, src = [], [], []
                for a, i in enumerate(active):
                    positions = runs[i].embed[1] if runs[i].embed is not None else np.arange(sizes[i])
                    rows.append(np.full(positions.size, 2 * a))
                    cols.append(positions)
                    src.append(np.arange(positions.size))
                scatter = (np.concatenate(rows), np.concatenate(cols), np.concatenate(src))
            lengths = full_sizes if embedded else sizes
            cached = circuits_of[key] = ([jobs[i][0] for i in active for _ in (0, 1)],
                                         [slice(0, int(lengths[i])) for i in active for _ in (0, 1)], scatter)
        circuits, cuts, scatter = cached
        delta = signs[active, iteration[active]]
        points = np.empty((2 * active.size, width))
        points[0::2] = x[active] + eps * delta
        points[1::2] = x[active] - eps * delta
        if embedded:
            rows, cols, src = scatter
            sent = np.repeat(full[active], 2, axis=0)
            sent[rows, cols] = points[rows, src]
            sent[rows + 1, cols] = points[rows + 1, src]
        else:
            sent = points
        values = np.asarray(evaluator.evaluate_circuits(circuits, [row[cut] for row, cut in zip(sent, cuts)]), dtype=np.float64)
        f_plus, f_minus = values[0::2], values[1::2]
        update = ((f_plus - f_minus) / (2 * eps))[:, None] * delta
        if cfg.trust_region:
            # (the exact norm -- the run's own sum, in its own order -- only where the update can be longer than 1 at all)
            for a in np.nonzero(np.einsum("ij,ij->i", update, update) > 0.999)[0]:
                row = update[a, : sizes[active[a]]]
                norm = math.sqrt(row.dot(row))  # (= numpy.linalg.norm of a real vector)
                if norm > 1:
                    update[a] = update[a] / norm
        update = update * lr
        x[active] = x[active] - update
        iteration[active] += 1
        nfev[active] += 2
        stop = iteration[active] >= cfg.maxiter
        if any_checker:
            for a, i in enumerate(active):
                run = runs[i]
                if run.checker is not None:
                    row = update[a, : sizes[i]]
                    if run.checker.termination_check(int(nfev[i]), x[i, : sizes[i]].copy(), 0.5 * (f_plus[a] + f_minus[a]),
                                                     float(math.sqrt(row.dot(row))), True):
                        stop[a] = True
        for i in active[stop]:
            runs[i].done = True
        active = active[~stop]
    for i, run in enumerate(runs):
        run.x = x[i, : sizes[i]].copy()
        run.iteration = int(iteration[i])
        run.nfev = int(nfev[i])


def _device_search_wanted(evaluator, n_runs: int, flag: Optional[bool], optimizer) -> bool:
    """Will a search of ``n_runs`` fresh runs of ``optimizer`` keep its state on the device (_minimize_batched's rule)?"""
    env = os.environ.get("QSV_DEVICE_SEARCH")
    if env == "0" or os.environ.get("QSV_SCALAR_SPSA") or not isinstance(optimizer, SPSA) or n_runs < 2:
        return False
    if flag is None:
        flag = n_runs >= _DEVICE_SEARCH_MIN_RUNS
    if not (flag or env == "1"):
        return False
    if not hasattr(evaluator, "evaluate_device_to_device") or not evaluator.device_resident_search_possible():
        return False
    checker = optimizer.termination_checker
    return optimizer.maxiter > 0 and (checker is None or type(checker) is SPSATerminationChecker)


_DEVICE_SEARCH_MIN_RUNS = 16  # (config 4 on one MI355X: searches of 25 - 64 runs 1.5 x faster end to end, of 10 runs no faster)


def _minimize_batched(evaluator, jobs: list, on_device: Optional[bool] = False) -> None:
    """Advance every (circuit, run) pair to completion; one evaluate_circuits call per optimiser iteration of the whole
    set (SPSA proposes two points per run and iteration, NFT two or three).  ``on_device``: SPSA runs whose evaluator can
    read points from and leave values in device memory keep their whole state there (evqe/device_search.py)."""
    spsa = [] if os.environ.get("QSV_SCALAR_SPSA") else [job for job in jobs if isinstance(job[1], _SPSARun) and not job[1].done]
    if len(spsa) > 1 and len(spsa) == sum(1 for job in jobs if not job[1].done) and all(job[1].config is spsa[0][1].config for job in spsa):
        env = os.environ.get("QSV_DEVICE_SEARCH")
        if on_device is None:  # (the solver's default: where it pays)
            on_device = len(spsa) >= _DEVICE_SEARCH_MIN_RUNS
        if (on_device or env == "1") and env != "0":
            from queasars_amd.evqe import device_search

            if device_search.supported(evaluator, spsa):
                device_search.minimize_spsa_on_device(evaluator, spsa)
                return
        _minimize_spsa_vectorised(evaluator, spsa)
        return
    active = [job for job in jobs if not job[1].done]
    while active:
        circuits, params, counts = [], [], []
        for circuit, run in active:
            points = run.propose()
            counts.append(len(points))
            circuits += [circuit] * len(points)
            params += [_full_point(run, p).tolist() for p in points]
        values = evaluator.evaluate_circuits(circuits, params)
        cur = 0
        for (_, run), k in zip(active, counts):
            run.accept(*values[cur : cur + k])
            cur += k
        active = [job for job in active if not job[1].done]


# ---- configuration / result -----------------------------------------------------------------------------------


class BestIndividualRelativeChangeTolerance:
    def __init__(self, minimum_relative_change: float, allowed_consecutive_violations: int = 0):
        if minimum_relative_change <= 0 or minimum_relative_change > 1:
            raise ValueError("The minimum relative improvement parameter must not exceed the range )0,1)!")
        if allowed_consecutive_violations < 0:
            raise ValueError("allowed_consecutive_violations must be at least 0!")
        self._threshold = minimum_relative_change
        self._window = allowed_consecutive_violations + 1
        self.reset_state()

    def reset_state(self) -> None:
        self._previous: Optional[float] = None
        self._history: list[float] = []

    def check_termination(self, best_expectation_value_of_generation: float) -> bool:
        if self._previous is None:
            self._previous = best_expectation_value_of_generation
            return False
        self._history.append(abs(self._previous - best_expectation_value_of_generation) / abs(self._previous))
        self._previous = best_expectation_value_of_generation
        if len(self._history) < self._window:
            return False
        return max(self._history[-self._window :]) < self._threshold


@dataclass
class EVQEMinimumEigensolverConfiguration:
    """The solver knobs of queasars/minimum_eigensolvers/evqe/evqe.py:34-177 that do not concern primitives/executors."""

    optimizer: object  # SPSA or NFT: anything with new_run(x0, seed) and n_circuit_evaluations
    population_size: int
    max_generations: Optional[int] = None
    max_circuit_evaluations: Optional[int] = None
    termination_criterion: Optional[BestIndividualRelativeChangeTolerance] = None
    random_seed: Optional[int] = None
    n_initial_layers: int = 1
    randomize_initial_population_parameters: bool = False
    speciation_genetic_distance_threshold: int = 2
    use_tournament_selection: bool = False
    tournament_size: Optional[int] = None
    selection_alpha_penalty: float = 0.1
    selection_beta_penalty: float = 0.1
    parameter_search_probability: float = 0.24
    topological_search_probability: float = 0.2
    layer_removal_probability: float = 0.05
    # SPSA searches keep iterates, points and values in device memory and never wait for the GPU inside a search
    # (evqe/device_search.py; iterates agree with the host driver's to the last bits, not bit for bit; the same stopping
    # iterations).  None: where the evaluator can do it and a search has at least 16 runs; False: never; True: whenever it can.
    device_resident_search: Optional[bool] = None
    # Layer searches of individuals whose circuits have no split form on the device (deep individuals: every evaluation is a few
    # passes over the 2^n state) keep the state in front of the searched layer on the device and evaluate from there
    # (mutation.py:57-59: only that layer's angles change) -- where the evaluator can (``keep_states``) and the circuit costs
    # say it pays.  None: yes; False: never (QSV_KEPT_STATES=0 / 1 overrides).
    kept_state_search: Optional[bool] = None

    def __post_init__(self):
        if self.population_size < 1:
            raise ValueError("population_size must be at least 1!")
        if self.max_generations is None and self.max_circuit_evaluations is None and self.termination_criterion is None:
            raise ValueError("At least one of max_generations, max_circuit_evaluations or termination_criterion is needed!")
        if self.use_tournament_selection and (self.tournament_size is None or self.tournament_size < 1):
            raise ValueError("tournament_size cannot be None, if tournament selection should be used!")
        for name in ("parameter_search_probability", "topological_search_probability", "layer_removal_probability"):
            if not 0 <= getattr(self, name) <= 1:
                raise ValueError(f"{name} must be in the range [0, 1]!")


@dataclass
class EVQEResult:
    eigenvalue: float
    best_individual: EVQEIndividual
    generations: int
    circuit_evaluations: list[int]
    best_expectation_values: list[float] = field(default_factory=list)
    median_expectation_values: list[float] = field(default_factory=list)
    mean_expectation_values: list[float] = field(default_factory=list)


# ---- the solver -----------------------------------------------------------------------------------------------


class EVQEMinimumEigensolver:
    def __init__(self, configuration: EVQEMinimumEigensolverConfiguration, log: Optional[Callable[[str], None]] = None,
                 on_generation: Optional[Callable[[dict], None]] = None, shard: bool = False, process_group=None):
        """``on_generation``: called when a generation has been scored, with {"generation", "population", "values",
        "circuit_evaluations" (so far)}.  ``shard``: one process per GPU (torch.distributed initialised), every rank running
        this same solver with the same seed: the individuals' optimiser runs and the fitness evaluations are dealt to the ranks
        (the reference hands each to a worker of its pool, mutation.py:194-235, selection.py:75-85) and ONE all-gather per
        search / per scoring gives every rank all results, so that the evolution is the same everywhere."""
        self.configuration = configuration
        self._log = log or (lambda message: None)
        self._on_generation = on_generation
        self.shard = bool(shard)
        self.process_group = process_group
        # (QSV_SHARE_CIRCUITS=0: a circuit per individual and search, the other layers' values bound into it, as the reference has it)
        self.share_circuits = os.environ.get("QSV_SHARE_CIRCUITS", "1") != "0"
        rng = Random(configuration.random_seed)
        # one seed per consumer, drawn in the reference's order (evqe.py:188-229)
        self._population_seed = new_random_seed(rng)
        self._rng_last_layer = Random(new_random_seed(rng))
        self._rng_speciation = Random(new_random_seed(rng))
        self._rng_selection = Random(new_random_seed(rng))
        self._rng_parameter_search = Random(new_random_seed(rng))
        self._rng_topological = Random(new_random_seed(rng))
        self._rng_layer_removal = Random(new_random_seed(rng))

    # -- mutation helpers -----------------------------------------------------------------------------------
    @staticmethod
    def _chosen(population: EVQEPopulation, rng: Random, probability: float) -> dict[int, int]:
        """index -> seed of the individuals an operator mutates (mutation.py:206-216)."""
        chosen = {}
        for i in range(len(population.individuals)):
            if rng.random() <= probability:
                chosen[i] = new_random_seed(rng)
        return chosen

    def _optimize_layers(self, evaluator, individuals: list[EVQEIndividual], layer_ids: list[int], seeds: list[int]):
        """optimize_layer_of_individual (mutation.py:28-89) for many individuals at once."""
        # The reference binds every other layer's values into the circuit (a new circuit per individual, layer and search:
        # get_partially_parameterized_quantum_circuit).  Here the search runs on the individual's FULLY parameterised circuit --
        # one shared object per structure, registered once --, the other layers' values travelling as parameter values that
        # do not move: the same matrices, gate for gate, hence the same numbers (tests hold the two forms to equality).
        # (Where the points are packed on the host -- small searches, evaluators that sample -- the wider rows cost more than the
        # registrations they save, measured on config 4's sampler branch: there the circuits are bound as the reference binds them.)
        embed = self.share_circuits and _device_search_wanted(evaluator, len(individuals), self.configuration.device_resident_search,
                                                              self.configuration.optimizer)
        if os.environ.get("QSV_SHARE_CIRCUITS") == "2":  # (measurements, tests: embedded also where the host packs the points)
            embed = True
        jobs = []
        for k, (individual, layer_id, seed) in enumerate(zip(individuals, layer_ids, seeds)):
            run = self.configuration.optimizer.new_run(individual.get_layer_parameter_values(layer_id), seed)
            if embed:
                circuit = individual.get_parameterized_quantum_circuit(shared=True)
                # (positions and the other layers' values in the CIRCUIT's parameter order -- name-sorted blocks, which is
                # the layer order only up to ten layers: layer10_ sorts before layer2_)
                layer = layer_id % len(individual.layers)
                start = individual.circuit_parameter_offsets[layer]
                positions = np.arange(start, start + individual.layers[layer].n_parameters, dtype=np.int64)
                run.embed = (np.asarray(individual.parameter_values_in_circuit_order(), dtype=np.float64), positions)
            else:
                circuit = individual.get_partially_parameterized_quantum_circuit({layer_id})
            jobs.append((circuit, run))
        # several ranks: each runs the searches of ITS individuals (dealt by what their circuits cost), one all-gather at the end
        mine = None
        if self.shard:
            from queasars_amd import distributed

            mine = distributed.shard_searches(evaluator, jobs, self.process_group)
        own = list(range(len(jobs))) if mine is None else list(mine)
        # deep individuals of this rank: from a kept state (the searched layer and what follows it: the layer's own parameters)
        kept = self._kept_state_circuits(evaluator, [individuals[j] for j in own], [layer_ids[j] for j in own])
        for position, circuit in kept.items():
            j = own[position]
            jobs[j][1].embed = None
            jobs[j] = (circuit, jobs[j][1])
        _minimize_batched(evaluator, [jobs[j] for j in own], on_device=self.configuration.device_resident_search)
        if kept:  # (nothing may hold the search's circuits any longer: their kept states are a 2^n-amplitude buffer each)
            evaluator.forget_circuits()
        if mine is not None:
            distributed.gather_search_results(jobs, own, self.process_group)
        out, nfev = [], 0
        for individual, layer_id, (_, run) in zip(individuals, layer_ids, jobs):
            out.append(EVQEIndividual.change_layer_parameter_values(individual, layer_id, tuple(run.x.tolist())))
            nfev += run.nfev
        return out, nfev

    def _kept_state_circuits(self, evaluator, individuals: list[EVQEIndividual], layer_ids: list[int]) -> dict:
        """position -> the circuit a layer search evaluates from a KEPT state: for individuals whose whole circuit takes gate
        passes over the 2^n state on the evaluator's device (no split form: deep individuals), the state in front of the
        searched layer is computed once, kept resident, and every evaluation of the search applies only the searched layer and
        what follows it (reference: the layers in front are bound and never change, mutation.py:57-59).  Taken where the
        evaluator reports fewer, cheaper passes for it (last-layer searches of deep individuals: 1.5 - 1.8 x at 20 / 24
        qubits and eight layers, profiles/r04_prefix_reuse.txt); a search in the middle of a deep circuit keeps its whole
        circuit (the same number of passes either way).  The values agree with the whole circuit's to rounding (1e-14), not
        bit for bit."""
        flag = self.configuration.kept_state_search
        env = os.environ.get("QSV_KEPT_STATES")
        if env == "0" or (flag is False and env != "1") or not hasattr(evaluator, "keep_states") or not hasattr(evaluator, "circuit_costs"):
            return {}
        force = env == "1" or flag is True  # (asked for: taken wherever the rest of the circuit is cheaper than the whole)
        candidates = [k for k, (ind, layer) in enumerate(zip(individuals, layer_ids)) if layer % len(ind.layers) > 0]
        if not candidates:
            return {}
        whole = [individuals[k].get_parameterized_quantum_circuit(shared=True) for k in candidates]
        first = evaluator.circuit_costs(whole[:1])[0]
        if first["route"] == "one tile":  # (a register of one tile: nothing to save)
            return {}
        costs = [first] + evaluator.circuit_costs(whole[1:])
        # Worth it?  A kept state costs about one and a half whole evaluations (the circuit in front run once, its state copied)
        # plus the registration of the rest; every evaluation of the search then saves (whole - rest) microseconds.  A search
        # makes at most n_circuit_evaluations of them -- about a third of that where a termination checker ends runs early
        # (measured on the benchmark's trajectory: 22 of 66).  The rest's cost is estimated from its layers: the passes of the
        # whole circuit scale with its depth, a single last layer takes two.
        optimizer = self.configuration.optimizer
        expected = float(optimizer.n_circuit_evaluations)
        if getattr(optimizer, "termination_checker", None) is not None:
            expected /= 3.0
        deep = []
        for k, c in zip(candidates, costs):
            if c["route"] != "gate passes":
                continue
            n_layers = len(individuals[k].layers)
            rest_layers = n_layers - layer_ids[k] % n_layers
            rest_passes = max(2.0, c["n_passes"] * rest_layers / n_layers + 1.0)
            per_pass = c["microseconds"] / max(1, c["n_passes"])
            saving = c["microseconds"] - 1.1 * per_pass * rest_passes
            if force or saving * expected > 1.5 * c["microseconds"] + 60.0:
                deep.append((k, c))
        if not deep:
            return {}
        fronts = [individuals[k].get_layer_search_state_circuit(layer_ids[k]) for k, _ in deep]
        states = evaluator.keep_states([circuit for circuit, _ in fronts], [list(values) for _, values in fronts])
        circuits = [individuals[k].get_layer_search_circuits(layer_ids[k])[1].continue_from(state) for (k, _), state in zip(deep, states)]
        out = {}
        for (k, cost), circuit, kept_cost in zip(deep, circuits, evaluator.circuit_costs(circuits)):
            if kept_cost["microseconds"] < 0.9 * cost["microseconds"]:
                out[k] = circuit
        return out

    def _last_layer_search(self, evaluator, population: EVQEPopulation) -> tuple[EVQEPopulation, int]:
        chosen = self._chosen(population, self._rng_last_layer, 1.0)
        idx = sorted(chosen)
        new, nfev = self._optimize_layers(evaluator, [population.individuals[i] for i in idx], [-1] * len(idx), [chosen[i] for i in idx])
        individuals = list(population.individuals)
        for i, ind in zip(idx, new):
            individuals[i] = ind
        return EVQEPopulation(tuple(individuals), population.species_representatives, None, None), nfev

    def _parameter_search(self, evaluator, population: EVQEPopulation) -> tuple[EVQEPopulation, int]:
        """optimize_all_parameters_of_individual (mutation.py:92-132): layers in a random order, one at a time; the
        j-th layer of every chosen individual is optimised in the same batch."""
        chosen = self._chosen(population, self._rng_parameter_search, self.configuration.parameter_search_probability)
        state = {}
        for i, seed in chosen.items():
            randomizer = Random(seed)
            state[i] = [population.individuals[i], list(range(len(population.individuals[i].layers))), randomizer]
        total = 0
        while any(todo for _, todo, _ in state.values()):
            batch = [i for i, (_, todo, _) in state.items() if todo]
            layers, seeds = [], []
            for i in batch:
                _, todo, randomizer = state[i]
                layer = randomizer.choice(todo)
                todo.remove(layer)
                layers.append(layer)
                seeds.append(new_random_seed(randomizer))
            new, nfev = self._optimize_layers(evaluator, [state[i][0] for i in batch], layers, seeds)
            total += nfev
            for i, ind in zip(batch, new):
                state[i][0] = ind
        individuals = list(population.individuals)
        for i, (ind, _, _) in state.items():
            individuals[i] = ind
        return EVQEPopulation(tuple(individuals), population.species_representatives, None, None), total

    def _topological_search(self, population: EVQEPopulation) -> EVQEPopulation:
        chosen = self._chosen(population, self._rng_topological, self.configuration.topological_search_probability)
        individuals = list(population.individuals)
        for i, seed in chosen.items():
            individuals[i] = EVQEIndividual.add_random_layers(individuals[i], 1, False, random_seed=seed)
        return EVQEPopulation(tuple(individuals), population.species_representatives, None, None)

    def _layer_removal(self, population: EVQEPopulation) -> EVQEPopulation:
        chosen = self._chosen(population, self._rng_layer_removal, self.configuration.layer_removal_probability)
        individuals = list(population.individuals)
        for i, seed in chosen.items():
            ind = individuals[i]
            if len(ind.layers) > 1:
                individuals[i] = EVQEIndividual.remove_layers(ind, Random(seed).randrange(1, len(ind.layers)))
        return EVQEPopulation(tuple(individuals), population.species_representatives, None, None)

    # -- speciation / selection -----------------------------------------------------------------------------
    def _speciation(self, population: EVQEPopulation) -> EVQEPopulation:
        threshold = self.configuration.speciation_genetic_distance_threshold
        representatives = list(population.species_representatives or [])
        members: dict[EVQEIndividual, list[int]] = {rep: [] for rep in representatives}
        if 0 < threshold <= 1:
            # A genetic distance below one is zero: the same number of layers and every layer equal (individual.py:217-237), and
            # an equal individual has equal layers too -- the first representative with the individual's layers, found by
            # hashing instead of by comparing with every representative in turn (36 k distance calls per run of the
            # benchmark's trajectory, a sixth of the driver's time; the notebooks' threshold is 1).
            first = {}
            for rep in representatives:
                first.setdefault(rep.layers, rep)
            for i, individual in enumerate(population.individuals):
                rep = first.get(individual.layers)
                if rep is None:
                    representatives.append(individual)
                    members[individual] = [i]
                    first[individual.layers] = individual
                else:
                    members[rep].append(i)
        else:
            for i, individual in enumerate(population.individuals):
                for rep in representatives:
                    if EVQEIndividual.get_genetic_distance(individual, rep) < threshold or individual == rep:
                        members[rep].append(i)
                        break
                else:
                    representatives.append(individual)
                    members[individual] = [i]
        new_members: dict[EVQEIndividual, list[int]] = {}
        for group in members.values():
            if not group:
                continue
            representative = population.individuals[self._rng_speciation.choice(group)]
            new_members.setdefault(representative, []).extend(group)
        membership = {i: rep for rep, group in new_members.items() for i in group}
        return EVQEPopulation(population.individuals, list(new_members), new_members, membership)

    def _selection(self, population: EVQEPopulation, values: list[float]) -> EVQEPopulation:
        cfg = self.configuration
        rng = self._rng_selection
        n = len(population.individuals)
        species_size = [float(len(population.species_members[population.species_membership[i]])) for i in range(n)]
        penalties = [
            cfg.selection_alpha_penalty * len(ind.layers) + cfg.selection_beta_penalty * ind.get_n_controlled_gates()
            for ind in population.individuals
        ]
        if not cfg.use_tournament_selection:
            best = min(values)
            offset = -best + 1 if best <= 0 else 0
            fitness = [(values[i] + offset + penalties[i]) * species_size[i] for i in range(n)]
            selected = rng.choices(population.individuals, weights=[1 / (f + offset) for f in fitness], k=n)
        else:
            fitness = [(values[i] + penalties[i]) * species_size[i] for i in range(n)]
            selected = []
            while len(selected) < n:
                contenders = rng.choices(range(n), k=cfg.tournament_size)
                winner = min(contenders, key=lambda j: (fitness[j], contenders.index(j)))
                selected.append(population.individuals[winner])
        return EVQEPopulation(tuple(selected), population.species_representatives, None, None)

    # -- main loop ------------------------------------------------------------------------------------------
    def compute_minimum_eigenvalue(self, evaluator, search_evaluator=None) -> EVQEResult:
        """``search_evaluator``: a second evaluator of the SAME operator for the parameter searches alone -- a single-precision
        handle: the points a search compares differ by far more than 1e-6 (the reference's own tests run their searches on an
        estimator with precision 0.05, test/minimum_eigensolvers/evqe/solver.py:20-27), a layer search on kept states is bound
        by the bytes of the state, and fp32 halves them (bench.py layer_search: 227 k against 160 k evaluations per second at
        seven layers, 170 k against 107 k at eight).  Every individual's FITNESS, what selection and the result see, comes from
        ``evaluator``."""
        cfg = self.configuration
        searcher = evaluator if search_evaluator is None else search_evaluator
        if searcher.n_qubits != evaluator.n_qubits:
            raise ValueError("the search evaluator must be over the same register")
        if cfg.termination_criterion is not None:
            cfg.termination_criterion.reset_state()
        population = EVQEPopulation.random_population(
            evaluator.n_qubits, cfg.n_initial_layers, cfg.population_size, cfg.randomize_initial_population_parameters,
            random_seed=self._population_seed,
        )
        evaluations: list[int] = [0]
        result = EVQEResult(eigenvalue=math.inf, best_individual=population.individuals[0], generations=0, circuit_evaluations=evaluations)
        n_opt = cfg.optimizer.n_circuit_evaluations

        def budget_left(expected: int) -> bool:
            if cfg.max_circuit_evaluations is None:
                return True
            used = sum(evaluations)
            return used < cfg.max_circuit_evaluations and used + expected < cfg.max_circuit_evaluations

        while True:
            if cfg.max_generations is not None and result.generations >= cfg.max_generations:
                break
            # 1. last-layer parameter search on everyone
            if not budget_left(len(population.individuals) * n_opt):
                break
            population, nfev = self._last_layer_search(searcher, population)
            evaluations[-1] += nfev
            # 2. speciation, 3. selection (this is where a generation is scored)
            population = self._speciation(population)
            if not budget_left(len(population.individuals)):
                break
            circuits = [ind.get_parameterized_quantum_circuit(shared=self.share_circuits) for ind in population.individuals]
            if self.shard:
                from queasars_amd.distributed import evaluate_population_sharded

                values = evaluate_population_sharded(evaluator, circuits, [list(ind.parameter_values) for ind in population.individuals],
                                                     group=self.process_group)
            else:
                values = evaluator.evaluate_circuits(circuits, [list(ind.parameter_values) for ind in population.individuals])
            evaluations[-1] += len(values)
            best = int(np.argmin(values))
            if values[best] < result.eigenvalue:
                result.eigenvalue, result.best_individual = values[best], population.individuals[best]
            result.best_expectation_values.append(values[best])
            result.median_expectation_values.append(median(values))
            result.mean_expectation_values.append(mean(values))
            self._log(f"generation {result.generations}: best {values[best]:.6f} median {median(values):.6f} mean {mean(values):.6f}")
            if self._on_generation is not None:
                self._on_generation({"generation": result.generations, "population": population, "values": list(values),
                                     "circuit_evaluations": sum(evaluations)})
            result.generations += 1
            evaluations.append(0)
            terminate = cfg.termination_criterion is not None and cfg.termination_criterion.check_termination(values[best])
            population = self._selection(population, values)
            if terminate or (cfg.max_generations is not None and result.generations >= cfg.max_generations):
                break
            # 4. full parameter search, 5. growth, 6. pruning
            expected = math.ceil(cfg.parameter_search_probability * sum(len(i.layers) for i in population.individuals) * n_opt)
            if not budget_left(expected):
                break
            population, nfev = self._parameter_search(searcher, population)
            evaluations[-1] += nfev
            population = self._topological_search(population)
            population = self._layer_removal(population)
        if evaluations and evaluations[-1] == 0 and len(evaluations) > 1:
            evaluations.pop()
        return result
