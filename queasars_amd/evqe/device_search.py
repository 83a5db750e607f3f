"""The lock-step SPSA search of a population with its state ON THE DEVICE.

``solver._minimize_spsa_vectorised`` keeps every run's iterate in host memory: an iteration builds the 2 R points in NumPy,
hands them to the evaluator (packing, PCIe), waits for the 2 R values and updates the iterates -- and the GPU idles while
the host does its share (8.0 ms per search of 64 individuals at 20 qubits, of which the device is busy for 3).  Here the
iterates, the pre-drawn sign vectors, the points and the function values are tensors in device memory, the evaluator reads
the points where they are and leaves the values where the update reads them (``qsv_eval_push_device`` /
``qsv_eval_set_output``), and everything of an iteration -- proposal, evaluation, update, the termination rule -- is queued
on ONE HIP stream without the host waiting for any of it; the host looks at the device every few iterations only to see
whether every run has stopped.

Arithmetic: element by element the expressions of ``_SPSARun.propose`` / ``accept`` (reference: qiskit_algorithms' SPSA with
constant gains as the notebook configures it, mutation.py:63-75 for the batched callback), in the same order; the one
difference is the trust region's norm, summed by the device in its own order, so an iterate can differ from the host
driver's in the last bits (tests hold the two to 1e-9 and to the same stopping iterations).  The termination rule is the
reference's ``SPSATerminationChecker`` (queasars/utility/spsa_termination.py:46-94) as array operations; the runs' host-side
checker objects are not fed (nothing reads them afterwards).  Runs that have stopped stay in the batch with their updates
masked -- taking them out would mean waiting for the device --, and are not counted: ``nfev`` is two per iteration a run
was active, as on the host.
"""

from __future__ import annotations

import numpy as np


_MAX_SIGN_BYTES = 256 << 20


def supported(evaluator, jobs) -> bool:
    """Can :func:`minimize_spsa_on_device` take these jobs?  An exact estimator on a GPU, fresh SPSA runs of one configuration."""
    if len(jobs) < 2 or not hasattr(evaluator, "evaluate_device_to_device"):
        return False
    if not evaluator.device_resident_search_possible():
        return False
    runs = [run for _, run in jobs]
    cfg = runs[0].config
    if any(run.config is not cfg or run.iteration != 0 or run.nfev != 0 or run.done for run in runs):
        return False
    # (the sign vectors of every iteration are drawn ahead, as float64 rows of the widest run's width -- on the host and again on
    # the device: long optimisations of many deep individuals with embedded parameter vectors would be hundreds of megabytes;
    # beyond a quarter of a gigabyte the host driver, which draws them iteration by iteration, takes the search)
    width = max(run.embed[0].size if run.embed is not None else run.x.size for run in runs)
    if int(cfg.maxiter) * len(runs) * width * 8 > _MAX_SIGN_BYTES:  # (as doubles on the device; bytes on the host)
        return False
    checker = cfg.termination_checker
    return cfg.maxiter > 0 and (checker is None or type(checker).__name__ == "SPSATerminationChecker")


def minimize_spsa_on_device(evaluator, jobs, look_every: int = 8) -> None:
    """One launch per iteration for the optimiser's share (``qsv_spsa_step``: accept iteration k, propose iteration k + 1) and
    one for the evaluation.  ``QSV_DEVICE_SEARCH_TORCH=1``: the same arithmetic as a few dozen torch operations per
    iteration (:func:`_minimize_with_torch_operations`; the tests hold the two against each other)."""
    import ctypes as C
    import os

    import torch

    from queasars_amd import _lib
    from queasars_amd.distributed import _chain_state

    if os.environ.get("QSV_DEVICE_SEARCH_TORCH") == "1":
        return _minimize_with_torch_operations(evaluator, jobs, look_every)
    runs = [run for _, run in jobs]
    cfg = runs[0].config
    n_iter = int(cfg.maxiter)
    # A run's variables may be entries of a longer parameter vector (run.embed: a layer inside the individual's fully
    # parameterised circuit): the row is that vector, the signs are zero everywhere else -- x +- eps * 0 leaves the other
    # entries where they are, the update is zero there and the norm does not see them.
    where = [run.embed[1] if run.embed is not None else np.arange(run.x.size) for run in runs]
    lengths = np.array([run.embed[0].size if run.embed is not None else run.x.size for run in runs])
    width, n_runs = int(lengths.max()), len(runs)
    x_host = np.zeros((n_runs, width))
    # (the signs travel as bytes -- an eighth of the transfer, which was a tenth of a short search -- and become doubles on the device)
    signs_host = np.zeros((n_iter, n_runs, width), dtype=np.int8)
    for i, run in enumerate(runs):
        if run.embed is not None:
            x_host[i, : lengths[i]] = run.embed[0]
        x_host[i, where[i]] = run.x
        signs_host[:, i, where[i]] = 1 - 2 * run.rng.binomial(1, 0.5, size=(n_iter, run.x.size))
    circuits = [circuit for circuit, _ in jobs for _ in (0, 1)]
    checker = cfg.termination_checker
    window = checker.allowed_consecutive_violations + 1 if checker is not None else 0

    dev = evaluator.statevector_device
    device = torch.device("cuda", dev.device_index)
    stream = _chain_state(evaluator, device)["stream"]  # (the stream the evaluator's handle launches on)
    caller = torch.cuda.current_stream(device)
    stream.wait_stream(caller)
    lib, handle = dev._lib, dev._handle
    with torch.cuda.stream(stream):
        x = torch.from_numpy(x_host).to(device)
        signs = torch.from_numpy(signs_host).to(device).to(torch.float64)
        points = torch.empty((2 * n_runs, width), dtype=torch.float64, device=device)
        values = torch.empty(2 * n_runs, dtype=torch.float64, device=device)
        active = torch.ones(n_runs, dtype=torch.uint8, device=device)
        iterations = torch.zeros(n_runs, dtype=torch.int64, device=device)
        previous = torch.zeros(n_runs, dtype=torch.float64, device=device)
        n_values = torch.zeros(n_runs, dtype=torch.int64, device=device)
        changes = torch.full((n_runs, max(window, 1)), float("inf"), dtype=torch.float64, device=device)
        args = _lib.QsvSpsaStepArgs(
            n_runs=n_runs, width=width, x=x.data_ptr(), active=active.data_ptr(), iterations=iterations.data_ptr(),
            delta_accept=None, values=None, delta_propose=None, points=points.data_ptr(), eps=cfg.perturbation,
            lr=cfg.learning_rate, trust_region=int(bool(cfg.trust_region)), maxiter=n_iter, window=window, reserved=0,
            min_rel=checker.minimum_relative_change if checker is not None else 0.0,
            maxfev=checker.maxfev if checker is not None and checker.maxfev is not None else -1,
            previous=previous.data_ptr(), n_values=n_values.data_ptr(), changes=changes.data_ptr())
        base, stride = signs.data_ptr(), n_runs * width * 8
        for k in range(n_iter + 1):
            # accept iteration k - 1 (its values are in `values`), propose iteration k
            args.delta_accept = base + (k - 1) * stride if k > 0 else None
            args.values = values.data_ptr() if k > 0 else None
            args.delta_propose = base + k * stride if k < n_iter else None
            dev._check(lib.qsv_spsa_step(handle, C.byref(args)))
            if k == n_iter:
                break
            if k > 0 and k % look_every == 0 and not bool(active.any()):
                break
            evaluator.evaluate_device_to_device(circuits, points, values)
        x_final = x.cpu().numpy()
        done_iterations = iterations.cpu().numpy()
    caller.wait_stream(stream)
    for i, run in enumerate(runs):
        run.x = x_final[i, where[i]].copy()
        run.iteration = int(done_iterations[i])
        run.nfev = 2 * int(done_iterations[i])
        run.done = True


def _minimize_with_torch_operations(evaluator, jobs, look_every: int = 8) -> None:
    import torch

    from queasars_amd.distributed import _chain_state

    runs = [run for _, run in jobs]
    cfg = runs[0].config
    eps, lr, n_iter = cfg.perturbation, cfg.learning_rate, int(cfg.maxiter)
    where = [run.embed[1] if run.embed is not None else np.arange(run.x.size) for run in runs]
    lengths = np.array([run.embed[0].size if run.embed is not None else run.x.size for run in runs])
    width, n_runs = int(lengths.max()), len(runs)
    x_host = np.zeros((n_runs, width))
    signs_host = np.zeros((n_iter, n_runs, width), dtype=np.int8)
    for i, run in enumerate(runs):
        if run.embed is not None:
            x_host[i, : lengths[i]] = run.embed[0]
        x_host[i, where[i]] = run.x
        # (what propose() would draw call by call: one draw of the lot gives the same numbers)
        signs_host[:, i, where[i]] = 1 - 2 * run.rng.binomial(1, 0.5, size=(n_iter, run.x.size))
    circuits = [circuit for circuit, _ in jobs for _ in (0, 1)]
    checker = cfg.termination_checker
    window = checker.allowed_consecutive_violations + 1 if checker is not None else 0

    dev = evaluator.statevector_device
    device = torch.device("cuda", dev.device_index)
    stream = _chain_state(evaluator, device)["stream"]  # (the stream the evaluator's handle launches on)
    caller = torch.cuda.current_stream(device)
    stream.wait_stream(caller)
    with torch.cuda.stream(stream):
        x = torch.from_numpy(x_host).to(device)
        signs = torch.from_numpy(signs_host).to(device).to(torch.float64)
        points = torch.empty((2 * n_runs, width), dtype=torch.float64, device=device)
        values = torch.empty(2 * n_runs, dtype=torch.float64, device=device)
        active = torch.ones(n_runs, dtype=torch.bool, device=device)
        iterations = torch.zeros(n_runs, dtype=torch.int64, device=device)
        one = torch.ones(n_runs, dtype=torch.float64, device=device)
        if checker is not None:
            previous = torch.zeros(n_runs, dtype=torch.float64, device=device)
            n_values = torch.zeros(n_runs, dtype=torch.int64, device=device)
            # the last `window` relative changes of every run, oldest first; +inf = not there yet
            changes = torch.full((n_runs, window), float("inf"), dtype=torch.float64, device=device)
        for k in range(n_iter):
            delta = signs[k]
            shift = delta * eps
            torch.add(x, shift, out=points[0::2])
            torch.sub(x, shift, out=points[1::2])
            evaluator.evaluate_device_to_device(circuits, points, values)
            f_plus, f_minus = values[0::2], values[1::2]
            update = ((f_plus - f_minus) / (2 * eps))[:, None] * delta
            if cfg.trust_region:
                norm = torch.sqrt((update * update).sum(dim=1))
                update = update / torch.where(norm > 1, norm, one)[:, None]
            update = update * lr
            x = x - update * active[:, None]
            iterations = iterations + active
            stop = iterations >= n_iter
            if checker is not None:
                # SPSATerminationChecker.termination_check with accepted = True, for every active run at once
                nfev = 2 * iterations
                if checker.maxfev is not None:
                    over = nfev >= checker.maxfev
                    stop = stop | over
                    fed = active & ~over  # (the reference returns before it stores anything)
                else:
                    fed = active
                value = 0.5 * (f_plus + f_minus)
                has_previous = fed & (n_values >= 1)
                change = (value - previous).abs() / previous
                shifted = torch.cat([changes[:, 1:], change[:, None]], dim=1)
                changes = torch.where(has_previous[:, None], shifted, changes)
                previous = torch.where(fed, value, previous)
                n_values = n_values + fed
                converged = has_previous & (changes.max(dim=1).values < checker.minimum_relative_change)
                stop = stop | converged
            active = active & ~stop
            if (k + 1) % look_every == 0 and k + 1 < n_iter and not bool(active.any()):
                break
        x_final = x.cpu().numpy()
        done_iterations = iterations.cpu().numpy()
    caller.wait_stream(stream)
    for i, run in enumerate(runs):
        run.x = x_final[i, where[i]].copy()
        run.iteration = int(done_iterations[i])
        run.nfev = 2 * int(done_iterations[i])
        run.done = True
