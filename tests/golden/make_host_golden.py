"""Golden vectors for two small host-side pieces next to the hot path, produced BY THE REFERENCE ITSELF (both modules are
plain Python / NumPy and import in the build container from /root/reference):

* ``SPSATerminationChecker.termination_check`` (queasars/utility/spsa_termination.py:46-94): what it answers, call by call,
  along sequences of (evaluations so far, function value, accepted) -- converging, noisy, with rejected steps, with a budget,
  and with one object serving several optimisations in a row; plus its bookkeeping after the last call;
* ``new_random_seed`` (queasars/utility/random.py:7-15): the seed chain of a ``random.Random``;
* ``BitstringEvaluator.evaluate_bitstring`` (queasars/circuit_evaluation/bitstring_evaluation.py:35-46): values and refusals.

    python tests/golden/make_host_golden.py        # writes tests/golden/spsa_termination_reference.json
"""
import json
import sys
from pathlib import Path
from random import Random

import numpy as np

sys.path.insert(0, "/root/reference")
from queasars.utility.random import new_random_seed  # noqa: E402
from queasars.utility.spsa_termination import SPSATerminationChecker  # noqa: E402


def sequences():
    rng = np.random.default_rng(5)
    out = []
    for case, (rel, allowed, maxfev) in enumerate([(0.01, 2, None), (0.05, 0, None), (0.01, 1, 40), (0.2, 3, None), (0.001, 2, 30), (0.5, 0, 6)]):
        calls = []
        for run in range(3):  # (the same checker object, one optimisation after another)
            value, nfev = float(rng.uniform(5, 30)) * (-1 if case == 3 and run == 1 else 1), 0
            for step in range(int(rng.integers(6, 25))):
                nfev += 2
                accepted = bool(rng.random() > 0.15)
                # (a value history that converges geometrically with noise on top)
                value = value * (1 - 0.3 * 0.7**step) + float(rng.normal(0, 0.02 * (case % 3)))
                calls.append({"nfev": nfev, "x": [float(v) for v in rng.normal(size=3)], "value": value,
                              "step_size": float(rng.uniform(0, 1)), "accepted": accepted})
        out.append({"minimum_relative_change": rel, "allowed_consecutive_violations": allowed, "maxfev": maxfev, "calls": calls})
    return out


def main() -> None:
    cases = sequences()
    for case in cases:
        checker = SPSATerminationChecker(case["minimum_relative_change"], case["allowed_consecutive_violations"], case["maxfev"])
        for call in case["calls"]:
            call["answer"] = bool(checker.termination_check(call["nfev"], np.asarray(call["x"]), call["value"], call["step_size"], call["accepted"]))
        try:
            best = checker.best_parameter_values
        except ValueError:  # (nothing stored since the history last started over: the property raises)
            best = None
        case["after"] = {"n_function_evaluations": checker.n_function_evaluations,
                         "function_value_history": list(checker.function_value_history),
                         "n_function_evaluation_history": list(checker.n_function_evaluation_history),
                         "best_function_value": checker.best_function_value,
                         "best_parameter_values": None if best is None else [float(v) for v in best]}
    # BitstringEvaluator (queasars/circuit_evaluation/bitstring_evaluation.py:7-57): what it returns or raises
    from queasars.circuit_evaluation.bitstring_evaluation import BitstringEvaluator, BitstringEvaluatorException

    evaluator = BitstringEvaluator(input_length=5, evaluation_function=lambda bits: float(int(bits, 2)) / 4 - bits.count("1"))
    bitstrings = []
    for text in ["00000", "11111", "01010", "10011", "0101", "010101", "", "01a10", "0 101", "01012", "１0101"]:
        try:
            bitstrings.append({"bitstring": text, "value": evaluator.evaluate_bitstring(text)})
        except BitstringEvaluatorException:
            bitstrings.append({"bitstring": text, "raises": "BitstringEvaluatorException"})
    seeds = {}
    for seed in (0, 1, 7, 2024):
        rng = Random(seed)
        seeds[str(seed)] = [new_random_seed(rng) for _ in range(6)]
    path = Path(__file__).resolve().parent / "spsa_termination_reference.json"
    path.write_text(json.dumps({"source": "the reference's own modules, run by tests/golden/make_host_golden.py",
                                "termination": cases, "seed_chains": seeds, "bitstring_evaluator": {"input_length": 5, "cases": bitstrings}},
                               separators=(",", ":")) + "\n")
    print(path, sum(len(c["calls"]) for c in cases), "calls,", sum(call["answer"] for c in cases for call in c["calls"]), "answered stop")


if __name__ == "__main__":
    main()
