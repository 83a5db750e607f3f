"""Coverage study: how many deep EVQE individuals split into THREE weakly coupled parts? (CPU only)

csrc/split.cpp cuts a register in two: psi = sum_kappa a_kappa (x) b_kappa over the cut KEYS (a key = one control qubit between
two gates that target it; a cu3 whose control sits on another side than its target doubles the product terms unless its key
is already cut).  At seven and eight layers almost nothing has such a form with <= 5 keys (profiles/r03_split_by_depth.txt).
A three-part form  psi = sum a (x) b (x) c  would need, per side, own qubits + the cut keys that touch it <= tile + 4 (the
largest virtual circuit the pass kernel takes), and few keys per pair of sides (2^keys product terms per Gram matrix).

This script restates split.cpp's key extraction (same drop rule, same (control, epoch) keys) and searches three-way
assignments of the qubits by hill climbing from many random starts -- a LOWER bound on the coverage (a found partition is a
partition; a circuit without one may still have one).  For comparison it runs the same search for two sides and reports what
the library's exact search finds (qsv_split_describe).

    python scripts/three_part_study.py [restarts]
"""
import json
import random
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from queasars_amd import _lib  # noqa: E402
from queasars_amd.evqe import EVQEPopulation  # noqa: E402
from queasars_amd.ir import QSV_OP_DTYPE  # noqa: E402


def keys_of(circuit):
    """split.cpp's keys: [(control, {targets})] of the cu3 gates that act (a cu3 whose control nobody has targeted yet is the
    identity), grouped by (control, number of gates that have targeted the control so far)."""
    n = circuit.n_qubits
    touched, epoch = [False] * n, [0] * n
    keys = {}
    for kind, target, control, *_ in circuit.bound_ops([0.3] * circuit.num_parameters):
        if kind == 0:
            continue
        if control >= 0:
            if not touched[control]:
                continue
            keys.setdefault((control, epoch[control]), set()).add(target)
        touched[target] = True
        epoch[target] += 1
    return [(c, sorted(ts)) for (c, _), ts in keys.items()]


def evaluate(side, keys, n_sides, limit, max_pair):
    """(penalty, total cut keys, per-side virtual sizes, per-pair keys) of an assignment."""
    own = [0] * n_sides
    for s in side:
        own[s] += 1
    touch = [0] * n_sides
    pair = {}
    cut = 0
    for control, targets in keys:
        sc = side[control]
        others = {side[t] for t in targets} - {sc}
        if not others:
            continue
        cut += 1
        touch[sc] += 1
        for s in others:
            touch[s] += 1
            p = (min(sc, s), max(sc, s))
            pair[p] = pair.get(p, 0) + 1
    penalty = sum(max(0, own[s] + touch[s] - limit) for s in range(n_sides)) + sum(max(0, v - max_pair) for v in pair.values())
    if min(own) == 0:
        penalty += 100
    return penalty, cut, [own[s] + touch[s] for s in range(n_sides)], pair


def search(n, keys, n_sides, limit, max_pair, restarts, rng):
    best = None
    for _ in range(restarts):
        # start: contiguous blocks of a random rotation of the qubits (EVQE couplings are random pairs: no locality to use)
        order = list(range(n))
        rng.shuffle(order)
        side = [0] * n
        for i, q in enumerate(order):
            side[q] = i * n_sides // n
        score = evaluate(side, keys, n_sides, limit, max_pair)
        improved = True
        while improved:
            improved = False
            qs = list(range(n))
            rng.shuffle(qs)
            for q in qs:
                old = side[q]
                for s in range(n_sides):
                    if s == old:
                        continue
                    side[q] = s
                    trial = evaluate(side, keys, n_sides, limit, max_pair)
                    if (trial[0], trial[1]) < (score[0], score[1]):
                        score, old, improved = trial, s, True
                side[q] = old
        if score[0] == 0 and (best is None or score[1] < best[1]):
            best = (list(side), score[1], score[2], score[3])
            if score[1] <= n_sides:  # good enough: stop early
                break
    return best


def library_split(circuit, max_side):
    lib = _lib.load()
    ops = circuit.packed()
    mask = _lib.C.c_uint64(0)
    na, nb = _lib.C.c_int(0), _lib.C.c_int(0)
    return lib.qsv_split_describe(circuit.n_qubits, len(ops), _lib.as_ptr(ops), max_side, _lib.C.byref(mask), None, 0, _lib.C.byref(na),
                                  None, 0, _lib.C.byref(nb))


def main():
    restarts = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(0)
    for n, tile, pop in ((20, 12, 64), (24, 13, 32)):
        limit = tile + 4
        for layers in (6, 7, 8):
            population = EVQEPopulation.random_population(n, layers, pop, True, 0)
            t0 = time.time()
            two_lib = three = two_heur = 0
            cuts3, sizes3, pairs3 = [], [], []
            for ind in population.individuals:
                circuit = ind.get_parameterized_quantum_circuit()
                keys = keys_of(circuit)
                k2 = library_split(circuit, limit)
                two_lib += k2 >= 0
                if k2 >= 0:
                    continue  # (already served by the two-part form)
                two = search(n, keys, 2, limit, 5, restarts, rng)
                two_heur += two is not None
                found = search(n, keys, 3, limit, 5, restarts, rng)
                if found is not None:
                    three += 1
                    cuts3.append(found[1])
                    sizes3.append(max(found[2]))
                    pairs3.append(max(found[3].values()) if found[3] else 0)
            print(json.dumps({
                "n": n, "layers": layers, "individuals": pop, "side_limit_qubits": limit, "max_keys_per_pair": 5,
                "two_parts_library_exact_search": two_lib,
                "of_the_rest_two_parts_by_this_heuristic": two_heur,
                "of_the_rest_three_parts_found": three,
                "three_parts_total_cut_keys": {"min": min(cuts3, default=None), "median": float(np.median(cuts3)) if cuts3 else None, "max": max(cuts3, default=None)},
                "three_parts_largest_virtual_circuit_qubits": {"median": float(np.median(sizes3)) if sizes3 else None},
                "three_parts_most_keys_on_one_pair": {"median": float(np.median(pairs3)) if pairs3 else None},
                "restarts": restarts, "seconds": round(time.time() - t0, 1),
            }), flush=True)


if __name__ == "__main__":
    main()
