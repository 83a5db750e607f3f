"""When the phases of a later gate pass run, workgroup by workgroup and compute unit by compute unit.

    python scripts/timeline.py build            # here (no GPU needed): writes queasars_amd/libqsv_timeline.so (-DQSV_TIMELINE)
    python scripts/timeline.py run [n P L]      # on the GPU box: one batch of P deep circuits, the later passes' timeline

A diagnostic build of the same sources: the first wave of every workgroup of a later pass writes down s_memtime where a
tile's loads are issued, where they are back (the production kernel's own vmcnt(0)), where the rounds are done and where the
stores are issued -- nothing else is drained -- and the compute unit it ran on (HW_ID, XCC_ID).  The script lays the
workgroups that shared a compute unit side by side and answers: for what share of a compute unit's time is at least one of
its workgroups in its rounds (issue) -- and for what share are all of them waiting for memory?
"""
import ctypes as C
import os
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
LIB = Path(os.environ.get("QSV_TIMELINE_LIB", ROOT / "queasars_amd" / "libqsv_timeline.so"))
WORDS, TILES, WGS = 6 + 4 * 14 + 2, 14, 16384


def covered(intervals):
    """Length of the union of intervals."""
    total, end = 0, None
    for a, b in sorted(intervals):
        if end is None or a > end:
            total += b - a
            end = b
        elif b > end:
            total += b - end
            end = b
    return total


def overlap_at_least(intervals, k):
    """Length of the set of points covered by at least k of the intervals."""
    events = sorted([(a, 1) for a, _ in intervals] + [(b, -1) for _, b in intervals])
    depth, last, total = 0, None, 0
    for x, d in events:
        if depth >= k and last is not None:
            total += x - last
        depth += d
        last = x
    return total


def main() -> None:
    if sys.argv[1] == "build":
        from queasars_amd import _build

        print(_build.build(force=True, defines=("QSV_TIMELINE",), lib_path=LIB))
        return
    os.environ["QSV_LIBRARY"] = str(LIB)
    import numpy as np
    from queasars_amd import workloads as helpers
    from queasars_amd import _lib
    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

    # (bench.py's deep rows: the same populations and operators)
    n, P, L = (int(x) for x in (sys.argv[2:5] + ["24", "32", "8"][len(sys.argv[2:5]):]))
    from queasars_amd.evqe import EVQEPopulation

    population = EVQEPopulation.random_population(n, L, P, True, 0)
    circuits = [ind.get_parameterized_quantum_circuit() for ind in population.individuals]
    params = [list(ind.parameter_values) for ind in population.individuals]
    ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=2020 if n == 20 else 2024), dtype=os.environ.get("QSV_TIMELINE_DTYPE", "fp64"))
    if os.environ.get("QSV_TIMELINE_SPLIT", "1") == "0":
        ev.statevector_device.set_option("split", 0)
    ev.statevector_device.set_option("streams", 1)  # a launch has the chip to itself
    lib = _lib.load()
    lib.qsv_debug_timeline.argtypes = [C.POINTER(C.c_ulonglong), C.c_size_t, C.POINTER(C.c_uint), C.c_int]
    lib.qsv_debug_timeline.restype = C.c_int
    table = (C.c_ulonglong * (WGS * WORDS))()
    count = C.c_uint(0)
    for _ in range(3):
        ev.evaluate_circuits(circuits, params)
    assert lib.qsv_debug_timeline(table, WGS * WORDS, C.byref(count), 1) == 0, "not a QSV_TIMELINE build"
    ev.evaluate_circuits(circuits, params)
    assert lib.qsv_debug_timeline(table, WGS * WORDS, C.byref(count), 1) == 0
    rec = np.frombuffer(table, dtype=np.uint64)[: count.value * WORDS].reshape(count.value, WORDS).astype(np.int64)
    print(f"n={n} P={P} L={L}: {count.value} workgroups of later passes recorded (first wave of each)")
    if os.environ.get("QSV_TIMELINE_DUMP"):
        np.save(os.environ["QSV_TIMELINE_DUMP"], rec)
    passes = sorted(set(int(r[1] >> 32) & 0xFF for r in rec))
    for p in passes:
        rows = [r for r in rec if (int(r[1] >> 32) & 0xFF) == p]
        report(np.array(rows), p)


def report(rec, p) -> None:
    import numpy as np

    # clock: s_memtime ticks per 10 ns of s_memrealtime, over the workgroups' own lifetimes
    life_real = (rec[:, 4] - rec[:, 3]).astype(np.float64) * 10e-9  # seconds
    n_t = np.minimum(rec[:, 2], TILES)
    first = rec[:, 6]
    life_ticks = (rec[:, 5] - first).astype(np.float64)
    ghz = float(np.median(life_ticks / np.maximum(life_real, 1e-9))) / 1e9
    us = lambda ticks: ticks / (ghz * 1e3)
    span_real = (rec[:, 4].max() - rec[:, 3].min()) * 10e-3
    print(f"\n== pass {p}: {len(rec)} workgroups, {int(rec[:, 2].sum())} tiles, launch span {span_real:.1f} us (s_memrealtime), s_memtime at {ghz:.2f} GHz ==")
    load, rounds, store, gap = [], [], [], []
    per_cu = defaultdict(list)
    for r, nt in zip(rec, n_t):
        hw, xcc = int(r[0]) & 0xFFFFFFFF, int(r[0] >> 32) & 0xF
        cu = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
        t = r[6 : 6 + 4 * nt].reshape(nt, 4)
        for j in range(nt):
            load.append(t[j, 1] - t[j, 0])
            rounds.append(t[j, 2] - t[j, 1])
            store.append(t[j, 3] - t[j, 2])
            if j + 1 < nt:
                gap.append(t[j + 1, 0] - t[j, 3])
            per_cu[cu].append(("L", t[j, 0], t[j, 1]))
            per_cu[cu].append(("R", t[j, 1], t[j, 2]))
            per_cu[cu].append(("S", t[j, 2], t[j, 3]))
    q = lambda v: "  ".join(f"{us(np.percentile(v, pc)):6.2f}" for pc in (10, 50, 90))
    print(f"per tile, first wave, us (10th / 50th / 90th percentile):")
    print(f"  loads issued -> back         {q(load)}     mean {us(np.mean(load)):6.2f}")
    print(f"  rounds (gates + exchanges)   {q(rounds)}     mean {us(np.mean(rounds)):6.2f}")
    print(f"  stores (+ D, sums) issued    {q(store)}     mean {us(np.mean(store)):6.2f}")
    tile_mean = np.mean(load) + np.mean(rounds) + np.mean(store)
    print(f"  a tile of one workgroup: {us(tile_mean):.2f} us: load wait {100 * np.mean(load) / tile_mean:.0f} %, rounds {100 * np.mean(rounds) / tile_mean:.0f} %, store issue {100 * np.mean(store) / tile_mean:.0f} %")
    # compute units: the union of the rounds of every workgroup that ran there, against the compute unit's busy span
    share_any, share_two, share_mem_only, wgs_per_cu = [], [], [], []
    for cu, ivs in per_cu.items():
        lo, hi = min(a for _, a, _ in ivs), max(b for _, _, b in ivs)
        span = hi - lo
        if span <= 0:
            continue
        r_iv = [(a, b) for k, a, b in ivs if k == "R"]
        m_iv = [(a, b) for k, a, b in ivs if k != "R"]
        any_r = covered(r_iv)
        share_any.append(any_r / span)
        share_two.append(overlap_at_least(r_iv, 2) / span)
        busy = covered([(a, b) for _, a, b in ivs])
        share_mem_only.append((busy - any_r) / span)
        wgs_per_cu.append(len(r_iv))
    print(f"compute units seen: {len(per_cu)}; of a compute unit's span (first load issued .. last store issued), mean over compute units:")
    print(f"  some workgroup in its rounds          {100 * np.mean(share_any):5.1f} %   (10th / 90th percentile {100 * np.percentile(share_any, 10):.0f} / {100 * np.percentile(share_any, 90):.0f})")
    print(f"  two or more in their rounds together  {100 * np.mean(share_two):5.1f} %")
    print(f"  none in its rounds (all in load wait or store issue) {100 * np.mean(share_mem_only):5.1f} %")
    print(f"  tiles per compute unit: {np.mean(wgs_per_cu):.1f}; workgroups with more than {TILES} tiles (their later tiles are not recorded): {int((rec[:, 2] > TILES).sum())}")


if __name__ == "__main__":
    main()
