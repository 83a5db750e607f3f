"""Pass scheduler checked on the CPU: plans from qsv_plan_build() are executed by tests/plan_interpreter.py
(a NumPy model of the kernel) and compared with the oracle."""

import numpy as np
import pytest

import helpers
import plan_interpreter as pi
from queasars_amd.ir import CircuitIR, ParamRef
from queasars_amd.planning import build_plan_words

GEOMETRIES = [
    {},
    dict(tile_bits=10, reg_bits=3, low_bits=3),
    dict(tile_bits=11, reg_bits=4, low_bits=4),
    dict(tile_bits=12, reg_bits=4, low_bits=2),
    dict(tile_bits=13, reg_bits=4, low_bits=2),
    dict(tile_bits=9, reg_bits=2, low_bits=2),
    dict(tile_bits=8, reg_bits=1, low_bits=1),
]


@pytest.mark.parametrize("n_qubits,n_layers", [(1, 2), (2, 3), (3, 2), (5, 4), (7, 2), (8, 3), (10, 3), (12, 4), (13, 3), (14, 3)])
def test_default_plans_reproduce_the_circuit(n_qubits, n_layers):
    _, circuits, params = helpers.population_circuits(n_qubits, n_layers, 3, seed=7)
    for c, p in zip(circuits, params):
        stats = {}
        got = pi.run(build_plan_words(c), n_qubits, p, stats)
        assert np.abs(got - helpers.oracle_state(c, p)).max() < 1e-13
        assert stats["conflicts"] == 0, "LDS exchanges must be bank-conflict free"


@pytest.mark.parametrize("cfg", GEOMETRIES[1:])
@pytest.mark.parametrize("exchange", [1, 3])
def test_other_geometries(cfg, exchange):
    n_qubits = 13
    _, circuits, params = helpers.population_circuits(n_qubits, 3, 2, seed=3)
    for c, p in zip(circuits, params):
        stats = {}
        got = pi.run(build_plan_words(c, exchange=exchange, **cfg), n_qubits, p, stats, lds_access_bytes=16 if exchange == 1 else 8)
        assert np.abs(got - helpers.oracle_state(c, p)).max() < 1e-13
        assert stats["conflicts"] == 0


def test_partially_parameterised_circuit_and_literals():
    pop, _, _ = helpers.population_circuits(9, 3, 2, seed=1)
    for ind in pop.individuals:
        for layer in range(3):
            c = ind.get_partially_parameterized_quantum_circuit({layer})
            vals = list(ind.get_layer_parameter_values(layer))
            got = pi.run(build_plan_words(c), 9, vals)
            assert np.abs(got - helpers.oracle_state(c, vals)).max() < 1e-13


def test_folding_rules():
    """u gates before any entangling gate fold into the product state; cu3 with an untouched control is dropped."""
    n = 4
    c = CircuitIR(n)
    c.u(0.3, 0.2, 0.1, 0).u(1.0, 0.5, 0.25, 0)          # both fold into qubit 0
    c.cu3(0.7, 0.1, 0.2, 2, 1)                            # control 2 still |0>: identity, dropped
    c.cu3(0.9, 0.3, 0.4, 0, 1)                            # real: control 0 is in superposition
    c.u(0.5, 0.6, 0.7, 0)                                  # qubit 0 is entangled now: real
    c.u(ParamRef(0), ParamRef(1), ParamRef(2), 3)          # untouched qubit: folds even this late
    c.id(2)
    params = [0.11, 0.22, 0.33]
    plan = pi.decode(build_plan_words(c))
    assert plan["n_real"] == 2 and plan["n_fold"] == 3
    assert [cnt for _, cnt in plan["fold_index"]] == [2, 0, 0, 1]
    got = pi.run(build_plan_words(c), n, params)
    assert np.abs(got - helpers.oracle_state(c, params)).max() < 1e-14


def test_empty_and_identity_only_circuits():
    for c in (CircuitIR(3), CircuitIR(3).id(0).id(2)):
        got = pi.run(build_plan_words(c), 3, [])
        assert got[0] == 1 and np.count_nonzero(got) == 1


def test_every_gate_scheduled_once_in_dependency_order():
    n = 14
    _, circuits, params = helpers.population_circuits(n, 4, 3, seed=11)
    for c in circuits:
        plan = pi.decode(build_plan_words(c, tile_bits=10, reg_bits=3, low_bits=3))
        ops = c.packed()
        # every gate of an EVQE circuit has its own parameters: the index of its theta names the op
        op_of_theta = {int(o["p_theta"]): i for i, o in enumerate(ops) if o["kind"] != 0}
        assert len(op_of_theta) == sum(1 for o in ops if o["kind"] != 0)
        w, off = plan["words"], plan["angle_off"]
        entries = [g for ps in plan["passes"] for rd in ps["rounds"] for g in rd["gates"]]
        # where each op acts: (scheduled entry, place in the entry's chain); the control-is-0 entry of a multiplexed gate
        # repeats the u gates of the entry before it and nothing else
        seen_at = {}
        for g, (first, count) in zip(entries, plan["chains"]):
            chain = [op_of_theta[int(np.int32(w[off + 9 * (first + i)]))] for i in range(count)]
            if g["negated"]:
                before = [op for op, (e, _) in seen_at.items() if e == g["sched"] - 1]
                assert sorted(chain) == sorted(op for op in before if ops[op]["kind"] == 1) and chain
                continue
            for i, op in enumerate(chain):
                assert op not in seen_at
                seen_at[op] = (g["sched"], i)
        folded = {op_of_theta[int(np.int32(w[off + 9 * (first + i)]))] for first, count in plan["fold_index"] for i in range(count)}
        assert not folded & set(seen_at)
        # (the rest are cu3 gates whose control is still |0>: dropped)
        assert all(ops[i]["kind"] == 2 for i in set(op_of_theta.values()) - folded - set(seen_at))
        order = sorted(seen_at)
        # two real gates that touch a common qubit (other than as shared control) keep their program order
        for i in order:
            for j in order:
                if i >= j:
                    continue
                a, b = ops[i], ops[j]
                qa = {int(a["target"])} | ({int(a["control"])} if a["kind"] == 2 else set())
                qb = {int(b["target"])} | ({int(b["control"])} if b["kind"] == 2 else set())
                shared = qa & qb
                commuting_share = shared and all(
                    q != a["target"] and q != b["target"] for q in shared
                )
                if shared and not commuting_share:
                    assert seen_at[i] < seen_at[j]


def test_plan_build_rejects_bad_input():
    c = CircuitIR(4).u(0.1, 0.2, 0.3, 1)
    with pytest.raises(ValueError):
        build_plan_words(c, tile_bits=3, reg_bits=4)
    with pytest.raises(ValueError):
        build_plan_words(c, reg_bits=7)


def test_generated_gate_loop_is_current():
    """gate_loop_gen.inc is committed next to its generator: the two must not drift apart."""
    import subprocess
    import sys
    from pathlib import Path

    gen = Path(__file__).resolve().parent.parent / "queasars_amd" / "csrc" / "gen_gate_loop.py"
    assert subprocess.run([sys.executable, str(gen), "--check"]).returncode == 0, "run gen_gate_loop.py"


def _family_stats(n_qubits=14, count=6):
    _, circuits, params = helpers.population_circuits(n_qubits, 4, count, seed=0)
    tot = dict(exchanges=0, intra=0, lane=0, gates=0, swap_rounds=0, swaps=0)
    for c, p in zip(circuits, params):
        stats = {}
        got = pi.run(build_plan_words(c), n_qubits, p, stats)
        assert np.abs(got - helpers.oracle_state(c, p)).max() < 1e-13
        assert stats["conflicts"] == 0
        tot["exchanges"] += stats["exchanges"]
        tot["intra"] += stats.get("intra_wave_exchanges", 0)
        tot["lane"] += stats["lane_ctrl"]
        tot["gates"] += stats["gates"]
        tot["swap_rounds"] += stats["swap_rounds"]
        tot["swaps"] += stats["swaps"]
    return tot


def test_most_relayouts_are_lane_swaps():
    """Scheduler quality on the benchmark family: relayouts whose targets sit in registers or on lane bits run as
    in-register lane swaps (no LDS, no barrier); LDS exchanges remain only for targets on wave-index bits."""
    tot = _family_stats()
    assert tot["swap_rounds"] > 0 and tot["swaps"] >= tot["swap_rounds"]
    assert tot["exchanges"] < tot["swap_rounds"]
    # (of the gates left after fusion, which takes most u gates away: a third; a quarter of the 119 gates before it)
    assert tot["lane"] <= 0.33 * tot["gates"]


def test_without_swaps_some_exchanges_stay_inside_a_wave(monkeypatch):
    """QSV_SWAPS=0 (measurement knob): every relayout is an LDS exchange again, some of them barrier-free, and few
    controls sit on lane bits."""
    monkeypatch.setenv("QSV_SWAPS", "0")
    tot = _family_stats()
    assert tot["swap_rounds"] == 0 and tot["intra"] > 0 and tot["intra"] <= tot["exchanges"]
    assert tot["lane"] <= 0.20 * tot["gates"]  # (of the gates left after fusion)


def test_swap_rounds_bring_the_low_tile_bits_home():
    """A gate on qubit 0 or 1 takes an always-resident low tile bit off its lane; the last layout of every pass must have
    it back (the global index maps are checked to be bijections inside the tile by the interpreter, and the state must
    match): exercise it with circuits that hammer the low qubits, in several geometries."""
    n = 14
    for cfg in ({}, dict(tile_bits=10, reg_bits=3, low_bits=3), dict(tile_bits=9, reg_bits=2, low_bits=2)):
        c = CircuitIR(n)
        rng = np.random.default_rng(5)
        for layer in range(6):
            for q in (0, 1, 2, int(rng.integers(3, n))):
                c.u(*rng.uniform(0, 6, 3), q)
            c.cu3(*rng.uniform(0, 6, 3), int(rng.integers(2, n)), int(rng.integers(0, 2)))
            c.cu3(*rng.uniform(0, 6, 3), int(rng.integers(0, 2)), int(rng.integers(2, n)))
        stats = {}
        words = build_plan_words(c, **cfg)
        assert np.abs(pi.run(words, n, [], stats) - helpers.oracle_state(c, [])).max() < 1e-13
        assert stats["swaps"] > 0
        for ps in pi.decode(words)["passes"]:
            cl = min(2, ps["t"])
            assert ps["store_cols"][:cl] == ps["load_cols"][:cl], "low tile bits must end on the lanes they started on"


def test_the_tile_search_keeps_its_pass_counts():
    """A pass is a full sweep of the state: what the scheduler's local search over the tiles buys (round 4) is held as a bound
    on the mean number of passes of fixed populations -- first-come tiles gave 3.94 (n = 24, eight layers), 2.44 (n = 20, six),
    3.09 (n = 20, eight); 256 random attempts 3.5 / 2.16 / 3.0 -- and every plan still reproduces its circuit at a size the
    interpreter can run."""
    for n, layers, count, tile, reg, bound in ((24, 8, 8, 13, 4, 3.2), (20, 6, 16, 12, 4, 2.2), (20, 8, 16, 12, 4, 3.05)):
        _, circuits, _ = helpers.population_circuits(n, layers, count, seed=0)
        passes = [pi.decode(build_plan_words(c, tile_bits=tile, reg_bits=reg))["n_passes"] for c in circuits]
        assert sum(passes) / count <= bound, (n, layers, sum(passes) / count)
    # (executed: registers of 14 - 16 qubits with 8- and 9-qubit tiles need the search as a 24-qubit one with 13-qubit tiles does;
    # QSV_RETRIES=0 is the first-come rule alone)
    import os

    improved = 0
    for n, layers, tile, reg in ((15, 8, 8, 3), (14, 8, 8, 2), (16, 8, 9, 3)):
        _, circuits, params = helpers.population_circuits(n, layers, 4, seed=5)
        for c, p in zip(circuits, params):
            os.environ["QSV_RETRIES"] = "0"
            try:
                plain = pi.decode(build_plan_words(c, tile_bits=tile, reg_bits=reg))["n_passes"]
            finally:
                del os.environ["QSV_RETRIES"]
            words = build_plan_words(c, tile_bits=tile, reg_bits=reg)
            searched = pi.decode(words)["n_passes"]
            assert searched <= plain
            improved += searched < plain
            assert np.abs(pi.run(words, n, p) - helpers.oracle_state(c, p)).max() < 1e-13
    assert improved >= 6, improved


def test_three_low_lane_bits_restored_in_several_swap_rounds(monkeypatch):
    """With three low tile bits pinned to lanes (64-byte runs of 8-byte single-precision amplitudes) bringing them home can take
    six transpositions: more than one swap round holds.  The scheduler then emits several gate-less rounds (round 4; it used
    to refuse the circuit), each with the layout its own swaps leave."""
    monkeypatch.setenv("QSV_LANE_BITS", "3")
    n = 15
    seen_long = False
    for cfg in (dict(tile_bits=13, reg_bits=4, low_bits=3), dict(tile_bits=12, reg_bits=3, low_bits=3), dict(tile_bits=10, reg_bits=3, low_bits=3)):
        rng = np.random.default_rng(8)
        for trial in range(6):
            c = CircuitIR(n)
            for layer in range(5):
                for q in (0, 1, 2, int(rng.integers(3, n))):
                    c.u(*rng.uniform(0, 6, 3), q)
                c.cu3(*rng.uniform(0, 6, 3), int(rng.integers(3, n)), int(rng.integers(0, 3)))
                c.cu3(*rng.uniform(0, 6, 3), int(rng.integers(0, 3)), int(rng.integers(3, n)))
            words = build_plan_words(c, **cfg)
            assert np.abs(pi.run(words, n, []) - helpers.oracle_state(c, [])).max() < 1e-13
            for ps in pi.decode(words)["passes"]:
                cl = min(3, ps["t"])
                assert ps["store_cols"][:cl] == ps["load_cols"][:cl]
                tail = 0
                for rd in reversed(ps["rounds"]):
                    if rd["gates"] or not rd["swaps"]:
                        break
                    tail += 1
                seen_long |= tail >= 2
    assert seen_long, "no pass needed more than one restoring round: the case this test is for did not occur"


@pytest.mark.parametrize("n_qubits,cfg", [(14, dict(tile_bits=8, reg_bits=2, low_bits=2)), (13, dict(tile_bits=7, reg_bits=2, low_bits=2)),
                                           (16, dict())])
def test_compact_first_pass_reproduces_the_circuit(n_qubits, cfg):
    """COMPACT (plan.hpp): pass 0 over the patterns of its outer control qubits, pass 1 reading W x F.  The interpreter
    executes the encoded plan literally and cross-checks the W / F index columns against their definition."""
    _, circuits, params = helpers.population_circuits(n_qubits, 3, 6, seed=3)
    seen = 0
    for c, p in zip(circuits, params):
        words = build_plan_words(c, **cfg)
        first = pi.decode(words)["passes"][0]["compact"]
        seen += int(first["store"])
        assert np.abs(pi.run(words, n_qubits, p) - helpers.oracle_state(c, p)).max() < 1e-13
    if cfg:
        assert seen >= 3, "small tiles leave several outer qubits: most plans should take the compact first pass"


def test_multiplexed_gates_in_the_plan(monkeypatch):
    """Fusion (plan.hpp): on the benchmark family most u gates that act are multiplied into a neighbouring cu3 on the same
    target -- entries with a general matrix, entries with negated predicates, fewer matrices than gates -- and the plan's
    state is the oracle's with it and without it (QSV_FUSE=0)."""
    n = 14
    _, circuits, params = helpers.population_circuits(n, 6, 4, seed=5)
    for c, p in zip(circuits, params):
        want = helpers.oracle_state(c, p)
        fused = pi.decode(build_plan_words(c))
        assert np.abs(pi.run(fused["words"], n, p) - want).max() < 1e-13
        monkeypatch.setenv("QSV_FUSE", "0")
        plain = pi.decode(build_plan_words(c))
        assert np.abs(pi.run(plain["words"], n, p) - want).max() < 1e-13
        monkeypatch.delenv("QSV_FUSE")
        entries = [g for ps in fused["passes"] for rd in ps["rounds"] for g in rd["gates"]]
        assert any(g["negated"] for g in entries) and any(g["general"] for g in entries)
        assert not any(g["negated"] or g["general"] for ps in plain["passes"] for rd in ps["rounds"] for g in rd["gates"])
        # every negated entry follows its control-is-1 twin: same target register, complementary pairs or predicates
        for a, b in zip(entries, entries[1:]):
            if b["negated"]:
                assert not a["negated"] and a["tbit"] == b["tbit"] and a["creg"] == b["creg"]
                if a["creg"] is not None:
                    assert a["pairs"] ^ b["pairs"] == (1 << (1 << (fused["passes"][0]["r"] - 1))) - 1
        # a multiplexed gate is one gate of the circuit fewer (the u gates it took in), and two entries
        units = sum(1 for g in entries if not g["negated"])
        assert units < plain["n_real"] and plain["n_factors"] == fused["n_factors"] - sum(
            cnt for g, (_, cnt) in zip(entries, fused["chains"]) if g["negated"])
