"""CPU interpreter for the encoded pass plans libqsv's scheduler emits (test helper).

It executes a plan exactly the way ``pass_kernel`` in queasars_amd/csrc/kernels.hip does -- thread/register
layouts, XOR-column index maps, LDS exchanges, control predicates -- but with NumPy arrays standing in for
registers and LDS.  Running it against the oracle checks the scheduler and the plan encoding without a GPU;
it also verifies on the way that every index map is a bijection and reports LDS bank conflicts.
"""

from __future__ import annotations

import numpy as np

CIRCUIT_HEADER_WORDS = 8
HEADER_WORDS = 4
GATE_WORDS = 4
MAX_TILE_BITS, MAX_THREAD_BITS, MAX_REG_BITS, POS_PAD = 13, 9, 4, 62
MAX_COMPACT_BITS, MAX_OUTER_BITS, COMPACT_STORE, COMPACT_LOAD = 8, 20, 1, 2
MAX_SWAPS, SWAP_PAD, SWAP_LANE_LO, SWAP_LANE_HI = 4, 0xFFFFFFFF, 0, 6
GATE_GENERAL, GATE_NEGATED, MAX_CHAIN = 1 << 24, 1 << 25, 6

# lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS table)
_READ_G0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
_READ_G1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]


def _xor_columns(cols: np.ndarray, idx: np.ndarray) -> np.ndarray:
    out = np.zeros_like(idx, dtype=np.uint64)
    for u, col in enumerate(cols):
        out ^= np.where((idx >> np.uint64(u)) & np.uint64(1), np.uint64(col), np.uint64(0))
    return out


def _insert_zeros(b: int, positions) -> int:
    for p in positions:
        b = ((b >> p) << (p + 1)) | (b & ((1 << p) - 1))
    return b


def bank_conflicts_b128(thread_cols, is_write: bool) -> int:
    """Extra LDS cycles of one 16-byte-per-lane wave access whose lane->element map is given by columns."""
    lanes = np.arange(64, dtype=np.uint64)
    n_lane_bits = min(6, len(thread_cols))
    off = _xor_columns(np.asarray(thread_cols[:n_lane_bits], dtype=np.uint64), lanes)
    active = 1 << n_lane_bits
    extra = 0
    if is_write:
        groups = [list(range(g * 8, g * 8 + 8)) for g in range(8)]
        mod = 8
    else:
        groups = [_READ_G0, _READ_G1, [x + 32 for x in _READ_G0], [x + 32 for x in _READ_G1]]
        mod = 16
    for grp in groups:
        slots = [int(off[lane]) % mod for lane in grp if lane < active]
        if slots:
            extra += max(slots.count(s) for s in set(slots)) - 1
    return extra


def _angle_entry(w: np.ndarray, off: int, params) -> tuple[float, float, float]:
    idx = [int(np.int32(w[off + i])) for i in range(3)]
    lit = np.asarray(w[off + 3 : off + 9], dtype=np.uint32).view(np.float64)
    return tuple(float(params[i]) if i >= 0 else float(l) for i, l in zip(idx, lit))


def _u_matrix(theta: float, phi: float, lam: float) -> np.ndarray:
    c, s = np.cos(theta / 2), np.sin(theta / 2)
    return np.array([[c, -np.exp(1j * lam) * s], [np.exp(1j * phi) * s, np.exp(1j * (phi + lam)) * c]])


def _entry_matrix(w: np.ndarray, off: int, params) -> np.ndarray:
    """One angle-table entry -> matrix; codes below -1 are the fixed matrices of the virtual circuits (split.hpp)."""
    code = int(np.int32(w[off]))
    if code >= -1:
        return _u_matrix(*_angle_entry(w, off, params))
    return {-2: np.array([[1, 0], [0, 0]]), -3: np.array([[1, 0], [1, 0]]), -4: np.array([[0, 1], [1, 0]])}[code].astype(np.complex128)


def bank_conflicts_b64(thread_cols, is_write: bool) -> int:
    """Same for 8-byte-per-lane accesses: ds_write_b64 = 4 groups of 16 contiguous lanes over 16 8-B slots,
    ds_read_b64 = 2 groups of 32 lanes over 32 8-B slots (MI355X_MICROARCH.md, LDS table)."""
    lanes = np.arange(64, dtype=np.uint64)
    n_lane_bits = min(6, len(thread_cols))
    off = _xor_columns(np.asarray(thread_cols[:n_lane_bits], dtype=np.uint64), lanes)
    active = 1 << n_lane_bits
    size, mod = (16, 16) if is_write else (32, 32)
    extra = 0
    for g in range(64 // size):
        slots = [int(off[lane]) % mod for lane in range(g * size, (g + 1) * size) if lane < active]
        if slots:
            extra += max(slots.count(s) for s in set(slots)) - 1
    return extra


def decode(words: np.ndarray) -> dict:
    w = np.asarray(words, dtype=np.uint32)
    n_passes, n_real, n_qubits = int(w[0]), int(w[1]), int(w[2])
    angle_off, fold_off, n_fold = int(w[3]), int(w[4]), int(w[5])
    passes = []
    for p in range(n_passes):
        o = int(w[CIRCUIT_HEADER_WORDS + p])
        hdr = int(w[o])
        k, r, t, n_rounds = hdr & 0xFF, (hdr >> 8) & 0xFF, (hdr >> 16) & 0xFF, hdr >> 24
        first_gate = int(w[o + 1])
        # fixed-shape blocks (plan.hpp): positions padded to 13 with 62, columns as 9 thread + 4 register slots
        def block_cols(at):
            cols = [int(x) for x in w[at : at + MAX_THREAD_BITS + MAX_REG_BITS]]
            assert all(c == 0 for c in cols[t:MAX_THREAD_BITS] + cols[MAX_THREAD_BITS + r :]), "padding must be zero"
            return cols[:t] + cols[MAX_THREAD_BITS : MAX_THREAD_BITS + r]

        cur = o + HEADER_WORDS
        pos_block = [int(x) for x in w[cur : cur + MAX_TILE_BITS]]
        assert all(x == POS_PAD for x in pos_block[k:]), "position padding"
        pos = pos_block[:k]
        cur += MAX_TILE_BITS
        gl = block_cols(cur)
        cur += MAX_THREAD_BITS + MAX_REG_BITS
        gs = block_cols(cur)
        cur += MAX_THREAD_BITS + MAX_REG_BITS
        # compact block (plan.hpp): control-qubit positions; W / F columns of the load layout and of the tile number
        flags = int(w[o + 2])
        compact = {"store": bool(flags & COMPACT_STORE), "load": bool(flags & COMPACT_LOAD), "m": (flags >> 8) & 0xFF}
        compact["ctrl_pos"] = [int(x) for x in w[cur : cur + MAX_COMPACT_BITS]][: compact["m"]] if compact["store"] else []
        cur += MAX_COMPACT_BITS
        if compact["load"]:
            compact["wcols"] = block_cols(cur)
            compact["fcols"] = block_cols(cur + MAX_THREAD_BITS + MAX_REG_BITS)
        else:
            assert not np.any(w[cur : cur + 2 * (MAX_THREAD_BITS + MAX_REG_BITS) + 2 * MAX_OUTER_BITS]), "unused compact block"
        cur += 2 * (MAX_THREAD_BITS + MAX_REG_BITS)
        compact["wbase"] = [int(x) for x in w[cur : cur + MAX_OUTER_BITS]]
        cur += MAX_OUTER_BITS
        compact["fbase"] = [int(x) for x in w[cur : cur + MAX_OUTER_BITS]]
        cur += MAX_OUTER_BITS
        rounds = []
        sched = first_gate
        for _ in range(n_rounds):
            rh = int(w[cur])
            cur += 1
            n_gates, exch, intra, swap = rh & 0xFFFF, (rh >> 16) & 1, (rh >> 17) & 1, (rh >> 18) & 1
            assert not (intra and not exch) and not (swap and exch)
            wc = rc = None
            swaps = []
            if swap:
                # relayout by lane swaps (plan.hpp): register bit v trades its tile bit with lane bit u, no LDS
                for x in w[cur : cur + MAX_SWAPS]:
                    if int(x) != SWAP_PAD:
                        swaps.append((int(x) & 0xFF, (int(x) >> 8) & 0xFF))
                assert swaps and all(int(x) == SWAP_PAD for x in w[cur + len(swaps) : cur + MAX_SWAPS]), "swap padding"
                assert all(v < r and SWAP_LANE_LO <= u < min(SWAP_LANE_HI, t) for v, u in swaps)
                cur += MAX_SWAPS
            if exch:
                wc = block_cols(cur)
                cur += MAX_THREAD_BITS + MAX_REG_BITS
                rc = block_cols(cur)
                cur += MAX_THREAD_BITS + MAX_REG_BITS
            gates = []
            for _g in range(n_gates):
                w0, ct, cg, ncg = (int(x) for x in w[cur : cur + GATE_WORDS])
                cur += GATE_WORDS
                creg = (w0 >> 8) & 0xFF
                assert (w0 >> 16) & 0xFF00 & ~((GATE_NEGATED | GATE_GENERAL) >> 16) == 0, "unknown gate flags"
                assert not (cg & ncg) and ct < (1 << (2 * MAX_THREAD_BITS))
                gates.append({"tbit": w0 & 0xFF, "creg": None if creg == 0xFF else creg, "pairs": (w0 >> 16) & 0xFF, "ct": ct, "cg": cg,
                              "ncg": ncg, "negated": bool(w0 & GATE_NEGATED), "general": bool(w0 & GATE_GENERAL), "sched": sched})
                sched += 1
            rounds.append({"write_cols": wc, "read_cols": rc, "gates": gates, "intra_wave": bool(intra), "swaps": swaps})
        passes.append({"k": k, "r": r, "t": t, "pos": pos, "load_cols": gl, "store_cols": gs, "rounds": rounds,
                       "first_gate": first_gate, "compact": compact})
    fold_index = [(int(w[fold_off + 2 * q]), int(w[fold_off + 2 * q + 1])) for q in range(n_qubits)]
    # chain index (plan.hpp): per scheduled entry its factors in the angle table
    chain_off, n_factors = int(w[6]), int(w[7])
    chains = [(int(w[chain_off + s]) & 0xFFFFFF, int(w[chain_off + s]) >> 24) for s in range(n_real)]
    at = 0
    for first, count in chains:
        assert first == at and 1 <= count <= MAX_CHAIN, "chain index must tile the factors in schedule order"
        at += count
    assert at == n_factors
    assert all(n_factors <= first and first + count <= n_factors + n_fold for first, count in fold_index if count)
    return {
        "n_passes": n_passes, "n_real": n_real, "n_qubits": n_qubits, "angle_off": angle_off, "n_fold": n_fold,
        "fold_index": fold_index, "passes": passes, "words": w, "chains": chains, "n_factors": n_factors,
    }


def prepare(plan: dict, params) -> tuple[np.ndarray, np.ndarray]:
    """What prepare_kernel computes: matrices of the scheduled gates and the initial product-state factors."""
    w, off = plan["words"], plan["angle_off"]
    mats = []
    for first, count in plan["chains"]:
        m = np.eye(2, dtype=np.complex128)
        for i in range(count):  # the factor that acts first rightmost
            m = _entry_matrix(w, off + 9 * (first + i), params) @ m
        mats.append(m)
    mats = np.array(mats).reshape(-1, 2, 2)
    vecs = np.zeros((plan["n_qubits"], 2), dtype=np.complex128)
    for q, (first, count) in enumerate(plan["fold_index"]):
        v = np.array([1.0 + 0j, 0.0 + 0j])
        for i in range(count):
            v = _entry_matrix(w, off + 9 * (first + i), params) @ v
        vecs[q] = v
    return mats, vecs


def run(words: np.ndarray, n_qubits: int, params, stats: dict | None = None, lds_access_bytes: int = 8) -> np.ndarray:
    """Execute the plan from |0..0> for the given flat parameter list, the way prepare_kernel + pass_kernel do."""
    plan = decode(words)
    assert plan["n_qubits"] == n_qubits
    mats, vecs = prepare(plan, params)
    dim = 1 << n_qubits
    state = np.zeros(dim, dtype=np.complex128)
    all_gates = [g for ps in plan["passes"] for rd in ps["rounds"] for g in rd["gates"]]
    assert [g["sched"] for g in all_gates] == list(range(plan["n_real"])), "schedule order must be contiguous"
    if stats is not None:
        stats.update({"passes": plan["n_passes"], "rounds": 0, "exchanges": 0, "swap_rounds": 0, "swaps": 0,
                      "gates": sum(1 for g in all_gates if not g["negated"]),
                      "folded": plan["n_fold"], "conflicts": 0, "wave_uniform_ctrl": 0, "lane_ctrl": 0})
    for pi, ps in enumerate(plan["passes"]):
        k, r, t = ps["k"], ps["r"], ps["t"]
        assert k == t + r and k <= n_qubits
        assert ps["pos"] == sorted(set(ps["pos"])) and all(0 <= q < n_qubits for q in ps["pos"])
        n_thr, n_reg = 1 << t, 1 << r
        tid = np.arange(n_thr, dtype=np.uint64)[:, None]
        reg = np.arange(n_reg, dtype=np.uint64)[None, :]

        def index_map(cols):
            return _xor_columns(np.asarray(cols[:t], dtype=np.uint64), tid) ^ _xor_columns(
                np.asarray(cols[t:], dtype=np.uint64), reg
            )

        g_load, g_store = index_map(ps["load_cols"]), index_map(ps["store_cols"])
        tile_mask = sum(1 << q for q in ps["pos"])
        cpt = ps["compact"]
        assert not (cpt["store"] and pi != 0) and not (cpt["load"] and pi != 1)
        for gmap, local in ((g_load, False), (g_store, cpt["store"])):
            flat = np.sort(gmap.reshape(-1))
            assert len(np.unique(flat)) == n_thr * n_reg, "global index map is not injective"
            if local:  # a compact pass 0 stores at offsets inside the pattern's own tile
                assert flat.max() < (1 << k)
            else:
                assert np.all((flat & ~np.uint64(tile_mask)) == 0), "global offsets leave the tile"
        lds_maps = []
        for rd in ps["rounds"]:
            if rd["write_cols"] is None:
                lds_maps.append(None)
                continue
            wmap, rmap = index_map(rd["write_cols"]), index_map(rd["read_cols"])
            for m in (wmap, rmap):
                flat = m.reshape(-1)
                assert flat.max() < (1 << k) and len(np.unique(flat)) == n_thr * n_reg, "LDS map is not a bijection"
            lds_maps.append((wmap, rmap))
            if rd["intra_wave"]:
                # the kernel runs this exchange without barriers: every wave must read back exactly the elements it
                # wrote, and (so that consecutive barrier-free exchanges cannot collide either) its elements must be
                # the ones whose address bits at the wave-held tile-bit positions spell the wave's own index
                waves = max(1, n_thr // 64)
                for wv in range(waves):
                    rows = slice(wv * 64, min((wv + 1) * 64, n_thr))
                    wrote, reads = set(wmap[rows].reshape(-1).tolist()), set(rmap[rows].reshape(-1).tolist())
                    assert wrote == reads, "an intra-wave exchange reads another wave's elements"
                if waves > 1:
                    wave_cols_w, wave_cols_r = rd["write_cols"][6:t], rd["read_cols"][6:t]
                    assert wave_cols_w == wave_cols_r, "wave-index bits moved in an intra-wave exchange"
                    assert all(c & (c - 1) == 0 for c in wave_cols_w), "wave columns must be pure address bits"
                    wave_mask = sum(wave_cols_w)
                    lane_reg = rd["write_cols"][:6] + rd["write_cols"][t:] + rd["read_cols"][:6] + rd["read_cols"][t:]
                    assert all((c & wave_mask) == 0 for c in lane_reg), "a swizzle term touches a wave address bit"
            if stats is not None:
                stats["exchanges"] += 1
                stats["intra_wave_exchanges"] = stats.get("intra_wave_exchanges", 0) + int(rd["intra_wave"])
                model = bank_conflicts_b128 if lds_access_bytes == 16 else bank_conflicts_b64
                stats["conflicts"] += model(rd["write_cols"][:t], True)
                stats["conflicts"] += model(rd["read_cols"][:t], False)
        if stats is not None:
            stats["swap_rounds"] += sum(1 for rd in ps["rounds"] if rd["swaps"])
            stats["swaps"] += sum(len(rd["swaps"]) for rd in ps["rounds"])
            stats["rounds"] += len(ps["rounds"])
            for rd in ps["rounds"]:
                for g in rd["gates"]:
                    if g["ct"] and not g["negated"]:  # (a multiplexed gate is counted once, by its control-is-1 entry)
                        stats["wave_uniform_ctrl" if (g["ct"] & 63) == 0 else "lane_ctrl"] += 1

        new_state = state.copy()
        n_tiles = 1 << (n_qubits - k)
        if cpt["store"]:
            # COMPACT pass 0 (plan.hpp): one tile per pattern of the outer control qubits, no tile factor
            assert len(cpt["ctrl_pos"]) == cpt["m"] and all(q not in ps["pos"] for q in cpt["ctrl_pos"])
            n_tiles = 1 << cpt["m"]
            new_state = np.zeros(dim, dtype=np.complex128)  # the slot now holds the table W (n_tiles * 2^k entries)
            tile_qubits0, ctrl0 = list(ps["pos"]), list(cpt["ctrl_pos"])
        if cpt["load"]:
            # what prepare_kernel's tile-factor table holds: product of the outer qubits' factors per pass-0 tile
            outer0 = [q for q in range(n_qubits) if q not in tile_qubits0]
            factor = np.ones(1 << len(outer0), dtype=np.complex128)
            for j, q in enumerate(outer0):
                factor = factor * np.where((np.arange(len(factor)) >> j) & 1, vecs[q, 1], vecs[q, 0])
            w_load, f_load = index_map(cpt["wcols"]), index_map(cpt["fcols"])
        for b in range(n_tiles):
            if cpt["store"]:
                base = sum(((b >> j) & 1) << q for j, q in enumerate(cpt["ctrl_pos"]))
            else:
                base = _insert_zeros(b, ps["pos"])
            if pi == 0:
                gidx = (np.uint64(base) + g_load).astype(np.int64)
                amp = np.ones(gidx.shape, dtype=np.complex128)
                for q in range(n_qubits):
                    if cpt["store"] and q not in ps["pos"]:
                        continue
                    amp = amp * np.where((gidx >> q) & 1, vecs[q, 1], vecs[q, 0])
            elif cpt["load"]:
                wbase = fbase = 0
                for j in range(MAX_OUTER_BITS):
                    if (b >> j) & 1:
                        wbase ^= cpt["wbase"][j]
                        fbase ^= cpt["fbase"][j]
                widx = (np.uint64(wbase) ^ w_load).astype(np.int64)
                fidx = (np.uint64(fbase) ^ f_load).astype(np.int64)
                amp = state[widx] * factor[fidx]
                # cross-check the plan's columns against the definition: psi[i] = F[o(i)] * W[x(i)][t(i)]
                gidx = (np.uint64(base) + g_load).astype(np.int64)
                t_of = sum((((gidx >> q) & 1) << j) for j, q in enumerate(tile_qubits0))
                x_of = sum((((gidx >> q) & 1) << j) for j, q in enumerate(ctrl0))
                o_of = sum((((gidx >> q) & 1) << j) for j, q in enumerate(outer0))
                assert np.array_equal(widx, (x_of << len(tile_qubits0)) | t_of) and np.array_equal(fidx, o_of)
            else:
                amp = state[(np.uint64(base) + g_load).astype(np.int64)]
            for rd, maps in zip(ps["rounds"], lds_maps):
                if maps is not None:
                    wmap, rmap = maps
                    lds = np.empty(1 << k, dtype=np.complex128)
                    lds[wmap.astype(np.int64)] = amp
                    amp = lds[rmap.astype(np.int64)]
                for v, u in rd["swaps"]:
                    # element (thread, e) with thread bit u != register bit v of e trades places with
                    # (thread ^ 2^u, e ^ 2^v): what v_permlane*_swap / the DPP row shifts do in the kernel
                    tt, ee = np.arange(n_thr)[:, None], np.arange(n_reg)[None, :]
                    differ = ((tt >> u) & 1) != ((ee >> v) & 1)
                    amp = np.where(differ, amp[tt ^ (1 << u), ee ^ (1 << v)], amp)
                for g in rd["gates"]:
                    # (the control-is-0 entry of a multiplexed gate lists its control among the complemented bits)
                    if (base & g["cg"]) != g["cg"] or (~base & g["ncg"]) != g["ncg"]:
                        continue
                    m = mats[g["sched"]]
                    assert g["general"] or abs(m[0, 0].imag) == 0.0, "a plain entry's m00 must be real"
                    assert g["negated"] == bool(g["ncg"] or g["ct"] >> MAX_THREAD_BITS) or g["creg"] is not None
                    bit = 1 << g["tbit"]
                    cbit = 0 if g["creg"] is None else 1 << g["creg"]
                    want = 0 if g["negated"] else cbit
                    tid_ext = np.arange(n_thr) | ((~np.arange(n_thr) & ((1 << MAX_THREAD_BITS) - 1)) << MAX_THREAD_BITS)
                    lane_on = (tid_ext & g["ct"]) == g["ct"]
                    pair = -1
                    for e0 in range(n_reg):
                        if e0 & bit:
                            continue
                        pair += 1  # the kernel's assembly gate loop goes by the pair mask: it must say the same
                        assert ((g["pairs"] >> pair) & 1) == int((e0 & cbit) == want), "pair mask disagrees with creg"
                        if (e0 & cbit) != want:
                            continue
                        a0, a1 = amp[:, e0].copy(), amp[:, e0 | bit].copy()
                        amp[:, e0] = np.where(lane_on, m[0, 0] * a0 + m[0, 1] * a1, a0)
                        amp[:, e0 | bit] = np.where(lane_on, m[1, 0] * a0 + m[1, 1] * a1, a1)
            if cpt["store"]:
                new_state[(b << k) + g_store.astype(np.int64)] = amp
            else:
                new_state[(np.uint64(base) + g_store).astype(np.int64)] = amp
        state = new_state
    return state
