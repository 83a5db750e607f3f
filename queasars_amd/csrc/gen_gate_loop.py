#!/usr/bin/env python3
"""Generates gate_loop_gen.inc: the fp64 gate loop of pass_kernel as one gfx950 assembly block per register width.

Why assembly: the loop applies a run-time sequence of 2x2 butterflies to 2^R amplitudes that must stay in the SAME
vector registers from one gate to the next.  Written in C++ (a switch over target bit / control bit inside the gate
loop) hipcc merges the cases through PHI nodes it cannot coalesce and moves the whole register file of amplitudes
around every gate: 40-60 v_mov_b64 per gate next to the 56 useful fp64 operations, and a 64-bit move costs a
v_fma_f64's issue time on gfx950.  Here every butterfly writes its results in place and the only branches are scalar.

Per gate entry (descriptor = 4 words w0, ct, cg, ncg; matrix = 8 doubles m00 m01 m10 m11 as (re, im); plan.hpp):
    skip unless the listed global index bits of this tile are set / clear   ((base, ~base) & (cg, ncg)) == (cg, ncg)   scalar, 64-bit
    mask lanes by the extended thread index (tid, ~tid): the listed bits set     (tid_ext & ct) == ct                  exec
        (a control-is-0 entry of a multiplexed gate lists its control among the complemented bits: no extra instruction)
    J = w0 & 0xff picks the target register bit; for each of the 2^(R-1) amplitude pairs p, bit 16 + p of w0 says
    whether the pair takes part (a register-held control switches half of them off)                 scalar
    pair update, in place, as four chains issued round-robin (p, q, u, w) -- 14 fp64 operations for a u-type matrix (Im m00 = 0):
        a1r' = ((m10r a0r - m10i a0i) + m11r a1r) - m11i a1i      a1i' = ((m10r a0i + m10i a0r) + m11i a1r) + m11r a1i
        a0r' = (m00 a0r + m01r a1r) - m01i a1i                    a0i' = (m00 a0i + m01r a1i) + m01i a1r
    16 for a product of matrices (bit 24 of w0: the two terms with Im m00 as well), a second copy of the body

Descriptor and matrix of gate g + 1 are fetched with scalar loads while gate g runs (two register sets, loop unrolled
by two).  Scalar registers are hard-coded and declared as clobbers; amplitudes and temporaries are operands.

    python gen_gate_loop.py            # rewrites gate_loop_gen.inc next to this file
"""

from __future__ import annotations

import sys
from pathlib import Path

OUT = Path(__file__).resolve().parent / "gate_loop_gen.inc"

# Timing experiments (scripts/ablate.py): QSV_GEN_ABL=name[,name..] leaves parts of the block out.  The results of such
# a build are wrong by construction; the shipped gate_loop_gen.inc is generated with the variable unset.
#   gatevalu  the 14 operations of a pair update          pairtest  the per-pair "takes part" test
#   swapvalu  the moves of a lane swap                     gateloop  the whole gate loop (descriptor fetches too)
import os

ABL = set(filter(None, os.environ.get("QSV_GEN_ABL", "").split(",")))

# Amplitude e of a thread lives in FOUR FIXED vector registers, v[AMP0 + 4e .. AMP0 + 4e + 3] = (re lo, re hi, im lo,
# im hi), in every assembly block (explicit "{v[a:b]}" constraints): a global_load/store_dwordx4 or an LDS access then
# reads or writes an amplitude where it lives -- left to the register allocator, real and imaginary parts ended up in
# separate pairs and every load, store and exchange was followed by 32 moves -- and the lane swaps can name the
# 32-bit halves directly.
AMP0 = 8


def amp_re(e: int) -> str:
    return f"v[{AMP0 + 4 * e}:{AMP0 + 4 * e + 1}]"


def amp_im(e: int) -> str:
    return f"v[{AMP0 + 4 * e + 2}:{AMP0 + 4 * e + 3}]"


def amp_operands(nr: int) -> list[str]:
    outs = []
    for e in range(nr):
        outs.append(f'"+{{{amp_re(e)}}}"(amp[{e}].re)')
        outs.append(f'"+{{{amp_im(e)}}}"(amp[{e}].im)')
    return outs


# hard-coded scalar registers (all inside one clobbered window)
MAT = {"A": 40, "B": 56}  # 16 SGPRs each: 8 doubles
DESC = {"A": 72, "B": 76}  # w0, ct, cg, op
SAVE = "s[80:81]"
T0 = "s82"
TP = "s[82:83]"    # (T0 and the round loop's T1 as a pair: the gate code's 64-bit temporary)
BASEP = "s[96:97]"  # (base, ~base) of the tile, set up at the head of each block
RP, MP, N = 84, 86, 88
CLOBBER_RANGE = list(range(40, 89)) + [96, 97]


def sreg2(base: int) -> str:
    return f"s[{base}:{base + 1}]"


def gate(lines: list[str], r: int, x: str, tag: str) -> None:
    """Emit the code of one gate that reads register set x ('A' or 'B')."""
    m, d = MAT[x], DESC[x]
    w0, ct, cg = f"s{d}", f"s{d + 1}", f"s{d + 2}"
    cgpair = sreg2(d + 2)  # (cg, ncg): global index bits that must be set / clear
    m00, m00i, m01r, m01i = sreg2(m), sreg2(m + 2), sreg2(m + 4), sreg2(m + 6)
    m10r, m10i, m11r, m11i = sreg2(m + 8), sreg2(m + 10), sreg2(m + 12), sreg2(m + 14)
    e = lines.append
    # the entry's predicates (plan.hpp): the listed bits of (base, ~base) and of the extended thread index (tid, ~tid) all set --
    # the control-is-0 entry of a multiplexed gate lists its control among the complemented bits, at no cost here
    e(f"s_and_b64 {TP}, {BASEP}, {cgpair}")
    e(f"s_cmp_eq_u64 {TP}, {cgpair}")
    e(f"s_cbranch_scc0 Lskip{tag}_%=")
    e(f"v_and_b32 %[vt], {ct}, %[tid]")
    e(f"v_cmp_eq_u32 vcc, {ct}, %[vt]")
    e(f"s_and_saveexec_b64 {SAVE}, vcc")
    e(f"s_cbranch_execz Lrest{tag}_%=")
    # bit 24 of the first word: the matrix is a product, its m00 complex -> the 16-operation body
    e(f"s_bitcmp1_b32 {w0}, 24")
    e(f"s_cbranch_scc1 Lgen{tag}_%=")
    for general in (False, True):
        x = "g" if general else ""
        if general:
            e(f"Lgen{tag}_%=:")
        if r > 1:
            e(f"s_and_b32 {T0}, {w0}, 0xff")
            for j in range(r - 1):
                e(f"s_cmp_eq_u32 {T0}, {j}")
                e(f"s_cbranch_scc1 Lj{x}{j}{tag}_%=")
        order = [r - 1] + list(range(r - 1))  # fall-through case first, then the branch targets
        for pos, j in enumerate(order):
            if j != r - 1:
                e(f"Lj{x}{j}{tag}_%=:")
            pair = 0
            for e0 in range(1 << r):
                if e0 & (1 << j):
                    continue
                e1 = e0 | (1 << j)
                a0r, a0i, a1r, a1i = amp_re(e0), amp_im(e0), amp_re(e1), amp_im(e1)
                if "pairtest" not in ABL:
                    e(f"s_bitcmp1_b32 {w0}, {16 + pair}")
                    e(f"s_cbranch_scc0 Ln{x}{j}_{pair}{tag}_%=")
                # four accumulation chains taken round-robin: every operation is four issue slots behind the one it
                # depends on (the dependent issue of v_fma_f64 is longer than two slots), and every amplitude register
                # is overwritten only after its last reader
                if "gatevalu" not in ABL:
                    e(f"v_mul_f64 %[p], {m10r}, {a0r}")
                    e(f"v_mul_f64 %[q], {m10r}, {a0i}")
                    e(f"v_mul_f64 %[u], {m00}, {a0r}")
                    e(f"v_mul_f64 %[w], {m00}, {a0i}")
                    e(f"v_fma_f64 %[p], -{m10i}, {a0i}, %[p]")
                    e(f"v_fma_f64 %[q], {m10i}, {a0r}, %[q]")
                    if general:  # (Im m00: the two operations a u-type matrix saves)
                        e(f"v_fma_f64 %[u], -{m00i}, {a0i}, %[u]")
                        e(f"v_fma_f64 %[w], {m00i}, {a0r}, %[w]")
                        e(f"v_fma_f64 %[p], {m11r}, {a1r}, %[p]")
                        e(f"v_fma_f64 %[q], {m11i}, {a1r}, %[q]")
                    e(f"v_fma_f64 %[u], {m01r}, {a1r}, %[u]")
                    e(f"v_fma_f64 %[w], {m01r}, {a1i}, %[w]")
                    if not general:
                        e(f"v_fma_f64 %[p], {m11r}, {a1r}, %[p]")
                        e(f"v_fma_f64 %[q], {m11i}, {a1r}, %[q]")
                    e(f"v_fma_f64 {a0r}, -{m01i}, {a1i}, %[u]")
                    e(f"v_fma_f64 {a0i}, {m01i}, {a1r}, %[w]")
                    e(f"v_fma_f64 {a1r}, -{m11i}, {a1i}, %[p]")
                    e(f"v_fma_f64 {a1i}, {m11r}, {a1i}, %[q]")
                e(f"Ln{x}{j}_{pair}{tag}_%=:")
                pair += 1
            if not (general and pos + 1 == len(order)):
                e(f"s_branch Lrest{tag}_%=")
    e(f"Lrest{tag}_%=:")
    e(f"s_mov_b64 exec, {SAVE}")
    e(f"Lskip{tag}_%=:")


def loop_body(r: int) -> list[str]:
    rp, mp = sreg2(RP), sreg2(MP)
    lines: list[str] = []
    e = lines.append
    e(f"s_mov_b64 {rp}, %[rp]")
    e(f"s_mov_b64 {mp}, %[mp]")
    e(f"s_mov_b32 s{N}, %[n]")
    e("s_mov_b32 s96, %[base]")
    e("s_not_b32 s97, %[base]")
    gate_loop_core(lines, r)
    return lines


def gate_loop_core(lines: list[str], r: int) -> None:
    """The gate loop proper; expects the descriptor pointer in s[RP:RP+1], the matrix pointer in s[MP:MP+1] and the
    gate count (> 0) in s[N].  Leaves the pointers somewhere inside the gate list (the caller keeps its own)."""
    a, b = MAT["A"], MAT["B"]
    da, db = DESC["A"], DESC["B"]
    rp, mp = sreg2(RP), sreg2(MP)
    n = f"s{N}"
    e = lines.append
    e(f"s_load_dwordx4 s[{da}:{da + 3}], {rp}, 0x0")
    e(f"s_load_dwordx16 s[{a}:{a + 15}], {mp}, 0x0")
    e("Lloop_%=:")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_load_dwordx4 s[{db}:{db + 3}], {rp}, 0x10")
    e(f"s_load_dwordx16 s[{b}:{b + 15}], {mp}, 0x40")
    gate(lines, r, "A", "a")
    e(f"s_sub_u32 {n}, {n}, 1")
    e(f"s_cmp_eq_u32 {n}, 0")
    e("s_cbranch_scc1 Ldone_%=")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_load_dwordx4 s[{da}:{da + 3}], {rp}, 0x20")
    e(f"s_load_dwordx16 s[{a}:{a + 15}], {mp}, 0x80")
    gate(lines, r, "B", "b")
    e(f"s_add_u32 s{RP}, s{RP}, 32")
    e(f"s_addc_u32 s{RP + 1}, s{RP + 1}, 0")
    e(f"s_add_u32 s{MP}, s{MP}, 128")
    e(f"s_addc_u32 s{MP + 1}, s{MP + 1}, 0")
    e(f"s_sub_u32 {n}, {n}, 1")
    e(f"s_cmp_lg_u32 {n}, 0")
    e("s_cbranch_scc1 Lloop_%=")
    e("Ldone_%=:")
    e("s_waitcnt lgkmcnt(0)")  # the last prefetch must land before the compiler may reuse these registers


def emit(r: int) -> str:
    nr = 1 << r
    out = []
    out.append(f"// R = {r}: {nr} amplitudes per thread")
    out.append("template <>")
    out.append(f"struct GateLoopF64<{r}> {{")
    out.append(
        f"    static __device__ __forceinline__ void run(cx<double> (&amp)[{nr}], cu32p rp, cf64p mp,\n"
        "                                               uint32_t n_gates, uint32_t base, uint32_t tid) {"
    )
    out.append("        // uniform by construction; say so to the compiler, which otherwise may hand a VGPR to an \"s\" operand")
    out.append("        base = __builtin_amdgcn_readfirstlane(base);")
    out.append("        n_gates = __builtin_amdgcn_readfirstlane(n_gates);")
    out.append("        double u, w, p, q;")
    out.append("        uint32_t vt;")
    out.append("        asm volatile(")
    for line in loop_body(r):
        out.append(f'            "{line}\\n\\t"')
    outs = amp_operands(nr)
    outs += ['[u] "=&v"(u)', '[w] "=&v"(w)', '[p] "=&v"(p)', '[q] "=&v"(q)', '[vt] "=&v"(vt)']
    out.append("            : " + ",\n              ".join(outs))
    out.append('            : [rp] "s"(rp), [mp] "s"(mp), [n] "s"(n_gates), [base] "s"(base), [tid] "v"(tid)')
    clob = ['"vcc"', '"scc"'] + [f'"s{i}"' for i in CLOBBER_RANGE]
    rows = [", ".join(clob[i : i + 12]) for i in range(0, len(clob), 12)]
    out.append("            : " + ",\n              ".join(rows) + ");")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


# ---- lane swaps (plan.hpp "swap" rounds; kernels.hip swap_reg_lane is the C++ statement of the same thing) ---------
# One assembly block over the fixed amplitude registers; the (register bit V, lane bit U) case is picked by a binary
# tree of scalar compares, so no amplitude register is ever copied (written with builtins in C++, hipcc kept a second
# copy of all amplitudes alive across the dispatch: 48 moves around 16 swaps and 64 more VGPRs).  Wait states: a DPP or permlane read needs 2 wait states after a VALU
# write of the register it reads; every case starts with s_nop 1 (the block's inputs may just have been written) and the
# DPP cases keep two pairs in flight so that a temporary is read three instructions after it was written.
def swap_case(lines: list[str], r: int, v: int, u: int) -> None:
    e = lines.append
    pairs = [(e0, e0 | (1 << v)) for e0 in range(1 << r) if not e0 & (1 << v)]
    # (A, B) dword pairs: the four 32-bit halves of both component planes
    regs = [(f"v{AMP0 + 4 * a + d}", f"v{AMP0 + 4 * b + d}") for a, b in pairs for d in range(4)]
    e("s_nop 1")
    if "swapvalu" in ABL:
        return
    if u >= 4:
        op = "v_permlane32_swap_b32" if u == 5 else "v_permlane16_swap_b32"
        for a, b in regs:
            e(f"{op} {a}, {b}")
    elif u >= 2:
        n = 1 << u
        upper, lower = ("0xa", "0x5") if u == 2 else ("0xc", "0x3")
        for i in range(0, len(regs), 2):
            chunk = regs[i : i + 2]
            temps = ["%[t0]", "%[t1]"]
            for (a, _b), t in zip(chunk, temps):
                e(f"v_mov_b32 {t}, {a}")
            for (a, b), _t in zip(chunk, temps):
                e(f"v_mov_b32_dpp {a}, {b} row_shr:{n} row_mask:0xf bank_mask:{upper}")
            for (_a, b), t in zip(chunk, temps):
                e(f"v_mov_b32_dpp {b}, {t} row_shl:{n} row_mask:0xf bank_mask:{lower}")
    else:
        perm = "quad_perm:[1,0,3,2]" if u == 0 else "quad_perm:[2,3,0,1]"
        e(f"v_and_b32 %[t0], {1 << u}, %[lane]")
        e("v_cmp_ne_u32 vcc, 0, %[t0]")  # vcc = lanes whose bit U is set
        for i in range(0, len(regs), 2):
            chunk = regs[i : i + 2]
            temps = ["%[t0]", "%[t1]"]
            for (_a, b), t in zip(chunk, temps):
                e(f"v_mov_b32_dpp {t}, {b} {perm} row_mask:0xf bank_mask:0xf")  # the partner's B
            for (a, b), _t in zip(chunk, temps):
                e(f"v_cndmask_b32_dpp {b}, {a}, {b}, vcc {perm} row_mask:0xf bank_mask:0xf")  # upper ? B : partner's A
            for (a, _b), t in zip(chunk, temps):
                e(f"v_cndmask_b32 {a}, {a}, {t}, vcc")  # upper ? partner's B : A


def dispatch_tree(lines: list[str], lo: int, hi: int, labels: list[str], tag: str) -> None:
    """Binary decision tree over %[sel] in [lo, hi): about log2(hi - lo) compare + branch pairs on any path."""
    if hi - lo == 1:
        lines.append(f"s_branch {labels[lo]}")
        return
    mid = (lo + hi) // 2
    lines.append(f"s_cmp_lt_u32 %[sel], {mid}")
    lines.append(f"s_cbranch_scc1 Lt{tag}{lo}_{mid}_%=")
    dispatch_tree(lines, mid, hi, labels, tag)
    lines.append(f"Lt{tag}{lo}_{mid}_%=:")
    dispatch_tree(lines, lo, mid, labels, tag)


def emit_swap(r: int) -> str:
    nr = 1 << r
    lines: list[str] = []
    e = lines.append
    cases = [(v, u) for v in range(r) for u in range(6)]
    labels = [f"Lc{v}_{u}_%=" for v, u in cases]
    dispatch_tree(lines, 0, len(cases), labels, "s")
    for (v, u), label in zip(cases, labels):
        e(f"{label}:")
        swap_case(lines, r, v, u)
        e("s_branch Lend_%=")
    e("Lend_%=:")
    out = []
    out.append(f"// R = {r}: lane swap of the {nr} amplitudes of a thread, both component planes (sel = 6 V + U < {6 * r})")
    out.append("template <>")
    out.append(f"struct SwapF64<{r}> {{")
    out.append(f"    static __device__ __forceinline__ void run(cx<double> (&amp)[{nr}], uint32_t sel, uint32_t lane) {{")
    out.append("        sel = __builtin_amdgcn_readfirstlane(sel);")
    out.append("        uint32_t t0, t1;")
    out.append("        asm volatile(")
    for line in lines:
        out.append(f'            "{line}\\n\\t"')
    outs = amp_operands(nr) + ['[t0] "=&v"(t0)', '[t1] "=&v"(t1)']
    out.append("            : " + ",\n              ".join(outs))
    out.append('            : [sel] "s"(sel), [lane] "v"(lane)')
    out.append('            : "vcc", "scc");')
    out.append("    }")
    out.append("};")
    return "\n".join(out)


# ---- the whole round loop of a tile (exchange mode 2) ---------------------------------------------------------------
# kernels.hip states the same loop in C++ (used for fp32 and the other exchange modes).  Around the separate assembly
# blocks of that version hipcc carries the amplitudes through temporaries -- 16 v_mov_b64 into the fixed registers
# before every block and 16 out after it -- because their definitions alternate between its own code (LDS reads) and
# the blocks.  Here rounds, LDS exchanges, lane swaps and gates are ONE block per tile and nothing moves.
RH, ROUNDS, FLAGS, T1, T2, T3 = 89, 90, 91, 83, 92, 93
SAVE2 = "s[94:95]"
NEXT_RP, NEXT_MP = "s[92:93]", "s[94:95]"  # (the gate loop's exit values; T2 / T3 / SAVE2 are free while gates run)
ROUND_CLOBBERS = range(40, 98)
WC, RC = 40, 56  # scalar registers of the LDS write / read columns during an exchange (9 thread + 4 register columns)
K_THREAD_COLS = 9


def emit_rounds(r: int) -> str:
    nr = 1 << r
    rp, mp = sreg2(RP), sreg2(MP)
    lines: list[str] = []
    e = lines.append
    e(f"s_mov_b64 {rp}, %[rp]")
    e(f"s_mov_b64 {mp}, %[mp]")
    e(f"s_mov_b32 s{ROUNDS}, %[rounds]")
    e(f"s_mov_b32 s{FLAGS}, %[flags]")
    e("s_mov_b32 s96, %[base]")
    e("s_not_b32 s97, %[base]")
    e("Lround_%=:")
    e(f"s_load_dword s{RH}, {rp}, 0x0")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_bitcmp1_b32 s{RH}, 16")
    e("s_cbranch_scc1 Lexch_%=")
    e(f"s_bitcmp1_b32 s{RH}, 18")
    e("s_cbranch_scc1 Lswap_%=")
    e(f"s_add_u32 s{RP}, s{RP}, 4")
    e(f"s_addc_u32 s{RP + 1}, s{RP + 1}, 0")
    e("Lgates_%=:")
    e(f"s_and_b32 s{N}, s{RH}, 0xffff")
    e(f"s_cmp_eq_u32 s{N}, 0")
    e("s_cbranch_scc1 Lnext_%=")
    # where the pointers stand after this round's gates: 16 bytes of descriptor, 64 bytes of matrix per gate
    e(f"s_lshl_b32 s{T1}, s{N}, 4")
    e(f"s_add_u32 s92, s{RP}, s{T1}")
    e(f"s_addc_u32 s93, s{RP + 1}, 0")
    e(f"s_lshl_b32 s{T1}, s{N}, 6")
    e(f"s_add_u32 s94, s{MP}, s{T1}")
    e(f"s_addc_u32 s95, s{MP + 1}, 0")
    if "gateloop" not in ABL:
        gate_loop_core(lines, r)
    e(f"s_mov_b64 {rp}, {NEXT_RP}")
    e(f"s_mov_b64 {mp}, {NEXT_MP}")
    e("Lnext_%=:")
    e(f"s_sub_u32 s{ROUNDS}, s{ROUNDS}, 1")
    e(f"s_cmp_lg_u32 s{ROUNDS}, 0")
    e("s_cbranch_scc1 Lround_%=")
    e("s_branch Lfinish_%=")

    # ---- lane swaps: up to four words, always taken from s40
    e("Lswap_%=:")
    e(f"s_load_dwordx4 s[40:43], {rp}, 0x4")
    e(f"s_add_u32 s{RP}, s{RP}, 20")
    e(f"s_addc_u32 s{RP + 1}, s{RP + 1}, 0")
    e("s_waitcnt lgkmcnt(0)")
    e("Lswapnext_%=:")
    e("s_cmp_eq_u32 s40, -1")
    e("s_cbranch_scc1 Lgates_%=")
    e(f"s_and_b32 {T0}, s40, 0xff")
    e(f"s_mul_i32 {T0}, {T0}, 6")
    e(f"s_bfe_u32 s{T1}, s40, 0x80008")
    e(f"s_add_u32 {T0}, {T0}, s{T1}")
    cases = [(v, u) for v in range(r) for u in range(6)]
    labels = [f"Lc{v}_{u}_%=" for v, u in cases]
    tree: list[str] = []
    dispatch_tree(tree, 0, len(cases), labels, "r")
    lines.extend(x.replace("%[sel]", T0) for x in tree)
    for (v, u), label in zip(cases, labels):
        e(f"{label}:")
        swap_case(lines, r, v, u)
        e("s_branch Lswapdone_%=")
    e("Lswapdone_%=:")
    e("s_mov_b32 s40, s41")
    e("s_mov_b32 s41, s42")
    e("s_mov_b32 s42, s43")
    e("s_mov_b32 s43, -1")
    e("s_branch Lswapnext_%=")

    # ---- LDS exchange (mode 2: real plane, then imaginary plane, through one plane buffer)
    e("Lexch_%=:")
    for base, off in ((WC, 4), (RC, 4 + 52)):
        e(f"s_load_dwordx8 s[{base}:{base + 7}], {rp}, {hex(off)}")
        e(f"s_load_dwordx4 s[{base + 8}:{base + 11}], {rp}, {hex(off + 32)}")
        e(f"s_load_dword s{base + 12}, {rp}, {hex(off + 48)}")
    e(f"s_add_u32 s{RP}, s{RP}, {4 + 104}")
    e(f"s_addc_u32 s{RP + 1}, s{RP + 1}, 0")
    # a barrier first if some wave may still be reading what the previous exchange left in LDS: after an exchange
    # that stayed inside each wave only this wave's own reads matter, and they have completed
    e(f"s_bitcmp1_b32 s{RH}, 17")
    e(f"s_cselect_b32 s{T2}, 2, 1")
    e(f"s_and_b32 s{T2}, s{FLAGS}, s{T2}")
    e("s_cbranch_scc0 Lnobar_%=")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    e(f"s_andn2_b32 s{FLAGS}, s{FLAGS}, 2")
    e("Lnobar_%=:")
    e("s_waitcnt lgkmcnt(0)")
    # thread parts of the two LDS offsets: lane bits on the vector unit, wave-index bits on the scalar unit
    e("v_mov_b32 %[vt], 0")
    e("v_mov_b32 %[t0], 0")
    for u in range(6):
        e(f"v_bfe_i32 %[t2], %[tid], {u}, 1")
        e(f"v_and_b32 %[t3], s{WC + u}, %[t2]")
        e("v_xor_b32 %[vt], %[vt], %[t3]")
        e(f"v_and_b32 %[t3], s{RC + u}, %[t2]")
        e("v_xor_b32 %[t0], %[t0], %[t3]")
    e(f"s_mov_b32 s{T2}, 0")
    e(f"s_mov_b32 s{T3}, 0")
    for u in range(6, K_THREAD_COLS):
        e(f"s_bitcmp1_b32 %[wave], {u - 6}")
        e(f"s_cselect_b32 s94, s{WC + u}, 0")
        e(f"s_cselect_b32 s95, s{RC + u}, 0")
        e(f"s_xor_b32 s{T2}, s{T2}, s94")
        e(f"s_xor_b32 s{T3}, s{T3}, s95")
    e(f"v_xor_b32 %[vt], s{T2}, %[vt]")
    e(f"v_xor_b32 %[t0], s{T3}, %[t0]")
    e("v_lshlrev_b32 %[vt], 3, %[vt]")
    e("v_lshlrev_b32 %[t0], 3, %[t0]")
    e("v_add_u32 %[vt], %[lds], %[vt]")
    e("v_add_u32 %[t0], %[lds], %[t0]")
    for v in range(r):
        e(f"s_lshl_b32 s{WC + K_THREAD_COLS + v}, s{WC + K_THREAD_COLS + v}, 3")
        e(f"s_lshl_b32 s{RC + K_THREAD_COLS + v}, s{RC + K_THREAD_COLS + v}, 3")

    def gray(i: int) -> int:
        return i ^ (i >> 1)

    def ctz(i: int) -> int:
        return (i & -i).bit_length() - 1

    def barrier_unless_intra(tag: str) -> None:
        e(f"s_bitcmp1_b32 s{RH}, 17")
        e(f"s_cbranch_scc1 Lib{tag}_%=")
        e("s_waitcnt lgkmcnt(0)")
        e("s_barrier")
        e(f"Lib{tag}_%=:")

    for plane, reg in (("re", amp_re), ("im", amp_im)):
        # write (threads of the tile only), barrier, read
        e(f"s_mov_b64 {SAVE2}, exec")
        e(f"s_and_b64 exec, exec, %[active]")
        e("v_mov_b32 %[t1], %[vt]")
        for i in range(nr):
            if i:
                e(f"v_xor_b32 %[t1], s{WC + K_THREAD_COLS + ctz(i)}, %[t1]")
            e(f"ds_write_b64 %[t1], {reg(gray(i))}")
        e(f"s_mov_b64 exec, {SAVE2}")
        barrier_unless_intra("w" + plane)
        e("v_mov_b32 %[t1], %[t0]")
        for i in range(nr):
            if i:
                e(f"v_xor_b32 %[t1], s{RC + K_THREAD_COLS + ctz(i)}, %[t1]")
            e(f"ds_read_b64 {reg(gray(i))}, %[t1]")
        if plane == "re":
            barrier_unless_intra("r" + plane)
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_or_b32 s{FLAGS}, s{FLAGS}, 1")
    e(f"s_bitcmp1_b32 s{RH}, 17")
    e("s_cbranch_scc1 Lgates_%=")
    e(f"s_or_b32 s{FLAGS}, s{FLAGS}, 2")
    e("s_branch Lgates_%=")

    e("Lfinish_%=:")
    e(f"s_mov_b32 %[flags], s{FLAGS}")

    out = []
    out.append(f"// R = {r}: every round of a tile (LDS exchanges in mode 2, lane swaps, gates)")
    out.append("template <>")
    out.append(f"struct RoundLoopF64<{r}> {{")
    out.append(
        f"    static __device__ __forceinline__ void run(cx<double> (&amp)[{nr}], cu32p rp, cf64p mp, uint32_t n_rounds,\n"
        "                                               uint32_t base, uint32_t tid, uint32_t wave, uint64_t active,\n"
        "                                               uint32_t lds, uint32_t& flags) {"
    )
    out.append("        base = __builtin_amdgcn_readfirstlane(base);")
    out.append("        n_rounds = __builtin_amdgcn_readfirstlane(n_rounds);")
    out.append("        wave = __builtin_amdgcn_readfirstlane(wave);")
    out.append("        lds = __builtin_amdgcn_readfirstlane(lds);")
    out.append("        uint32_t fl = __builtin_amdgcn_readfirstlane(flags);")
    out.append("        double u, w, p, q;")
    out.append("        uint32_t vt, t0, t1, t2, t3;")
    out.append("        asm volatile(")
    for line in lines:
        out.append(f'            "{line}\\n\\t"')
    outs = amp_operands(nr)
    outs += ['[u] "=&v"(u)', '[w] "=&v"(w)', '[p] "=&v"(p)', '[q] "=&v"(q)', '[vt] "=&v"(vt)', '[t0] "=&v"(t0)',
             '[t1] "=&v"(t1)', '[t2] "=&v"(t2)', '[t3] "=&v"(t3)', '[flags] "+s"(fl)']
    out.append("            : " + ",\n              ".join(outs))
    out.append('            : [rp] "s"(rp), [mp] "s"(mp), [rounds] "s"(n_rounds), [base] "s"(base), [tid] "v"(tid), [wave] "s"(wave),\n'
               '              [active] "s"(active), [lds] "s"(lds), [lane] "v"(tid & 63u)')
    clob = ['"vcc"', '"scc"', '"memory"'] + [f'"s{i}"' for i in ROUND_CLOBBERS]
    rows = [", ".join(clob[i : i + 12]) for i in range(0, len(clob), 12)]
    out.append("            : " + ",\n              ".join(rows) + ");")
    out.append("        flags = fl;")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


# ---- the same round loop for single precision (exchange mode 0) -------------------------------------------------------
# An amplitude is ONE 64-bit register pair v[AMP0 + 2e : AMP0 + 2e + 1] = (re, im): a global_load_dwordx2 lands in it, an LDS
# exchange moves it with one ds_write_b64 / ds_read_b64 (mode 0: the whole element, one phase), a lane swap moves two dwords --
# and a butterfly is EIGHT packed operations (v_pk_mul_f32 / v_pk_fma_f32: both components of a complex number per
# instruction, the second operand's halves swapped and one product negated by op_sel / neg_lo) where the fp64 body takes 14 - 16:
#     t = (mr, mr) * (ar, ai)              v_pk_mul_f32 t, m, a       op_sel:[0,0]   op_sel_hi:[0,1]
#     t = (-mi, mi) * (ai, ar) + t         v_pk_fma_f32 t, m, a, t    op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]
# The matrix arrives as eight FLOATS at the start of the gate's 64-byte record (single-precision handles: prepare_eval rounds once
# per evaluation) in four scalar register pairs, which the packed operations read directly.  One body serves u-type matrices
# and products of matrices (Im m00 != 0) alike.  Vector registers behind the amplitudes are fixed and declared as clobbers.
def amp32(e: int) -> str:
    return f"v[{AMP0 + 2 * e}:{AMP0 + 2 * e + 1}]"


def amp32_operands(nr: int) -> list[str]:
    outs = []
    for e in range(nr):
        outs.append(f'"+{{v{AMP0 + 2 * e}}}"(amp[{e}].re)')
        outs.append(f'"+{{v{AMP0 + 2 * e + 1}}}"(amp[{e}].im)')
    return outs


def vregs32(r: int) -> dict:
    base = AMP0 + 2 * (1 << r)
    names = {"m00": base, "m01": base + 2, "m10": base + 4, "m11": base + 6, "p": base + 8, "u": base + 10}
    scal = {"vt": base + 12, "t0": base + 13, "t1": base + 14, "t2": base + 15, "t3": base + 16}
    return {"pairs": names, "scal": scal, "last": base + 16}


def gate32(lines: list[str], r: int, x: str, tag: str) -> None:
    m, d = MAT[x], DESC[x]
    w0, ct = f"s{d}", f"s{d + 1}"
    cgpair = sreg2(d + 2)
    vr = vregs32(r)
    P = {k: f"v[{v}:{v + 1}]" for k, v in vr["pairs"].items()}
    vt = f"v{vr['scal']['vt']}"
    e = lines.append
    e(f"s_and_b64 {TP}, {BASEP}, {cgpair}")
    e(f"s_cmp_eq_u64 {TP}, {cgpair}")
    e(f"s_cbranch_scc0 Lskip{tag}_%=")
    e(f"v_and_b32 {vt}, {ct}, %[tid]")
    e(f"v_cmp_eq_u32 vcc, {ct}, {vt}")
    e(f"s_and_saveexec_b64 {SAVE}, vcc")
    e(f"s_cbranch_execz Lrest{tag}_%=")
    # the matrix arrives as eight floats (prepare_eval's float_mats): (re, im) of m00 m01 m10 m11 are four scalar register
    # pairs, read by the packed operations as they are (one scalar operand per instruction)
    for i, name in enumerate(("m00", "m01", "m10", "m11")):
        P[name] = sreg2(m + 2 * i)
    if r > 1:
        e(f"s_and_b32 {T0}, {w0}, 0xff")
        for j in range(r - 1):
            e(f"s_cmp_eq_u32 {T0}, {j}")
            e(f"s_cbranch_scc1 Lj{j}{tag}_%=")
    order = [r - 1] + list(range(r - 1))
    first, second = "op_sel:[0,0] op_sel_hi:[0,1]", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
    first3 = "op_sel:[0,0,0] op_sel_hi:[0,1,1]"
    for pos, j in enumerate(order):
        if j != r - 1:
            e(f"Lj{j}{tag}_%=:")
        pair = 0
        for e0 in range(1 << r):
            if e0 & (1 << j):
                continue
            e1 = e0 | (1 << j)
            a0, a1 = amp32(e0), amp32(e1)
            if "pairtest" not in ABL:
                e(f"s_bitcmp1_b32 {w0}, {16 + pair}")
                e(f"s_cbranch_scc0 Ln{j}_{pair}{tag}_%=")
            if "gatevalu" not in ABL:
                # p = m10 a0, u = m00 a0 (a0 is free afterwards); a0 = u + m01 a1; a1 = p + m11 a1
                e(f"v_pk_mul_f32 {P['p']}, {P['m10']}, {a0} {first}")
                e(f"v_pk_mul_f32 {P['u']}, {P['m00']}, {a0} {first}")
                e(f"v_pk_fma_f32 {P['p']}, {P['m10']}, {a0}, {P['p']} {second}")
                e(f"v_pk_fma_f32 {P['u']}, {P['m00']}, {a0}, {P['u']} {second}")
                e(f"v_pk_fma_f32 {P['p']}, {P['m11']}, {a1}, {P['p']} {first3}")
                e(f"v_pk_fma_f32 {P['u']}, {P['m01']}, {a1}, {P['u']} {first3}")
                e(f"v_pk_fma_f32 {a0}, {P['m01']}, {a1}, {P['u']} {second}")
                e(f"v_pk_fma_f32 {a1}, {P['m11']}, {a1}, {P['p']} {second}")
            e(f"Ln{j}_{pair}{tag}_%=:")
            pair += 1
        if pos + 1 != len(order):
            e(f"s_branch Lrest{tag}_%=")
    e(f"Lrest{tag}_%=:")
    e(f"s_mov_b64 exec, {SAVE}")
    e(f"Lskip{tag}_%=:")


def gate_loop_core32(lines: list[str], r: int) -> None:
    a, b = MAT["A"], MAT["B"]
    da, db = DESC["A"], DESC["B"]
    rp, mp = sreg2(RP), sreg2(MP)
    n = f"s{N}"
    e = lines.append
    e(f"s_load_dwordx4 s[{da}:{da + 3}], {rp}, 0x0")
    e(f"s_load_dwordx8 s[{a}:{a + 7}], {mp}, 0x0")
    e("Lloop_%=:")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_load_dwordx4 s[{db}:{db + 3}], {rp}, 0x10")
    e(f"s_load_dwordx8 s[{b}:{b + 7}], {mp}, 0x40")
    gate32(lines, r, "A", "a")
    e(f"s_sub_u32 {n}, {n}, 1")
    e(f"s_cmp_eq_u32 {n}, 0")
    e("s_cbranch_scc1 Ldone_%=")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_load_dwordx4 s[{da}:{da + 3}], {rp}, 0x20")
    e(f"s_load_dwordx8 s[{a}:{a + 7}], {mp}, 0x80")
    gate32(lines, r, "B", "b")
    e(f"s_add_u32 s{RP}, s{RP}, 32")
    e(f"s_addc_u32 s{RP + 1}, s{RP + 1}, 0")
    e(f"s_add_u32 s{MP}, s{MP}, 128")
    e(f"s_addc_u32 s{MP + 1}, s{MP + 1}, 0")
    e(f"s_sub_u32 {n}, {n}, 1")
    e(f"s_cmp_lg_u32 {n}, 0")
    e("s_cbranch_scc1 Lloop_%=")
    e("Ldone_%=:")
    e("s_waitcnt lgkmcnt(0)")


def swap_case32(lines: list[str], r: int, v: int, u: int) -> None:
    e = lines.append
    vr = vregs32(r)
    t0, t1 = f"v{vr['scal']['t0']}", f"v{vr['scal']['t1']}"
    pairs = [(e0, e0 | (1 << v)) for e0 in range(1 << r) if not e0 & (1 << v)]
    regs = [(f"v{AMP0 + 2 * a + d}", f"v{AMP0 + 2 * b + d}") for a, b in pairs for d in range(2)]
    e("s_nop 1")
    if "swapvalu" in ABL:
        return
    if u >= 4:
        op = "v_permlane32_swap_b32" if u == 5 else "v_permlane16_swap_b32"
        for a, b in regs:
            e(f"{op} {a}, {b}")
    elif u >= 2:
        n = 1 << u
        upper, lower = ("0xa", "0x5") if u == 2 else ("0xc", "0x3")
        for i in range(0, len(regs), 2):
            chunk = regs[i : i + 2]
            temps = [t0, t1]
            for (a, _b), t in zip(chunk, temps):
                e(f"v_mov_b32 {t}, {a}")
            for (a, b), _t in zip(chunk, temps):
                e(f"v_mov_b32_dpp {a}, {b} row_shr:{n} row_mask:0xf bank_mask:{upper}")
            for (_a, b), t in zip(chunk, temps):
                e(f"v_mov_b32_dpp {b}, {t} row_shl:{n} row_mask:0xf bank_mask:{lower}")
    else:
        perm = "quad_perm:[1,0,3,2]" if u == 0 else "quad_perm:[2,3,0,1]"
        e(f"v_and_b32 {t0}, {1 << u}, %[lane]")
        e(f"v_cmp_ne_u32 vcc, 0, {t0}")
        for i in range(0, len(regs), 2):
            chunk = regs[i : i + 2]
            temps = [t0, t1]
            for (_a, b), t in zip(chunk, temps):
                e(f"v_mov_b32_dpp {t}, {b} {perm} row_mask:0xf bank_mask:0xf")
            for (a, b), _t in zip(chunk, temps):
                e(f"v_cndmask_b32_dpp {b}, {a}, {b}, vcc {perm} row_mask:0xf bank_mask:0xf")
            for (a, _b), t in zip(chunk, temps):
                e(f"v_cndmask_b32 {a}, {a}, {t}, vcc")


def emit_rounds32(r: int) -> str:
    nr = 1 << r
    rp, mp = sreg2(RP), sreg2(MP)
    vr = vregs32(r)
    vt, t0, t1, t2, t3 = (f"v{vr['scal'][k]}" for k in ("vt", "t0", "t1", "t2", "t3"))
    lines: list[str] = []
    e = lines.append
    e(f"s_mov_b64 {rp}, %[rp]")
    e(f"s_mov_b64 {mp}, %[mp]")
    e(f"s_mov_b32 s{ROUNDS}, %[rounds]")
    e(f"s_mov_b32 s{FLAGS}, %[flags]")
    e("s_mov_b32 s96, %[base]")
    e("s_not_b32 s97, %[base]")
    e("Lround_%=:")
    e(f"s_load_dword s{RH}, {rp}, 0x0")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_bitcmp1_b32 s{RH}, 16")
    e("s_cbranch_scc1 Lexch_%=")
    e(f"s_bitcmp1_b32 s{RH}, 18")
    e("s_cbranch_scc1 Lswap_%=")
    e(f"s_add_u32 s{RP}, s{RP}, 4")
    e(f"s_addc_u32 s{RP + 1}, s{RP + 1}, 0")
    e("Lgates_%=:")
    e(f"s_and_b32 s{N}, s{RH}, 0xffff")
    e(f"s_cmp_eq_u32 s{N}, 0")
    e("s_cbranch_scc1 Lnext_%=")
    e(f"s_lshl_b32 s{T1}, s{N}, 4")
    e(f"s_add_u32 s92, s{RP}, s{T1}")
    e(f"s_addc_u32 s93, s{RP + 1}, 0")
    e(f"s_lshl_b32 s{T1}, s{N}, 6")
    e(f"s_add_u32 s94, s{MP}, s{T1}")
    e(f"s_addc_u32 s95, s{MP + 1}, 0")
    if "gateloop" not in ABL:
        gate_loop_core32(lines, r)
    e(f"s_mov_b64 {rp}, {NEXT_RP}")
    e(f"s_mov_b64 {mp}, {NEXT_MP}")
    e("Lnext_%=:")
    e(f"s_sub_u32 s{ROUNDS}, s{ROUNDS}, 1")
    e(f"s_cmp_lg_u32 s{ROUNDS}, 0")
    e("s_cbranch_scc1 Lround_%=")
    e("s_branch Lfinish_%=")

    e("Lswap_%=:")
    e(f"s_load_dwordx4 s[40:43], {rp}, 0x4")
    e(f"s_add_u32 s{RP}, s{RP}, 20")
    e(f"s_addc_u32 s{RP + 1}, s{RP + 1}, 0")
    e("s_waitcnt lgkmcnt(0)")
    e("Lswapnext_%=:")
    e("s_cmp_eq_u32 s40, -1")
    e("s_cbranch_scc1 Lgates_%=")
    e(f"s_and_b32 {T0}, s40, 0xff")
    e(f"s_mul_i32 {T0}, {T0}, 6")
    e(f"s_bfe_u32 s{T1}, s40, 0x80008")
    e(f"s_add_u32 {T0}, {T0}, s{T1}")
    cases = [(v, u) for v in range(r) for u in range(6)]
    labels = [f"Lc{v}_{u}_%=" for v, u in cases]
    tree: list[str] = []
    dispatch_tree(tree, 0, len(cases), labels, "r")
    lines.extend(x.replace("%[sel]", T0) for x in tree)
    for (v, u), label in zip(cases, labels):
        e(f"{label}:")
        swap_case32(lines, r, v, u)
        e("s_branch Lswapdone_%=")
    e("Lswapdone_%=:")
    e("s_mov_b32 s40, s41")
    e("s_mov_b32 s41, s42")
    e("s_mov_b32 s42, s43")
    e("s_mov_b32 s43, -1")
    e("s_branch Lswapnext_%=")

    # ---- LDS exchange, mode 0: the whole 8-byte element, one phase
    e("Lexch_%=:")
    for base, off in ((WC, 4), (RC, 4 + 52)):
        e(f"s_load_dwordx8 s[{base}:{base + 7}], {rp}, {hex(off)}")
        e(f"s_load_dwordx4 s[{base + 8}:{base + 11}], {rp}, {hex(off + 32)}")
        e(f"s_load_dword s{base + 12}, {rp}, {hex(off + 48)}")
    e(f"s_add_u32 s{RP}, s{RP}, {4 + 104}")
    e(f"s_addc_u32 s{RP + 1}, s{RP + 1}, 0")
    e(f"s_bitcmp1_b32 s{RH}, 17")
    e(f"s_cselect_b32 s{T2}, 2, 1")
    e(f"s_and_b32 s{T2}, s{FLAGS}, s{T2}")
    e("s_cbranch_scc0 Lnobar_%=")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    e(f"s_andn2_b32 s{FLAGS}, s{FLAGS}, 2")
    e("Lnobar_%=:")
    e("s_waitcnt lgkmcnt(0)")
    e(f"v_mov_b32 {vt}, 0")
    e(f"v_mov_b32 {t0}, 0")
    for u in range(6):
        e(f"v_bfe_i32 {t2}, %[tid], {u}, 1")
        e(f"v_and_b32 {t3}, s{WC + u}, {t2}")
        e(f"v_xor_b32 {vt}, {vt}, {t3}")
        e(f"v_and_b32 {t3}, s{RC + u}, {t2}")
        e(f"v_xor_b32 {t0}, {t0}, {t3}")
    e(f"s_mov_b32 s{T2}, 0")
    e(f"s_mov_b32 s{T3}, 0")
    for u in range(6, K_THREAD_COLS):
        e(f"s_bitcmp1_b32 %[wave], {u - 6}")
        e(f"s_cselect_b32 s94, s{WC + u}, 0")
        e(f"s_cselect_b32 s95, s{RC + u}, 0")
        e(f"s_xor_b32 s{T2}, s{T2}, s94")
        e(f"s_xor_b32 s{T3}, s{T3}, s95")
    e(f"v_xor_b32 {vt}, s{T2}, {vt}")
    e(f"v_xor_b32 {t0}, s{T3}, {t0}")
    e(f"v_lshlrev_b32 {vt}, 3, {vt}")
    e(f"v_lshlrev_b32 {t0}, 3, {t0}")
    e(f"v_add_u32 {vt}, %[lds], {vt}")
    e(f"v_add_u32 {t0}, %[lds], {t0}")
    for v in range(r):
        e(f"s_lshl_b32 s{WC + K_THREAD_COLS + v}, s{WC + K_THREAD_COLS + v}, 3")
        e(f"s_lshl_b32 s{RC + K_THREAD_COLS + v}, s{RC + K_THREAD_COLS + v}, 3")

    def gray(i: int) -> int:
        return i ^ (i >> 1)

    def ctz(i: int) -> int:
        return (i & -i).bit_length() - 1

    e(f"s_mov_b64 {SAVE2}, exec")
    e("s_and_b64 exec, exec, %[active]")
    e(f"v_mov_b32 {t1}, {vt}")
    for i in range(nr):
        if i:
            e(f"v_xor_b32 {t1}, s{WC + K_THREAD_COLS + ctz(i)}, {t1}")
        e(f"ds_write_b64 {t1}, {amp32(gray(i))}")
    e(f"s_mov_b64 exec, {SAVE2}")
    e(f"s_bitcmp1_b32 s{RH}, 17")
    e("s_cbranch_scc1 Libw_%=")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    e("Libw_%=:")
    e(f"v_mov_b32 {t1}, {t0}")
    for i in range(nr):
        if i:
            e(f"v_xor_b32 {t1}, s{RC + K_THREAD_COLS + ctz(i)}, {t1}")
        e(f"ds_read_b64 {amp32(gray(i))}, {t1}")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_or_b32 s{FLAGS}, s{FLAGS}, 1")
    e(f"s_bitcmp1_b32 s{RH}, 17")
    e("s_cbranch_scc1 Lgates_%=")
    e(f"s_or_b32 s{FLAGS}, s{FLAGS}, 2")
    e("s_branch Lgates_%=")

    e("Lfinish_%=:")
    e(f"s_mov_b32 %[flags], s{FLAGS}")

    out = []
    out.append(f"// R = {r}: every round of a tile in single precision (LDS exchanges in mode 0, lane swaps, gates in packed arithmetic)")
    out.append("template <>")
    out.append(f"struct RoundLoopF32<{r}> {{")
    out.append(
        f"    static __device__ __forceinline__ void run(cx<float> (&amp)[{nr}], cu32p rp, cf64p mp, uint32_t n_rounds,\n"
        "                                               uint32_t base, uint32_t tid, uint32_t wave, uint64_t active,\n"
        "                                               uint32_t lds, uint32_t& flags) {"
    )
    out.append("        base = __builtin_amdgcn_readfirstlane(base);")
    out.append("        n_rounds = __builtin_amdgcn_readfirstlane(n_rounds);")
    out.append("        wave = __builtin_amdgcn_readfirstlane(wave);")
    out.append("        lds = __builtin_amdgcn_readfirstlane(lds);")
    out.append("        uint32_t fl = __builtin_amdgcn_readfirstlane(flags);")
    out.append("        asm volatile(")
    for line in lines:
        out.append(f'            "{line}\\n\\t"')
    outs = amp32_operands(nr) + ['[flags] "+s"(fl)']
    out.append("            : " + ",\n              ".join(outs))
    out.append('            : [rp] "s"(rp), [mp] "s"(mp), [rounds] "s"(n_rounds), [base] "s"(base), [tid] "v"(tid), [wave] "s"(wave),\n'
               '              [active] "s"(active), [lds] "s"(lds), [lane] "v"(tid & 63u)')
    first_free = AMP0 + 2 * nr
    clob = ['"vcc"', '"scc"', '"memory"'] + [f'"s{i}"' for i in ROUND_CLOBBERS] + [f'"v{i}"' for i in range(first_free, vr["last"] + 1)]
    rows = [", ".join(clob[i : i + 12]) for i in range(0, len(clob), 12)]
    out.append("            : " + ",\n              ".join(rows) + ");")
    out.append("        flags = fl;")
    out.append("    }")
    out.append("};")
    return "\n".join(out)


def render() -> str:
    head = (
        "// GENERATED by gen_gate_loop.py -- do not edit; regenerate with `python gen_gate_loop.py`.\n"
        "// fp64 gate loop of pass_kernel in gfx950 assembly: see the generator's docstring for the design.\n"
        "// Included inside namespace qsv by kernels.hip, after cx<> is defined.\n\n"
        "template <int R>\nstruct GateLoopF64;  // specialised below for R = 1 .. 4\n\n"
    )
    swap_head = (
        "\n\n// Lane swaps (plan.hpp \"swap\" rounds), see gen_gate_loop.py.\n"
        "template <int R>\nstruct SwapF64;  // specialised below for R = 1 .. 4\n\n"
    )
    rounds_head = (
        "\n\n// The whole round loop of a tile (exchange mode 2), see gen_gate_loop.py.\n"
        "template <int R>\nstruct RoundLoopF64;  // specialised below for R = 1 .. 4\n\n"
    )
    rounds32_head = (
        "\n\n// The whole round loop of a tile in single precision (exchange mode 0, packed arithmetic), see gen_gate_loop.py.\n"
        "template <int R>\nstruct RoundLoopF32;  // specialised below for R = 1 .. 4\n\n"
    )
    return (head + "\n\n".join(emit(r) for r in (1, 2, 3, 4)) + swap_head + "\n\n".join(emit_swap(r) for r in (1, 2, 3, 4))
            + rounds_head + "\n\n".join(emit_rounds(r) for r in (1, 2, 3, 4))
            + rounds32_head + "\n\n".join(emit_rounds32(r) for r in (1, 2, 3, 4)) + "\n")


if __name__ == "__main__":
    text = render()
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(0 if OUT.exists() and OUT.read_text() == text else 1)
    out = Path(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] == "--out" else OUT
    if ABL and out == OUT:
        sys.exit("QSV_GEN_ABL is set: refusing to overwrite the shipped gate_loop_gen.inc (use --out)")
    out.write_text(text)
    print(f"wrote {out} ({len(text.splitlines())} lines)")
