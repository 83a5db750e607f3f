// Host-side pass scheduler for the tiled statevector kernel.  Pure C++ (no HIP): it is exercised on the CPU by
// tests/test_plan.py through qsv_plan_build().
//
// Model.  A circuit is a list of single-target gates (u: no control, cu3: one control) on n qubits, applied to
// |0..0>.
//
// 1. FOLDING.  Leading gates are absorbed into the initial state: a qubit that no real gate has touched yet is
//    still a tensor factor of the state, so a u gate on it only changes that factor, and a cu3 whose control has
//    never been targeted (it is still exactly |0>) is the identity.  What remains after folding are the REAL
//    gates; the state they act on first is the product state  (x)_q v_q,  which pass 0 synthesises on the fly
//    instead of reading anything from memory.
//
// 2. PASSES.  The real gates are swept over the state in passes.  A pass picks k qubits (the TILE); every tile of
//    2^k amplitudes (the other n-k index bits fixed) stays on chip for the whole pass.  Inside a pass the tile
//    lives in registers: each of 2^t threads holds 2^r amplitudes (k = t + r).  A ROUND chooses which r tile bits
//    are "register bits"; a gate is applied in a round in which its TARGET is a register bit (a 2x2 butterfly
//    between two registers of one thread).  The CONTROL of a cu3 never has to be a register bit, nor even in the
//    tile: it is a predicate on the register index, on the thread index or on the tile's fixed bits.  Between
//    rounds the tile is transposed through LDS (an EXCHANGE) so that other tile bits become register bits.
//
// 1b. FUSION.  On the timeline of a qubit, a u gate that directly follows or precedes a cu3 TARGETING that qubit (nothing
//    else touches the qubit in between) is multiplied into it: the pair becomes one MULTIPLEXED gate, matrix M1 = the
//    product with the controlled matrix where the control is 1, M0 = the product of the u gates alone where it is 0.
//    The schedule carries it as two entries with complementary predicates (a negated control for M0), each a plain
//    2x2 butterfly -- every amplitude pair of the target is still updated exactly once, where the separate u swept all
//    pairs and the cu3 half of them again (EVQE layers alternate rotations and controlled rotations: seven in ten u
//    gates of a deep individual disappear this way).  Neighbouring u gates on one qubit merge the same way.  A product
//    of u matrices no longer has a real m00: such entries carry a flag and take the 16-operation butterfly.
//
// All index maps (thread/register -> global offset, thread/register -> LDS offset) are GF(2)-linear, so each is
// shipped to the device as one column per thread bit and per register bit; the kernel XORs columns together.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace qsv {

struct PlanConfig {
    int tile_bits = 12;  // k for n >= k
    int reg_bits = 3;    // r
    int low_bits = 2;    // c: tile always contains qubits 0..c-1, so contiguous runs in HBM are >= 16 * 2^c bytes (64 B).
                         // Measured: single-gate sweeps stream at the same rate for c = 2, 3, 4 (5.25 TB/s at n = 26),
                         // and every qubit not spent on c is one more NEW qubit per later pass (+9 % evals/s at n = 20,
                         // +15 % at n = 24 for c = 2 against 4)
    int lane_bits = 2;   // of those, the lowest `lane_bits` must sit on lanes in the layouts that touch global memory
                         // (64-byte runs per 4 lanes keep full bandwidth; measured)
    int elem_bytes = 16; // bytes of one LDS access of the exchange (16: complex fp64; 8: complex fp32 or one fp64 plane)
    int amp_bytes = 16;  // bytes of one complex amplitude (16 = fp64, 8 = fp32)
    int xmode = 2;       // LDS exchange mode (kernels.hip): 0 whole element, 1 two resident planes, 2 one plane buffer
    bool fold = true;    // absorb leading gates into the synthesised initial product state
    bool compact = true; // (needs fold) pass 0 computes one tile per pattern of its outer control qubits, see below
    int retries = 32;    // randomised scheduling attempts (tiles only: cheap) when the first-come rule AND the local search over
                         // the tiles (plan.cpp) need more than two passes; > 0 also switches the local search on
    bool swaps = true;   // relayouts that only trade register bits for lane bits 2..5 run as in-register lane swaps
                         // (v_permlane16/32_swap, DPP row shifts) instead of an LDS exchange, see "round" below
    bool fuse = true;    // FUSION (below): u gates next to a cu3 on the same target become part of it
};

struct GateIn {
    int target;
    int control;  // -1 if none
    int op;       // index of the originating qsv_op (angle source)
};

// Resolved geometry for a given n (same for every circuit on the handle).
struct Geometry {
    int n = 0, k = 0, r = 0, t = 0, c = 0, cl = 0;
    int threads_active = 0;  // 2^t
    int threads_launch = 0;  // max(64, 2^t)
    uint32_t blocks_per_state = 0;  // tiles per state = 2^(n-k)
    size_t lds_bytes = 0;
};
Geometry make_geometry(int n_qubits, const PlanConfig& cfg);

struct PlanStats {
    int n_passes = 0;
    int n_rounds = 0;
    int n_exchanges = 0;
    int n_intra_wave_exchanges = 0;  // exchanges that stay inside each wave (no barrier)
    int n_swap_rounds = 0;           // relayouts done by lane swaps instead of an LDS exchange
    int n_swaps = 0;                 // ... and the (register bit, lane bit) transpositions they took
    int compact_bits = -1;           // >= 0: pass 0 is compact over this many outer control qubits
    int n_real_gates = 0;     // scheduled entries (a multiplexed gate takes two)
    int n_fused_gates = 0;    // u gates multiplied into a neighbouring gate on the same target
    int n_folded_gates = 0;   // u gates absorbed into the initial product state
    int n_dropped_gates = 0;  // cu3 gates whose control is still |0>: identity
    int lds_conflict_cycles = 0;  // extra LDS cycles per wave-instruction summed over exchanges (0 = conflict free)
    // amplitude pairs each pass really updates per state (controls and a compact first pass taken into account: a
    // control held by a register, a thread or the tile index halves a gate's pairs); one pair = 4 multiplications +
    // 10 fused multiply-adds
    std::vector<double> pass_pairs;
};

struct CircuitPlan {
    std::vector<uint32_t> words;  // encoded plan (layout below)
    PlanStats stats;
};

// ---- encoded layout (uint32 words; offsets relative to the circuit plan's first word) -------------------------
// circuit: [0] n_passes  [1] n_real (scheduled ENTRIES = matrices)  [2] n_qubits  [3] offset of the ANGLE TABLE
//          [4] offset of the FOLD INDEX  [5] n_fold_entries  [6] offset of the CHAIN INDEX  [7] n_factors
//          [8 .. 8+n_passes) pass offsets
// pass:    [0] k | r<<8 | t<<16 | n_rounds<<24      [1] index of the pass's first scheduled gate
//          [2] flags: bit 0 COMPACT_STORE, bit 1 COMPACT_LOAD, bits 8..15 m (see COMPACT below)   [3] reserved
//          [4 .. 17)  tile bit j -> qubit position (ascending), padded to kMaxTileBits entries with kPosPad
//          [17 .. 30) load layout:  kMaxThreadBits thread columns then kMaxRegBits register columns (global
//                     amplitude offsets), unused entries 0
//          [30 .. 43) store layout, same shape (COMPACT_STORE: offsets inside the pattern's tile, 1 << tile bit)
//          [43 .. 117) compact block, zero unless a flag is set:
//                     [43 .. 51)  COMPACT_STORE: positions of the m outer control qubits (pad 63)
//                     [51 .. 64)  COMPACT_LOAD: the load layout's columns in W-index space
//                     [64 .. 77)  ... and in tile-factor-index space
//                     [77 .. 97)  W-index column of tile-number bit j (the pass's outer qubits, ascending; pad 0)
//                     [97 .. 117) tile-factor-index column of tile-number bit j
//          [117 ..)   the rounds
//          Every block has a FIXED size whatever k, r, t are: the kernel fetches each block with a few wide scalar
//          loads issued together and indexes it with compile-time offsets; padded columns are 0 (XOR no-ops) and
//          padded positions insert a zero bit above every index bit (a no-op too), so nothing is predicated.
// COMPACT: pass 0 starts from a product state and only targets its tile qubits T; the other (outer) qubits O stay
//          in product form and matter only as controls.  With C = the m outer qubits pass 0 uses as controls, the
//          state after pass 0 is  psi[i] = F[o(i)] * W[x(i)][t(i)]  (o, x, t = bits of i on O, C, T): F = the product
//          of the outer qubits' initial factors (prepare_kernel's tile-factor table), W[x] = the tile vector for
//          control pattern x WITHOUT that factor.  A COMPACT_STORE pass 0 therefore computes 2^m tiles instead of
//          2^(n-k) and writes them back to back at the start of the state slot (W[x][t] at x * 2^k + t); the
//          COMPACT_LOAD pass 1 builds its input from W and F (both cache resident) instead of reading the state:
//          no full-state write and read between the first two passes.
// round:   [0] n_gates | has_exchange<<16 | intra_wave<<17 | swap<<18 (the exchange moves data only inside each wave: the wave-
//              index thread bits hold the same tile bits before and after, so the kernel skips the barriers)
//          if has_exchange: [1 .. 14) LDS write columns (previous layout), [14 .. 27) LDS read columns (this
//                           layout), each kMaxThreadBits + kMaxRegBits entries, ELEMENT units, same swizzle
//          else if swap (bit 18 of the header word): [1 .. 5) up to kMaxSwaps transpositions v | u << 8 (pad
//                           0xFFFFFFFF): the tile bit under register bit v trades places with the one under lane bit u
//                           (u < 6), applied in the order listed.  No LDS, no barrier: v_permlane32_swap /
//                           v_permlane16_swap for u = 5 / 4, DPP row shifts under a bank mask for u = 3 / 2, DPP quad
//                           permutes + selects for u = 1 / 0.  The scheduler takes this form whenever the round's
//                           targets already sit in registers or on lane bits (the wave-index bits can only be
//                           reached through LDS).  The first and the last layout of a pass keep the low tile bits on
//                           the low lanes, where global memory wants them: a gate-less round of swaps at the end of a
//                           pass brings them back when gates on those qubits took them away.
//          then 4 words per gate: [0] target register bit | control register bit<<8 (0xFF: none) | pair mask<<16
//                                     (bit p: the p-th amplitude pair, register indices with the target bit
//                                     clear in ascending order, takes part; R = 4: eight pairs) | flags (bits 24 ..):
//                                     kGateGeneral = Im m00 may be non-zero (a product of matrices), kGateNegated = the
//                                     entry applies where its control is 0 (informational: the predicates say it all)
//                                 [1] ctrl mask over the EXTENDED thread index (tid | (~tid & 511) << 9): a bit below 9 wants
//                                     that thread bit set, bit 9 + u wants thread bit u clear
//                                 [2] global index bits that must be set   [3] global index bits that must be clear
//          Entries are numbered in the order they appear here (the SCHEDULE ORDER); the matrix of scheduled entry s
//          of an evaluation lives at mats[mat_base + 8 s].
// ANGLE TABLE: 9 words per entry {p_theta, p_phi, p_lambda (int32; <0 = literal), theta, phi, lambda (3 doubles)}:
//          first the n_factors factors of the scheduled entries (entry by entry in schedule order, the factors of one
//          entry in the order they act), then the fold entries.
// CHAIN INDEX: per scheduled entry one word, first factor (index into the angle table) | number of factors << 24: the
//          entry's matrix is the product of its factors' matrices, the one that acts first rightmost.
// FOLD INDEX: per qubit q two words {first fold entry (index into the angle table), count}: the u gates folded
//          into qubit q's initial factor, in program order.
constexpr uint32_t kCircuitHeaderWords = 8;
constexpr uint32_t kPassHeaderWords = 4;
constexpr uint32_t kMaxTileBits = 13, kMaxThreadBits = 9, kMaxRegBits = 4;
constexpr uint32_t kColumnWords = kMaxThreadBits + kMaxRegBits;                 // one layout's columns
constexpr uint32_t kPassLoadColsOffset = kPassHeaderWords + kMaxTileBits;       // 17
constexpr uint32_t kPassStoreColsOffset = kPassLoadColsOffset + kColumnWords;   // 30
constexpr uint32_t kMaxCompactBits = 8, kMaxOuterBits = 20;
constexpr uint32_t kPassCompactOffset = kPassStoreColsOffset + kColumnWords;    // 43: positions of the control qubits
constexpr uint32_t kPassCompactWCols = kPassCompactOffset + kMaxCompactBits;    // 51
constexpr uint32_t kPassCompactFCols = kPassCompactWCols + kColumnWords;        // 64
constexpr uint32_t kPassCompactWBase = kPassCompactFCols + kColumnWords;        // 77
constexpr uint32_t kPassCompactFBase = kPassCompactWBase + kMaxOuterBits;       // 97
constexpr uint32_t kPassRoundsOffset = kPassCompactFBase + kMaxOuterBits;       // 117
constexpr uint32_t kPassCompactStore = 1u, kPassCompactLoad = 2u;               // flags word
constexpr uint32_t kExchangeWords = 2 * kColumnWords;                           // after the round's header word
constexpr uint32_t kMaxSwaps = 4, kSwapPad = 0xFFFFFFFFu;                       // swap round: kMaxSwaps words
constexpr int kSwapLaneLo = 0, kSwapLaneHi = 6;                                 // lane bits a swap may use: [lo, hi)
constexpr uint32_t kPosPad = 62;  // inserting a zero bit at position 62 leaves every index below 2^62 unchanged
constexpr uint32_t kGateWords = 4;
constexpr uint32_t kGateGeneral = 1u << 24, kGateNegated = 1u << 25;  // flags in the first word of a gate entry
constexpr uint32_t kMaxChain = 6;                         // factors per scheduled entry
constexpr uint32_t kAngleEntryWords = 9;
constexpr uint32_t kPlanPadWords = 32;  // readable padding after every plan (the kernel prefetches one gate ahead)

struct AngleSource {
    int32_t p_theta, p_phi, p_lambda;
    double theta, phi, lambda;
};

CircuitPlan build_plan(int n_qubits, const std::vector<GateIn>& gates, const std::vector<AngleSource>& op_angles,
                       const PlanConfig& cfg);

}  // namespace qsv
