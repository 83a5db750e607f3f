// Device-side data structures and kernel launch wrappers shared between kernels.hip and qsv_api.cpp.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace qsv {

// One evaluation (circuit + parameter vector) inside a launch group.
struct EvalDesc {
    uint32_t plan_base;   // word offset of the circuit plan in the plan arena
    uint32_t mat_base;    // offset (in doubles) of this evaluation's matrix region (mat_region_doubles below)
    uint32_t state_slot;  // which resident state buffer the evaluation uses
    uint32_t out_index;   // row of `partials` / entry of the result vector
    uint32_t param_base;  // offset (in doubles) of this evaluation's parameter vector in the parameter buffer
    uint32_t n_params;
    uint32_t flags;       // kEval* below
    uint32_t split_base;  // word offset (plan arena) of the split block (kEvalSide descriptors, see split.hpp)
};
// A split evaluation (split.hpp) has TWO descriptors, one per virtual circuit: side A where an ordinary evaluation has
// its own (region 0), side B in a second region of the descriptor array, which holds kEvalNull entries for everybody
// else.  A side's pass kernel stores its final state in the side's half of the evaluation's compact-table slot.
constexpr uint32_t kEvalSide = 1u, kEvalSideB = 2u, kEvalNull = 4u;
// (split evaluations) both virtual circuits are one pass: under a quadratic diagonal operator the kernel that runs them (one
// workgroup per side, sweeping the side's tiles one after the other) also forms their weighted Gram matrices and combines
// them (kModeFusedFactor) -- one launch per push
constexpr uint32_t kEvalFused = 8u;
// The circuit continues a KEPT state (qsv.h: qsv_prefix_create -- the state a layer search's evaluations have in common,
// reference mutation.py:57-59): its plan is unfolded, its first pass runs the later-pass instantiation and reads slot
// `split_base` of PassArgs::prefix_states instead of the evaluation's own state slot (which it then writes as usual).
constexpr uint32_t kEvalPrefix = 16u;
// (with kEvalFused) a three-key side of thirteen virtual qubits runs as TWO workgroups, one per value of its third key qubit --
// the plan's one qubit outside its 12-qubit tile, never a target -- four product terms (rows) each; each exports the other half of x
// of its rows through memory, imports the partner's rows for its own half of x and forms ALL Gram sums over that half; four partial
// tables meet at the hand-off (kernels.hip fused_factor_tail).  Worth it only with eight amplitudes per thread (R = 3): a gate
// phase is one wave's issue time over its own amplitudes, and a half side of 4096 is then eight waves of eight.
constexpr uint32_t kEvalHalves = 32u;
// A virtual circuit (a side's own qubits + one per key) may be this many qubits larger than a tile: it then takes the pass
// kernel a few passes over up to 16 tiles -- nothing next to what a split evaluation saves.  (With + 2 only, populations of
// 26 and 28 qubits mostly found no split form: 21 k and 2 k evaluations per second against 550 k at 24 qubits.)
constexpr int kSideExtraBits = 4;
constexpr int kSideMaxOwnBits = 16;  // ... and a side's own qubits (launch_factor keeps one weight per bit)
// split block (what the contraction kernel needs to know about one split circuit).  The roles of the index bits do not
// depend on the circuit: bits 0 .. 5 are the lanes, the next W the wave index inside a workgroup (W = log2(threads / 64)),
// the next kSplitLoopBits a thread's own bits (it walks their 32 combinations itself), the rest the chunk number.  Per
// circuit: which side each of them belongs to, as ready-made pieces of the two table indices.
//   [0] number of keys K  [1] index bits of side X's table (without the keys)  [2] of side Y's
//   [3] bit 0: X is side B (else A); bits 8..: LX, how many of the thread's own bits belong to side X (0 .. 2: X is the
//       side with fewer of them)
//   [4 .. 9)    thread's own bit b (X's bits first, then Y's): its bit in the table index of the side it belongs to
//   [9 .. 14)   ... and its value in the full index (1 << position)
//   [14] the qubits of side X as a mask of the full index  [15] of side Y (the split sampler deposits table indices there)
//   [16 .. 144)  lane l -> (x piece, y piece) of index bits 0 .. 5 = l
//   [144 .. 160) wave w -> pieces of the wave-index bits
//   [160 .. 416) chunk & 127 -> pieces of the low seven chunk bits      [416 .. 672) chunk >> 7 -> pieces of the rest
//   [672] where side X's OWN values of D start in PassArgs::side_diag (in doubles; kNoSideDiag: none) [673] side Y's.  Entry x of
//       such a table is D[x deposited in the side's qubits]: what a side's Gram sums read.  Gathered from D they are one cache line
//       per value -- a 12-qubit side's 4096 lines, times the sixteen sides an XCD serves, do not fit its L2 (measured: a zero-key
//       circuit split 8 + 12 took 39 us where one split 10 + 10 took 31; with the values as one run 34).  Written by the host when
//       the plan is uploaded (qsv_api.hip upload_plans), filled by side_diag_kernel right behind the copy.
constexpr uint32_t kNoSideDiag = 0xffffffffu;
constexpr uint32_t kSplitLoopCols = 4, kSplitLoopPos = 9, kSplitMaskX = 14, kSplitMaskY = 15, kSplitLaneTable = 16, kSplitWaveTable = 144,
                   kSplitChunkLow = 160, kSplitChunkHigh = 416, kSplitSideDiag = 672, kSplitBlockWords = 674;
constexpr int kSplitLoopBits = 5, kSplitMaxLoopX = 2;

enum PassMode : uint32_t {
    kModeSynthFirst = 1u,  // pass 0 synthesises |0..0> instead of reading the state
    kModeFinalStore = 2u,  // the last pass writes the state back
    kModeFinalDiag = 4u,   // the last pass reduces sum_i |a_i|^2 D[i] into `partials`
    kModeDirectResult = 32u, // every evaluation of the launch is ONE workgroup (gridDim.x = 1, one tile): the fused last pass
                             // adds its waves' sums itself (fixed order) and writes result_out[out_index]; no partials
    kModeFinalProbs = 64u,   // the last pass writes |a_i|^2 (fp64) to `partials` used as [state_slot][2^n] (the sampler's input)
                             // instead of the state (n <= 28)
    kModeStreaming = 16u,    // the states do not fit the Infinity Cache: non-temporal state loads and stores
    kModeSidesOnly = 128u,   // split evaluations: run the two virtual circuits, no contraction (the split sampler follows)
    kModeFusedFactor = 256u, // (pass 0 only) split evaluations flagged kEvalFused: each side's workgroup goes on to the weighted Gram
                             // matrices of its final state (what launch_factor's first kernel computes), hands them over
                             // through PassArgs::factor_scratch, and the side that finishes second combines them and writes
                             // result_out[out_index] -- no further launch for these evaluations
    kModeFusedLdsTable = 512u, // (with kModeFusedFactor, fp64) the launch has the LDS for it (kFusedLdsTableEnd) and at most one
                               // workgroup per CU anyway: sides of up to kFusedLdsTableBits qubits hand their state to the
                               // Gram matrices through LDS.  Same values, same order of every sum: a launch may choose.
    kModeTileMajor = 1024u,  // (later passes) grid = (evaluations, tile chunks) instead of (tile chunks, evaluations): the workgroups
                             // of ONE tile of every state are dispatched together (the last pass's reads of the diagonal table then
                             // meet in the memory-side cache); measurement knob QSV_TILE_MAJOR
    kModeFusedPrepare = 8u,  // (pass 0 only) every workgroup first does prepare_kernel's work for its evaluation, reading
                             // the descriptor from host_evals (PassArgs below); no prepare launch ran for these evaluations
};

struct PassArgs {
    const uint32_t* plan;  // plan arena
    const double* mats;    // matrix regions (see EvalDesc::mat_base); 8 doubles per gate: m00 m01 m10 m11 as (re, im)
    const EvalDesc* evals; // blockIdx.y indexes this array
    void* states;          // slot s starts at s * state_stride amplitudes
    void* wtab;            // compact tables (plan.hpp COMPACT): slot s starts at s * wtab_stride amplitudes
    const double* diag;    // D[i] for the diagonal fast path (may be null)
    double* partials;      // [out_index][partial_chunks][waves of a workgroup]
    uint64_t state_stride;
    uint64_t wtab_stride;
    uint32_t pass_index;
    uint32_t mode;
    uint32_t tiles_per_block;  // consecutive tiles each workgroup sweeps
    uint32_t region_stride;    // gridDim.z = 2: blockIdx.z = 1 takes descriptor evals[region_stride + blockIdx.y] (side B of
                               // split evaluations, kEvalNull for the others)
    // kModeFusedPrepare: where prepare_kernel would read and write (indexed like `evals`)
    const EvalDesc* host_evals;  // pinned host memory
    EvalDesc* evals_out;         // device copy of the descriptors (the contraction kernel reads it)
    const double* host_params;   // pinned host memory
    double* mats_out;            // = mats
    double* result_out;          // kModeDirectResult: one double per evaluation (pinned host memory)
    uint32_t partial_chunks;   // workgroup slots per evaluation in `partials` (>= gridDim.x; 0 means gridDim.x): launches
                               // of one batch may use different grids (pass 0 / later passes), the reducer sees one shape
    // kModeFusedFactor: what launch_factor takes (quad: n_full x n_full couplings; scratch: factor_slot_doubles() per
    // side-table slot) and one counter per side-table slot (zero before the handle's first launch; every evaluation adds two)
    const double* quad;
    double* factor_scratch;
    uint32_t* factor_counters;
    uint32_t n_full;
    const void* prefix_states;  // kEvalPrefix: kept states, slot s at s * state_stride amplitudes (may be null otherwise)
    uint32_t dephase;           // (measurement knob QSV_DEPHASE) odd workgroups of a later pass sleep this many s_sleep(127) first
    const double* side_diag;    // the sides' own values of D (split block word kSplitSideDiag; may be null when no block names one)
};
// LDS bytes the fused factor tail of a pass launch needs (up to eight waves form a side's Gram matrices)
constexpr size_t kFusedFactorLdsBytes = 8 * (18 * 64 + 64) * sizeof(double) + 64;
// kModeFusedLdsTable: a side of at most kFusedLdsTableBits qubits keeps its final state in LDS, behind the tail's own scratch,
// for the Gram matrices to read -- it is neither stored to the side table nor fetched back
constexpr int kFusedLdsTableBits = 12;
constexpr size_t kFusedLdsTableOffset = ((kFusedFactorLdsBytes + 1023) / 1024) * 1024;
constexpr size_t kFusedLdsTableEnd = kFusedLdsTableOffset + (size_t(16) << kFusedLdsTableBits);  // 141 KiB
// ... and a THREE-KEY side of kFusedLdsRowsBits virtual qubits (eight product terms of 2^10 amplitudes: the largest side the
// one-launch route takes) keeps its state in LDS too, FROM OFFSET 0 -- under the tail's scratch, which is only written once
// the Gram sums are done -- as eight rows with one amplitude of padding between them, so that the lanes of the eight-term
// Gram body read their two rows' amplitudes straight from it without bank conflicts: no staging, nothing stored or fetched.
constexpr int kFusedLdsRowsBits = 13, kFusedLdsRowsKeys = 3;
constexpr size_t kFusedLdsRowPitch = (size_t(1) << (kFusedLdsRowsBits - kFusedLdsRowsKeys)) + 1;  // amplitudes
constexpr size_t kFusedLdsRowsDstage = ((kFusedLdsRowPitch * 8 * 16 + 127) / 128) * 128;      // [wave][64] values of D
constexpr size_t kFusedLdsRowsEnd = kFusedLdsRowsDstage + 8 * 64 * sizeof(double) + 64;
// (kEvalHalves: four own rows from offset 0, pitch kFusedLdsRowPitch; behind them the partner's four half rows, one amplitude apart)
constexpr size_t kFusedHalvesImport = kFusedLdsRowPitch * 4 * 16;
constexpr size_t kFusedHalvesImportPitch = (size_t(1) << (kFusedLdsRowsBits - kFusedLdsRowsKeys - 1)) + 1;  // amplitudes
static_assert(kFusedHalvesImport + kFusedHalvesImportPitch * 4 * 16 <= kFusedLdsRowsDstage, "the imported rows end before the staged values of D");
// hand-off counters per side-table slot: [0] the evaluation's (it grows by four per evaluation: a side's one workgroup adds two, a
// half side's one), [1 + side] the exchange of a side's two halves (each adds one when its rows are out)
constexpr uint32_t kFactorCountersPerSlot = 4;
static_assert(kFusedLdsRowsEnd <= kFusedLdsTableEnd, "a launch with kModeFusedLdsTable has the LDS for either form");

// Diagnostic stamps (only in a -DQSV_STAMPS build): [pass][phase] shader cycles summed over WAVES; the last phase
// slot counts waves.  Phases: 0 setup, 1 load / synthesis, 2 xor-column setup + wait for the previous exchange's
// readers, 3 write re, 4 barrier, 5 read re, 6 barrier, 7 write im, 8 barrier, 9 read im (exchange mode 2 only),
// 10 gates, 11 store / reduce, 12 epilogue.
constexpr int kStampPasses = 8, kStampPhases = 16;
// A batch of sides' own tables of D: out[base + x] = diag[x deposited in mask], x < 2^bits (upload_plans)
struct SideDiagJob {
    uint32_t base, mask, bits;
};
constexpr int kSideDiagJobsPerLaunch = 128;
struct SideDiagJobs {
    SideDiagJob job[kSideDiagJobsPerLaunch];
};
hipError_t launch_side_diag(const double* diag, double* side_diag, const SideDiagJobs& jobs, int n_jobs, hipStream_t stream);
hipError_t read_stamps(unsigned long long* out, int reset);  // hipErrorNotSupported in the shipped build
// (-DQSV_TIMELINE builds: when each tile's phases began, per workgroup of a later pass)
constexpr unsigned kTimelineWgs = 16384, kTimelineTiles = 14, kTimelineWords = 6 + 4 * kTimelineTiles + 2;
hipError_t read_timeline(unsigned long long* out, size_t max_words, unsigned int* n_records, int reset);

// doubles an evaluation's matrix region occupies for a circuit with n_real scheduled gates in n_passes passes on n qubits
// (gate matrices | 4 doubles per qubit: initial factors | kMatPadDoubles | thread factors: 2 * 2^t | tile factors:
// 2 * 2^(n-k) | tile info: n_passes * 2 * 2^(n-k)); t = thread bits, n - k = bits outside a tile.  prepare_kernel
// fills all of it: the synthesis tables for pass 0 and, per pass and tile, one TileInfo record, so that a workgroup
// of the pass kernel fetches what depends on its tile number with ONE scalar load instead of decoding it (every
// wave redid ~190 scalar and ~40 vector instructions per tile for that, and the masks it kept cost ~40 SGPRs).
constexpr uint32_t kMatPadDoubles = 16;
// LDS bytes a pass launch with kModeFusedPrepare needs at least (prepare_eval's scratch, kernels.hip)
constexpr size_t kFusedPrepareLdsBytes = (4 * 32 + 1024 + 8 * 128 + 6 * 256) * sizeof(double);
struct TileInfo {     // 16 bytes = 2 doubles
    uint32_t base_lo, base_hi;  // amplitude index of the tile's element 0: the tile number spread over the outer
                                // qubits (a compact pass 0: the pattern number spread over its control qubits)
    uint32_t wbase, fbase;      // COMPACT_LOAD pass: the tile's part of the W index / of the tile-factor index
};
inline size_t mat_region_doubles(uint32_t n_real, uint32_t n_qubits, int thread_bits, int outer_bits, int n_passes) {
    return size_t(8) * n_real + size_t(4) * n_qubits + kMatPadDoubles + (size_t(2) << thread_bits) +
           (size_t(2) << outer_bits) + size_t(n_passes > 0 ? n_passes : 0) * (size_t(2) << outer_bits);
}
// offset (in doubles, from the start of the region) of pass p's TileInfo table
inline __host__ __device__ size_t tile_info_offset(uint32_t n_real, uint32_t n_qubits, int thread_bits, int outer_bits, uint32_t p) {
    return size_t(8) * n_real + size_t(4) * n_qubits + kMatPadDoubles + (size_t(2) << thread_bits) +
           (size_t(2) << outer_bits) + size_t(p) * (size_t(2) << outer_bits);
}

// Angles -> gate matrices, initial product-state factors and synthesis tables, one workgroup per evaluation.
// host_evals / params are pinned host memory read by the kernel; evals receives the device copy of the descriptors.
// n_regions = 2: also descriptors [region_stride + i] (the second descriptors of split evaluations).
// dtype != 0 (single precision): the gate matrices are left as floats (kernels.hip prepare_eval, float_mats).
hipError_t launch_prepare(const uint32_t* plan, const EvalDesc* host_evals, EvalDesc* evals, const double* params,
                          double* mats, int n_evals, hipStream_t stream, int n_regions = 1, uint32_t region_stride = 0, int dtype = 0);

// Split evaluations (split.hpp): <psi|D|psi> with psi[i] = sum_kappa A_kappa[a(i)] B_kappa[b(i)] formed on the fly from
// the two side tables; grid and partial-sum layout as the pass kernel's fused last pass (PassArgs: plan, evals, wtab,
// wtab_stride, diag, partials, partial_chunks; state_stride = 2^n).  Descriptors without kEvalSide return at once.
// n_chunks workgroups per evaluation (n_chunks * threads * 32 = 2^n), n_evals evaluations.
hipError_t launch_contract(int dtype, unsigned n_chunks, unsigned n_evals, int threads, hipStream_t stream, const PassArgs& args);

// Split evaluations under a QUADRATIC diagonal operator (every term has at most two Z factors: Ising / QUBO operators,
// which is what the reference's problem encoders produce): <psi|D|psi> from the two side tables alone, no sweep over
// the 2^n indices.  With b_q = (1 - z_q) / 2 the bit of qubit q,
//     D(x, y) = D(x, 0) + D(0, y) - D(0, 0) + 4 sum_{a in X, b in Y} J_ab b_a(x) b_b(y)
// and for psi = sum_j X_j (x) Y_j every product term factorises: <psi| f(x) g(y) |psi> = sum_{j'j} F[j'j] G[j'j] with the
// weighted Gram matrices F[j'j] = sum_x conj(X_j'[x]) f(x) X_j[x] (J x J, Hermitian).  Two launches: the Gram matrices
// of both sides for the weights 1, D(., 0) and the |side| bits (factor_moments_kernel: eight workgroups per evaluation,
// fixed assignment of blocks of 64 table entries to waves, partial sums added in a fixed order), then their combination
// (factor_combine_kernel: one workgroup per evaluation, writes result_out[out_index]).  Work per evaluation about
// (|X| / 2 + 6) 2^|X| J^2 per side.
// quad: n x n doubles, quad[a * n + b] = J_ab (symmetric, zero diagonal).  PassArgs: plan, evals (device descriptors,
// side A region), wtab / wtab_stride (side tables), diag, result_out.  scratch: factor_slot_doubles() per side-table slot
// (indexed by the descriptors' state_slot).
constexpr size_t factor_slot_doubles() { return size_t(2) * 4 * 18 * 64; }
// Evaluations with four or five cut keys (16 / 32 product terms) take the same two launches with a thread per matrix entry
// (entry groups x slices x sides workgroups per evaluation): most_keys = the most keys of the range, scratch_big =
// factor_big_slot_doubles() per side-table slot (may be null when most_keys <= 3).
// big_counters: factor_big_slot_counters() uint32 per side-table slot, zero before the first launch (the workgroups of one
// (side, entry group) count themselves: the last one adds the slices' partial sums, in slice order).
size_t factor_big_slot_doubles();
size_t factor_big_slot_counters();
hipError_t launch_factor(int dtype, unsigned n_evals, double* scratch, const double* quad, int n_qubits, hipStream_t stream,
                         const PassArgs& args, double* scratch_big = nullptr, uint32_t* big_counters = nullptr, int most_keys = 3);

// Split evaluations under ANY Pauli operator: for psi = sum_j X_j (x) Y_j and a Pauli string P = P_X (x) P_Y
//     <psi|P|psi> = sum_{j'j} <X_j'|P_X|X_j> <Y_j'|P_Y|Y_j>
// -- two J x J matrices per term, each a sum over ONE side table (a Pauli string maps index u to u ^ f with a sign
// (-1)^popcount(u & z) and a global power of i).  factor_terms_kernel: a wave per term (terms dealt out over
// kFactorTermWaves waves per evaluation), both matrices, their pairing, times the coefficient; the waves' sums leave as
// partial sums, [out_index][kFactorTermWaves], for reduce_partials_kernel.  Work per evaluation and term
// (2^|X| + 2^|Y|) J^2 -- against a sweep over 2^n amplitudes per group of terms.
struct FactorTerm {
    uint32_t x, z;  // masks of the full index
    double coeff;   // real part of the coefficient (the expectation of a Pauli string is real)
};
constexpr uint32_t kFactorTermWaves = 64;
// PassArgs: plan, evals (device descriptors, side A region), wtab / wtab_stride (side tables).
hipError_t launch_factor_terms(int dtype, unsigned n_evals, const FactorTerm* terms, uint32_t n_terms, double* partials,
                               hipStream_t stream, const PassArgs& args);

// dtype: 0 = fp64, 1 = fp32.  r = register bits (1..4).  xmode = LDS exchange mode (see kernels.hip).
// Returns hipSuccess or the launch error.
hipError_t launch_pass(int dtype, int r, int xmode, dim3 grid, int threads, size_t lds_bytes, hipStream_t stream,
                       const PassArgs& args);
hipError_t configure_pass_kernels(int dtype, int r, int xmode, size_t lds_bytes);

hipError_t launch_diag_table(int n_qubits, int n_terms, const uint64_t* z_mask, const double* coeff, double* table,
                             hipStream_t stream);

// out[e] = sum_b partials[e * blocks + b]   (fixed-order tree: bitwise reproducible)
// evals != null: the n_evals evaluations are evals[i].out_index (partials and out are then indexed by that number)
hipError_t launch_reduce_partials(const double* partials, uint32_t blocks, int n_evals, double* out,
                                  hipStream_t stream, const EvalDesc* evals = nullptr);

// Off-diagonal Pauli terms grouped by x mask (see kernels.hip).
struct PauliGroup {
    uint64_t x;       // common x mask (never 0)
    uint32_t first;   // first term of the group in the term arrays
    uint32_t count;
    uint32_t pivot;   // highest set bit of x
    uint32_t pad;
};
hipError_t launch_pauli_groups(int dtype, const void* states, uint64_t state_stride, int n_qubits, int n_slots,
                               int n_groups, const PauliGroup* groups, const uint64_t* term_z,
                               const double* term_coef, const uint32_t* term_odd, int nb, double* partials,
                               hipStream_t stream);
hipError_t launch_pauli_combine(const double* partials, uint32_t per_slot, const double* diag_partials,
                                uint32_t diag_per_eval, int n_slots, const EvalDesc* evals, double* out,
                                hipStream_t stream);

// For each of n_slots probability vectors (slot s at probs + s * dim, need not be normalised) draw `shots` basis
// states; evaluation (first_eval + s) gets its own random stream and writes out[(first_eval + s) * shots ..].
// chunk_sums: scratch of n_slots * sample_chunk_count(dim) doubles.  With diag != null, out_values receives D[state].
// evals != null: slot s belongs to evaluation evals[s].out_index instead (a batch that put its split evaluations first).
hipError_t launch_sample(const double* probs, uint64_t dim, int n_slots, double* chunk_sums, int shots, uint64_t seed,
                         uint32_t first_eval, const double* diag, uint64_t* out, double* out_values,
                         hipStream_t stream, const EvalDesc* evals = nullptr);
uint32_t sample_chunk_count(uint64_t dim);

// Sampling a split evaluation (split.hpp) WITHOUT forming its 2^n probabilities.  With psi(x, y) = sum_j X_j[x] Y_j[y]
// the probability of (x, y) summed over any set S of y values is a quadratic form in the J values X_j[x] whose matrix
// is the Gram matrix of the Y_j restricted to S.  So: Gram matrices of Y per block of 64 consecutive y (y1 = y >> 6),
// their sum -> the marginal of x and its running sums (launch_split_tables, one workgroup per evaluation); then per shot
// x by binary search, y1 by its quadratic forms (one candidate per lane), the last six bits of y by the amplitudes of
// that block themselves (launch_split_sample, one wave per shot).  Work per evaluation 2^|X| J^2 + 2^|Y| J^2 / 2 + shots
// * (J^2 + 64 J), against 2^n for the probabilities; the samples come in a different order of the index space than
// launch_sample's (x-major), so the same seed gives different -- equally distributed -- samples.
// PassArgs: plan, evals (device descriptors of the group, side A region), wtab / wtab_stride (side tables).
// scratch: n_evals * split_sample_slot_doubles(side_bits) doubles, side_bits = the most qubits a virtual circuit may have.
size_t split_sample_slot_doubles(int side_bits);
hipError_t launch_split_tables(int dtype, int side_bits, unsigned n_evals, double* scratch, hipStream_t stream, const PassArgs& args);
hipError_t launch_split_sample(int dtype, int side_bits, unsigned n_evals, const double* scratch, int shots, uint64_t seed,
                               const double* diag, uint64_t* out, double* out_values, hipStream_t stream, const PassArgs& args,
                               uint32_t table_doubles = 0);  // (the largest Gram table of the launch, 0 = not known)

// out[first_eval + e] = CVaR_alpha of values[e * shots .. (e + 1) * shots) for e < n_evals (shots <= kCvarMaxShots):
// bitonic sort in LDS, fixed-order sum of the lowest alpha * shots values (the boundary value weighted fractionally).
constexpr int kCvarMaxShots = 4096;
hipError_t launch_cvar(const double* values, int n_evals, int shots, double alpha, double* out, hipStream_t stream);

// ---- exact-probability CVaR (the sampler branch without sampling noise) ---------------------------------------------------
// order[j] = the basis state of rank j when the states are sorted by their value under the diagonal operator (ties in index
// order), sorted_values[j] = that value (sort.hip, once per operator).
hipError_t sort_states_by_value(const double* values, uint64_t dim, uint32_t* order, double* sorted_values, hipStream_t stream);
// CVaR_alpha of the EXACT distribution |a_i|^2 over the operator's values, as the reference's accumulation loop computes it
// from a measured distribution (queasars/circuit_evaluation/expectation_calculation.py:14-32): states in ascending order of
// value, probability mass gathered until numpy.isclose(gathered, alpha) (rtol 1e-5, atol 1e-8), the last state's mass
// clipped to what is missing, the sum divided by alpha.  The probabilities of evaluation e come from probs[slot e][2^n]
// (what a gate pass with kModeFinalProbs leaves), or -- descriptors with kEvalSide -- from the two side tables of a split
// circuit (|sum_j X_j[x(i)] Y_j[y(i)]|^2, formed per state; at most three keys).  Two launches: sums per chunk of
// kCvarChunk ranks (mass and mass x value; fixed order), then per evaluation the chunk in which the mass is reached and
// the rank inside it.  PassArgs: plan, evals (the group's device descriptors), wtab / wtab_stride (side tables).
// chunk_scratch: 2 * n_evals * cvar_exact_chunks(dim) doubles.  out[evals[e].out_index] receives the result.
constexpr uint32_t kCvarChunk = 4096;
inline uint32_t cvar_exact_chunks(uint64_t dim) { return uint32_t((dim + kCvarChunk - 1) / kCvarChunk); }
hipError_t launch_cvar_exact(int dtype, const double* probs, uint64_t dim, unsigned n_evals, const uint32_t* order,
                             const double* sorted_values, double alpha, double* chunk_scratch, double* out, hipStream_t stream,
                             const PassArgs& args);

// ---- the optimiser's share of a lock-step SPSA iteration (qsv.h: qsv_spsa_step) -----------------------------------------------
// One workgroup per run: accept the iteration whose two values are in `values` (update x, count, stopping rules), then write
// the two points of the next iteration.  Either half may be left out (values / delta_propose null).
struct SpsaStepArgs {
    int n_runs, width;
    double* x;                    // [n_runs][width]
    unsigned char* active;        // [n_runs]
    long long* iterations;        // [n_runs]
    const double* delta_accept;   // [n_runs][width], the signs the values were measured with
    const double* values;         // [2 n_runs]: f(x + eps delta), f(x - eps delta) per run
    const double* delta_propose;  // [n_runs][width]
    double* points;               // [2 n_runs][width]
    double eps, lr;
    int trust_region, maxiter;
    int window;                   // termination rule: allowed_consecutive_violations + 1, 0 = no rule
    double min_rel;
    long long maxfev;             // < 0: none
    double* previous;             // [n_runs]
    long long* n_values;          // [n_runs]
    double* changes;              // [n_runs][window]
};
hipError_t launch_spsa_step(const SpsaStepArgs& args, hipStream_t stream);

hipError_t launch_probabilities(int dtype, const void* state, uint64_t dim, int n_slots, double* probs,
                                hipStream_t stream);
hipError_t launch_state_to_f64(int dtype, const void* state, uint64_t dim, double* out_re_im, hipStream_t stream);

}  // namespace qsv
