// HIP kernels for gfx950 (MI355X).  See plan.hpp for the pass/round/exchange model and the plan encoding,
// DESIGN.md section 4 for measurements.
//
// pass_kernel: one workgroup owns one tile of 2^k amplitudes at a time for a whole pass.
//   input             : pass 0 SYNTHESISES the initial product state from per-thread and per-tile factor tables
//                       (prepare_kernel); a compact pass 0 does so only for one tile per pattern of its outer control
//                       qubits.  Later passes load 2^R amplitudes per thread (16 B per lane per instruction, >= 64 B
//                       contiguous runs) -- or, behind a compact pass 0, build them from two cache-resident tables.
//   rounds            : 2x2 butterflies between registers of one thread; for fp64 the whole gate loop is a generated
//                       assembly block (gate_loop_gen.inc): in-place v_fma_f64, matrices in scalar registers
//   exchanges         : tile transposed through LDS, real plane then imaginary plane, under a host-chosen XOR swizzle
//                       that makes both sides bank-conflict free; exchanges that stay inside a wave run barrier-free
//   output            : same layout as the load; or, on the last pass of an evaluation with a diagonal operator,
//                       no store at all: sum_i |a_i|^2 D[i] is reduced on chip and one double per workgroup leaves
// No MFMA: 2x2 gates; fp64 MFMA has the vector rate on this part anyway.
#include "kernels.hpp"
#include "plan.hpp"

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

namespace qsv {

template <typename real>
struct alignas(2 * sizeof(real)) cx {
    real re, im;
};

// Plan words, gate matrices and evaluation descriptors are written by earlier launches and are constant while a pass
// runs.  Reading them through constant-address-space pointers is what makes hipcc use SCALAR loads for them: through
// ordinary global pointers every read that follows a barrier or a state store is a vector load with its own
// s_waitcnt vmcnt(0), and the column loops below turn into chains of serialized L2 round trips.
#if defined(__HIP_DEVICE_COMPILE__)
#define QSV_CONST_AS __attribute__((address_space(4)))
#else
#define QSV_CONST_AS  // the host pass only parses the kernels
#endif
using cu32p = const QSV_CONST_AS uint32_t*;
using cf64p = const QSV_CONST_AS double*;
template <typename T>
__device__ __forceinline__ const QSV_CONST_AS T* as_constant(const T* p) {
    return (const QSV_CONST_AS T*)(p);
}

// N consecutive plan words into scalar registers with wide loads (x4 chunks, 4-byte aligned is enough for s_load)
// that are all issued before the first use.
template <int N>
__device__ __forceinline__ void load_words(cu32p p, uint32_t (&w)[N]) {
    typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
    for (int i = 0; i + 4 <= N; i += 4) {
        const u32x4a v = *(const QSV_CONST_AS u32x4a*)(p + i);
        w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w;
    }
#pragma unroll
    for (int i = N & ~3; i < N; ++i) w[i] = p[i];
}

// Offset of a thread inside a layout: XOR of the columns selected by the bits of tid.  The block always holds
// kMaxThreadBits columns (unused ones are 0), so there is nothing to predicate and the loop unrolls.  Lane bits cost
// two VALU operations per column (bit -> mask, mask & column ^ x); the wave-index bits are uniform, so their columns
// are folded on the scalar unit.
__device__ __forceinline__ uint32_t xor_columns(cu32p cols, uint32_t tid, uint32_t wave) {
    uint32_t c[kMaxThreadBits];
    load_words<int(kMaxThreadBits)>(cols, c);
    uint32_t x = 0;
#pragma unroll
    for (int u = 6; u < int(kMaxThreadBits); ++u) x ^= ((wave >> (u - 6)) & 1u) ? c[u] : 0u;
#pragma unroll
    for (int u = 0; u < 6; ++u) x ^= uint32_t(int32_t(tid << (31 - u)) >> 31) & c[u];
    return x;
}

// Visit the 2^R register indices in Gray-code order: consecutive indices differ in one bit, so the running offset
// needs one XOR with one column (a scalar) per step and no 2^R-entry offset table has to stay live in SGPRs.
//   for (int i = 0; i < NR; ++i) { off = gray_step<R>(i, off, cols); use(gray_index(i), off); }
__device__ __forceinline__ constexpr int gray_index(int i) { return i ^ (i >> 1); }
__device__ __forceinline__ uint32_t gray_step(int i, uint32_t off, cu32p reg_cols) {
    return i == 0 ? off : off ^ reg_cols[__builtin_ctz(i)];
}

// 2x2 butterfly with a u-type matrix (m00 is real: 14 multiply-adds per amplitude pair instead of 16).
// m = {m00, Re m01, Im m01, Re m10, Im m10, Re m11, Im m11}.  J = target register bit, C = control register bit
// (-1: none): both compile-time, so a controlled gate touches exactly the 2^(R-2) pairs it must and there is no
// per-pair branching.  The operation order lets every result be written by its last FMA straight into the
// register that held the input it replaces (a1 is consumed first, then a0), so no v_mov_b64 is needed -- on
// gfx950 a 64-bit move costs as much issue time as a v_fma_f64.
template <typename real, int R, int J, int C>
__device__ __forceinline__ void butterfly(cx<real> (&amp)[1 << R], const real (&m)[7]) {
    constexpr int tbit = 1 << J;
    constexpr int cbit = C >= 0 ? (1 << (C >= 0 ? C : 0)) : 0;
#pragma unroll
    for (int e0 = 0; e0 < (1 << R); ++e0) {
        if ((e0 & tbit) || (e0 & cbit) != cbit) continue;
        const real a0r = amp[e0].re, a0i = amp[e0].im, a1r = amp[e0 | tbit].re, a1i = amp[e0 | tbit].im;
        real u = m[1] * a1r;   // a1 part of the new a0
        real w = m[1] * a1i;
        real p = m[5] * a1r;   // a1 part of the new a1
        real q = m[5] * a1i;
        u = fma(-m[2], a1i, u);
        w = fma(m[2], a1r, w);
        p = fma(-m[6], a1i, p);
        q = fma(m[6], a1r, q);   // last use of the old a1
        p = fma(m[3], a0r, p);
        q = fma(m[3], a0i, q);
        amp[e0 | tbit].re = fma(-m[4], a0i, p);
        amp[e0 | tbit].im = fma(m[4], a0r, q);
        amp[e0].re = fma(m[0], a0r, u);
        amp[e0].im = fma(m[0], a0i, w);
    }
}

// The same for an entry with flags (plan.hpp FUSION): a general matrix (m00 complex: 16 operations) on the pairs the
// entry's pair mask names.  Only the C++ gate loop (fp32, diagnostic builds) comes here; fp64 runs gate_loop_gen.inc.
template <typename real, int R, int J>
__device__ __forceinline__ void butterfly_general(cx<real> (&amp)[1 << R], const real (&m)[8], uint32_t pair_mask) {
    constexpr int tbit = 1 << J;
    int pair = 0;
#pragma unroll
    for (int e0 = 0; e0 < (1 << R); ++e0) {
        if (e0 & tbit) continue;
        if ((pair_mask >> pair) & 1u) {
            const real a0r = amp[e0].re, a0i = amp[e0].im, a1r = amp[e0 | tbit].re, a1i = amp[e0 | tbit].im;
            amp[e0].re = m[0] * a0r - m[1] * a0i + m[2] * a1r - m[3] * a1i;
            amp[e0].im = m[0] * a0i + m[1] * a0r + m[2] * a1i + m[3] * a1r;
            amp[e0 | tbit].re = m[4] * a0r - m[5] * a0i + m[6] * a1r - m[7] * a1i;
            amp[e0 | tbit].im = m[4] * a0i + m[5] * a0r + m[6] * a1i + m[7] * a1r;
        }
        ++pair;
    }
}
template <typename real, int R, int J>
struct GeneralDispatch {
    static __device__ __forceinline__ void run(int j, cx<real> (&amp)[1 << R], const real (&m)[8], uint32_t pair_mask) {
        if (j == J)
            butterfly_general<real, R, J>(amp, m, pair_mask);
        else
            GeneralDispatch<real, R, J - 1>::run(j, amp, m, pair_mask);
    }
};
template <typename real, int R>
struct GeneralDispatch<real, R, -1> {
    static __device__ __forceinline__ void run(int, cx<real> (&)[1 << R], const real (&)[8], uint32_t) {}
};

// sel = J * (R + 1) + (C + 1)
template <typename real, int R, int SEL>
struct ButterflyDispatch {
    static __device__ __forceinline__ void run(int sel, cx<real> (&amp)[1 << R], const real (&m)[7]) {
        if (sel == SEL) {
            constexpr int J = SEL / (R + 1), C = SEL % (R + 1) - 1;
            if constexpr (C != J) butterfly<real, R, J, C>(amp, m);
        } else {
            ButterflyDispatch<real, R, SEL - 1>::run(sel, amp, m);
        }
    }
};
template <typename real, int R>
struct ButterflyDispatch<real, R, -1> {
    static __device__ __forceinline__ void run(int, cx<real> (&)[1 << R], const real (&)[7]) {}
};

#ifdef QSV_GATE_LOOP_INC  // (timing experiments, scripts/ablate.py: a variant of the generated block)
#include QSV_GATE_LOOP_INC
#else
#include "gate_loop_gen.inc"
#endif

// ---- lane swaps (plan.hpp, "swap" rounds) ---------------------------------------------------------------------
// Transposition of the tile bit under register bit V with the one under lane bit U: element (lane, e) with lane bit U
// != register bit V of e trades places with (lane ^ 2^U, e ^ 2^V).  For a register pair A = amp[e0] (bit V clear),
// B = amp[e0 | 2^V] that is: A's lanes with bit U set receive B from their partner lane, B's lanes with bit U clear
// receive A from theirs -- per 32-bit half
//   U = 5: v_permlane32_swap_b32 A, B  (lanes 32-63 of vdst trade with lanes 0-31 of src)
//   U = 4: v_permlane16_swap_b32 A, B  (odd rows of vdst trade with even rows of src; a row = 16 lanes)
//   U = 3, 2: two DPP moves, a row shift by 8 / 4 lanes each way under a bank mask that selects the receiving lanes
//   U = 1, 0: two DPP quad permutes (every lane reads its partner) and two selects
// No LDS, no barrier; 16 (U >= 4) to about 64 vector instructions per transposition at 8 fp64 amplitudes per thread.
template <int U>
__device__ __forceinline__ void swap_words(uint32_t& a, uint32_t& b) {
    if constexpr (U == 5) {
        const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        a = r[0];
        b = r[1];
    } else if constexpr (U == 4) {
        const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        a = r[0];
        b = r[1];
    } else if constexpr (U <= 1) {
        // quad_perm [1,0,3,2] / [2,3,0,1]: every lane reads its partner; the lanes with bit U set keep their b and
        // take the partner's b into a, the others keep a and take the partner's a into b
        constexpr int ctrl = U == 0 ? 0xB1 : 0x4E;
        const uint32_t pb = uint32_t(__builtin_amdgcn_update_dpp(0, int(b), ctrl, 0xF, 0xF, true));
        const uint32_t pa = uint32_t(__builtin_amdgcn_update_dpp(0, int(a), ctrl, 0xF, 0xF, true));
        const bool upper = (__lane_id() >> U) & 1u;
        a = upper ? pb : a;
        b = upper ? b : pa;
    } else {
        // row_shr:n (0x110 + n): lane i reads lane i - n of its row; row_shl:n (0x100 + n): lane i reads lane i + n.
        // bank b of a row = its lanes 4b .. 4b+3: lane bit 2 = bank bit 0, lane bit 3 = bank bit 1.
        constexpr int n = 1 << U;                         // 4 or 8
        constexpr int upper = U == 2 ? 0xA : 0xC;         // banks whose lanes have bit U set
        constexpr int lower = U == 2 ? 0x5 : 0x3;
        const uint32_t na = uint32_t(__builtin_amdgcn_update_dpp(int(a), int(b), 0x110 + n, 0xF, upper, false));
        const uint32_t nb = uint32_t(__builtin_amdgcn_update_dpp(int(b), int(a), 0x100 + n, 0xF, lower, false));
        a = na;
        b = nb;
    }
}

template <int U>
__device__ __forceinline__ void swap_reals(double& a, double& b) {
    uint32_t alo = uint32_t(__double2loint(a)), ahi = uint32_t(__double2hiint(a));
    uint32_t blo = uint32_t(__double2loint(b)), bhi = uint32_t(__double2hiint(b));
    swap_words<U>(alo, blo);
    swap_words<U>(ahi, bhi);
    a = __hiloint2double(int(ahi), int(alo));
    b = __hiloint2double(int(bhi), int(blo));
}
template <int U>
__device__ __forceinline__ void swap_reals(float& a, float& b) {
    uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
    swap_words<U>(x, y);
    a = __uint_as_float(x);
    b = __uint_as_float(y);
}

template <typename real, int R, int V, int U>
__device__ __forceinline__ void swap_reg_lane(cx<real> (&amp)[1 << R]) {
#pragma unroll
    for (int e0 = 0; e0 < (1 << R); ++e0) {
        if (e0 & (1 << V)) continue;
        swap_reals<U>(amp[e0].re, amp[e0 | (1 << V)].re);
        swap_reals<U>(amp[e0].im, amp[e0 | (1 << V)].im);
    }
}

// sel = V * 6 + U
template <typename real, int R, int SEL>
struct SwapDispatch {
    static __device__ __forceinline__ void run(int sel, cx<real> (&amp)[1 << R]) {
        if (sel == SEL)
            swap_reg_lane<real, R, SEL / 6, SEL % 6>(amp);
        else
            SwapDispatch<real, R, SEL - 1>::run(sel, amp);
    }
};
template <typename real, int R>
struct SwapDispatch<real, R, -1> {
    static __device__ __forceinline__ void run(int, cx<real> (&)[1 << R]) {}
};

// Diagnostic build only (-DQSV_STAMPS, scripts/stamps.sh): per-phase shader-cycle counters of the pass kernel,
// summed over waves.  The shipped library compiles every QSV_STAMP to nothing.
#ifdef QSV_STAMPS
#ifdef QSV_STAMPS_WAVE0  // (only the first wave of every workgroup is counted: the critical path where the others wait at barriers)
#define QSV_STAMP_WAVE_OK (tid < 64u)
#else
#define QSV_STAMP_WAVE_OK true
#endif
__device__ unsigned long long qsv_stamp_table[kStampPasses * kStampPhases];
#define QSV_STAMP_DECL unsigned long long st_acc[kStampPhases] = {}; unsigned long long st_last = qsv_stamp_now();
// the waves' counters summed in LDS first: one global atomic per phase per workgroup (row `row` of the table)
#define QSV_STAMP_FLUSH(row) do { \
        unsigned long long* tab = reinterpret_cast<unsigned long long*>(lds_raw); \
        __syncthreads(); \
        if (tid < kStampPhases) tab[tid] = 0; \
        __syncthreads(); \
        if ((tid & 63u) == 0 && QSV_STAMP_WAVE_OK) { \
            for (int ph = 0; ph < kStampPhases - 1; ++ph) atomicAdd(&tab[ph], st_acc[ph]); \
            atomicAdd(&tab[kStampPhases - 1], 1ull); \
        } \
        __syncthreads(); \
        if (tid < kStampPhases && (row) < kStampPasses) atomicAdd(&qsv_stamp_table[(row) * kStampPhases + tid], tab[tid]); \
        __syncthreads(); \
    } while (0)
#define QSV_STAMP(ph) do { const unsigned long long st_t = qsv_stamp_now(); st_acc[ph] += st_t - st_last; st_last = st_t; } while (0)
static __device__ __forceinline__ unsigned long long qsv_stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#else
#define QSV_STAMP_DECL
#define QSV_STAMP(ph)
#endif

// Diagnostic build only (-DQSV_TIMELINE, scripts/timeline.py): the first wave of every workgroup of a LATER pass writes down
// WHEN each of its tiles' phases began (s_memtime, nothing drained: the waits the production kernel has are the only ones) and
// on which compute unit it ran -- so that the phases of the workgroups sharing a compute unit can be laid side by side.
// Record of a workgroup: [0] HW_ID | XCC_ID << 32, [1] block | pass << 32, [2] tiles, [3] s_memrealtime at its start (100 MHz),
// [4] s_memrealtime at its end, [5] s_memtime at its end, then per tile: loads issued from, loads back at, rounds done at,
// stores issued at.
#ifdef QSV_TIMELINE
__device__ unsigned long long qsv_timeline[kTimelineWgs * kTimelineWords];
__device__ unsigned int qsv_timeline_count;
static __device__ __forceinline__ unsigned long long qsv_time_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
static __device__ __forceinline__ unsigned long long qsv_realtime_now() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define QSV_TL(slot) do { if (tl_rec && j < kTimelineTiles && threadIdx.x == 0) tl_rec[6 + 4 * j + (slot)] = qsv_time_now(); } while (0)
#else
#define QSV_TL(slot)
#endif

// 16-byte (8-byte) store that leaves no dirty line in L2 (sc1: write-through); the compiler does not count it, the caller
// waits for it with s_waitcnt vmcnt(0).
__device__ __forceinline__ void store_through(void* p, const cx<double>& v) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 bits = {uint32_t(__double2loint(v.re)), uint32_t(__double2hiint(v.re)), uint32_t(__double2loint(v.im)),
                        uint32_t(__double2hiint(v.im))};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(bits) : "memory");
}
__device__ __forceinline__ void store_through(void* p, const cx<float>& v) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 bits = {__float_as_uint(v.re), __float_as_uint(v.im)};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(bits) : "memory");
}

// Workgroup barrier for the LDS exchanges.  __syncthreads() carries a release fence over GLOBAL memory as well: hipcc
// puts s_waitcnt vmcnt(0) in front of s_barrier, so every wave would sit out the full latency of the previous
// tile's state stores at the next barrier.  Nothing a workgroup writes to global memory is read by the same launch,
// so only LDS traffic has to be ordered here.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Read-only inputs are separate `const __restrict__` kernel parameters (not members of a by-value struct): that is
// what lets hipcc prove they cannot alias the state stores and fetch plan words and matrices with SCALAR loads.
struct PassScalars {
    uint64_t state_stride;
    uint64_t wtab_stride;  // amplitudes per state slot in the compact-table buffer
    uint32_t pass_index;
    uint32_t mode;
    uint32_t tiles_per_block;
    uint32_t partial_chunks;
    uint32_t region_stride;
    const EvalDesc* host_evals;
    EvalDesc* evals_out;
    const double* host_params;
    double* mats_out;
    double* result_out;
    const double* quad;
    double* factor_scratch;
    uint32_t* factor_counters;
    uint32_t n_full;
    const void* prefix_states;
    uint32_t dephase;
    const double* side_diag;
};

// XMODE selects how a tile is transposed through LDS:
//   0  one complex element per access (ds_*_b128 for fp64, ds_*_b64 for fp32); LDS = 2^k * sizeof(complex)
//   1  real and imaginary planes, both resident (ds_*_b64 for fp64);           LDS = 2^k * sizeof(complex)
//   2  real plane then imaginary plane through ONE plane buffer;                LDS = 2^k * sizeof(real)
// Mode 2 halves the LDS footprint (32 KiB at k = 12, fp64) so three or four 512-thread workgroups fit a CU.
// Occupancy is what sets the v_fma_f64 issue rate on gfx950 (measured with scripts/ubench/valu_rate.hip:
// 13 / 7.2 / 5.8 / 4.7 cycles per instruction at 1 / 2 / 4 / 8 waves per SIMD), so the kernel is compiled for
// 6 waves per SIMD (<= 80 VGPRs) in mode 2: three 512-thread workgroups per CU.  Four (8 waves per SIMD, 64 VGPRs)
// fit too since the gate loop is assembly, but measured 3-4% slower (twice): fewer scalar registers, more spills.
#ifndef QSV_WAVES_PER_SIMD
#define QSV_WAVES_PER_SIMD 6
#endif
#ifndef QSV_WAVES_R4
#define QSV_WAVES_R4 4
#endif
#ifndef QSV_WAVES_R4_FIRST
#define QSV_WAVES_R4_FIRST 4
#endif
template <int R, int XMODE, bool FIRST, bool FUSED = false>
struct Occupancy {
    // 2^R amplitudes = 4 * 2^R VGPRs: R = 3 fits the 80-VGPR budget of 6 waves per SIMD, R = 4 needs the 128 of 4 --
    // (the synthesising pass 0 has no load phase and would fit the 96 of 5 waves; measured: no faster)
    // (FUSED: the one-launch route's instantiation -- two workgroups per evaluation on a chip of 256 CUs, one workgroup per CU
    // by its LDS anyway: its tail, the sides' Gram sums, may take the registers of two waves per SIMD)
    static constexpr int waves_per_simd =
        FUSED ? 2 : R >= 4 ? (FIRST && XMODE == 2 ? QSV_WAVES_R4_FIRST : QSV_WAVES_R4) : (XMODE == 2 ? QSV_WAVES_PER_SIMD : 4);
};

template <typename T>
struct Log2Size;
template <> struct Log2Size<float> { static constexpr int value = 2; };
template <> struct Log2Size<double> { static constexpr int value = 3; };

// ---- angles -> matrices ---------------------------------------------------------------------------------
struct Angles {
    double theta, phi, lam;
};

__device__ __forceinline__ Angles read_angles(const uint32_t* __restrict__ e, const double* __restrict__ params) {
    const int32_t pt = int32_t(e[0]), pf = int32_t(e[1]), pl = int32_t(e[2]);
    Angles a;
    a.theta = pt >= 0 ? params[pt] : __hiloint2double(int(e[4]), int(e[3]));
    a.phi = pf >= 0 ? params[pf] : __hiloint2double(int(e[6]), int(e[5]));
    a.lam = pl >= 0 ? params[pl] : __hiloint2double(int(e[8]), int(e[7]));
    return a;
}

// Qiskit UGate: [[cos(t/2), -e^{i lam} sin(t/2)], [e^{i phi} sin(t/2), e^{i(phi+lam)} cos(t/2)]]
__device__ __forceinline__ void u_matrix(const Angles& a, double* m) {
    double s, c, sl, cl, sp, cp;
    sincos(a.theta * 0.5, &s, &c);
    sincos(a.lam, &sl, &cl);
    sincos(a.phi, &sp, &cp);
    // e^{i(phi+lam)} from the two factors (a fourth sincos costs more than everything else in this function)
    const double cpl = cp * cl - sp * sl, spl = sp * cl + cp * sl;
    m[0] = c;        m[1] = 0.0;
    m[2] = -cl * s;  m[3] = -sl * s;
    m[4] = cp * s;   m[5] = sp * s;
    m[6] = cpl * c;  m[7] = spl * c;
}

// One angle-table entry -> matrix.  Entries with p_theta below -1 are the fixed matrices of split.hpp's virtual circuits.
__device__ __forceinline__ void entry_matrix(const uint32_t* __restrict__ e, const double* __restrict__ params, double* m) {
    const int32_t code = int32_t(e[0]);
    if (code >= -1) {
        u_matrix(read_angles(e, params), m);
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = 0.0;
    if (code == -2) {         // projector on |0>
        m[0] = 1.0;
    } else if (code == -3) {  // |0> -> |0> + |1>
        m[0] = 1.0;
        m[4] = 1.0;
    } else {                  // X
        m[2] = 1.0;
        m[4] = 1.0;
    }
}

// The same from sines and cosines somebody else has computed: trig = {sin, cos} of theta / 2, of phi, of lambda (prepare_eval
// spreads the sincos calls over the workgroup, three threads per entry; the arithmetic below is u_matrix's, to the bit).
__device__ __forceinline__ void entry_matrix_trig(const uint32_t* __restrict__ e, const double* trig, double* m) {
    if (int32_t(e[0]) < -1) {
        entry_matrix(e, nullptr, m);  // (a fixed matrix: no angles)
        return;
    }
    const double s = trig[0], c = trig[1], sp = trig[2], cp = trig[3], sl = trig[4], cl = trig[5];
    const double cpl = cp * cl - sp * sl, spl = sp * cl + cp * sl;
    m[0] = c;        m[1] = 0.0;
    m[2] = -cl * s;  m[3] = -sl * s;
    m[4] = cp * s;   m[5] = sp * s;
    m[6] = cpl * c;  m[7] = spl * c;
}

// Matrix of one scheduled entry: the product of its factors (plan.hpp CHAIN INDEX), the factor that acts first rightmost.
// trig (may be null): 6 doubles per angle-table entry, see entry_matrix_trig.
__device__ __forceinline__ void chain_matrix(const uint32_t* __restrict__ table, uint32_t chain_word,
                                             const double* __restrict__ params, double* m, const double* trig = nullptr) {
    const uint32_t first = chain_word & 0xffffffu, count = chain_word >> 24;
    if (trig)
        entry_matrix_trig(table + size_t(first) * kAngleEntryWords, trig + size_t(first) * 6, m);
    else
        entry_matrix(table + size_t(first) * kAngleEntryWords, params, m);
    for (uint32_t i = 1; i < count; ++i) {
        double b[8], c[8];
        if (trig)
            entry_matrix_trig(table + size_t(first + i) * kAngleEntryWords, trig + size_t(first + i) * 6, b);
        else
            entry_matrix(table + size_t(first + i) * kAngleEntryWords, params, b);
#pragma unroll
        for (int row = 0; row < 2; ++row)
#pragma unroll
            for (int col = 0; col < 2; ++col) {
                // c[row][col] = b[row][0] m[0][col] + b[row][1] m[1][col]
                const double xr = b[4 * row], xi = b[4 * row + 1], yr = b[4 * row + 2], yi = b[4 * row + 3];
                const double pr = m[2 * col], pi = m[2 * col + 1], qr = m[4 + 2 * col], qi = m[4 + 2 * col + 1];
                c[4 * row + 2 * col] = xr * pr - xi * pi + yr * qr - yi * qi;
                c[4 * row + 2 * col + 1] = xr * pi + xi * pr + yr * qi + yi * qr;
            }
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = c[e];
    }
}

// What prepare_kernel does for ONE evaluation, by one workgroup (every thread of it calls this; `scratch` = LDS for
// kPrepScratchDoubles doubles).  The pass kernel's synthesising instantiation runs it itself for the virtual circuits of split
// evaluations (one launch and one dependent launch latency less in front of the contraction).
// LDS scratch of prepare_eval: the qubits' initial factors, the evaluation's parameter vector and the matrices of the
// folded gates.
constexpr uint32_t kPrepMaxParams = 1024, kPrepMaxFold = 128, kPrepMaxTrig = 256;
constexpr uint32_t kPrepScratchDoubles = 4 * 32 + kPrepMaxParams + 8 * kPrepMaxFold + 6 * kPrepMaxTrig;  // 29 KiB
static_assert(kPrepScratchDoubles * sizeof(double) == kFusedPrepareLdsBytes, "kernels.hpp: kFusedPrepareLdsBytes");

#ifdef QSV_STAMPS  // (diagnostic build: phases 2 .. 8 of the stamp table take prepare_eval's steps)
#define QSV_PSTAMP(ph) do { if (st_acc) { const unsigned long long st_t = qsv_stamp_now(); st_acc[ph] += st_t - *st_last; *st_last = st_t; } } while (0)
#define QSV_PSTAMP_ARGS , st_acc, st_last
#define QSV_PSTAMP_PARAMS , unsigned long long* st_acc = nullptr, unsigned long long* st_last = nullptr
#define QSV_PSTAMP_PARAMS_DEF , unsigned long long* st_acc, unsigned long long* st_last
#else
#define QSV_PSTAMP(ph)
#define QSV_PSTAMP_ARGS
#define QSV_PSTAMP_PARAMS
#define QSV_PSTAMP_PARAMS_DEF
#endif
// float_mats: the scheduled entries' matrices are left as eight FLOATS at the start of their 64-byte records (single-precision
// handles: the round loops read them as they are, instead of rounding eight doubles per gate, wave and tile)
__device__ __forceinline__ void prepare_eval(const uint32_t* __restrict__ plan, const EvalDesc& ev,
                                             const double* __restrict__ params, double* __restrict__ mats,
                                             double* scratch, bool float_mats QSV_PSTAMP_PARAMS) {
    double* sv = scratch;                   // initial factors (v0, v1) of every qubit, n <= 32
    double* sp = scratch + 4 * 32;          // the parameter vector
    double* fm = sp + kPrepMaxParams;       // matrices of the fold entries
    const uint32_t* __restrict__ cp = plan + ev.plan_base;
    const uint32_t n_passes = cp[0], n_real = cp[1], n_qubits = cp[2];
    const uint32_t* __restrict__ table = cp + cp[3];
    const uint32_t* __restrict__ fold = cp + cp[4];
    const uint32_t n_fold = cp[5];
    const uint32_t* __restrict__ chains = cp + cp[6];
    const uint32_t n_factors = cp[7];  // the fold entries follow the scheduled entries' factors in the angle table
    const double* p = params + ev.param_base;
    double* __restrict__ out = mats + ev.mat_base;
    // The parameters live in pinned host memory: every read is a trip over PCIe, and a qubit's folded gates used to be
    // worked through one after the other, each waiting for its own angles and then for its sincos -- 13 of the kernel's
    // 18 microseconds.  Now: the whole vector comes over in one go, every matrix (scheduled gates and folded ones) is
    // computed by its own thread, and the per-qubit loop only multiplies 2x2 matrices.
    const bool staged = ev.n_params <= kPrepMaxParams && n_fold <= kPrepMaxFold;
    QSV_PSTAMP(2);  // plan header
    if (staged) {
        for (uint32_t i = threadIdx.x; i < ev.n_params; i += blockDim.x) sp[i] = p[i];
        __syncthreads();
        p = sp;
    }
    QSV_PSTAMP(3);  // parameters staged
    // One sincos per thread: a gate's matrix takes three (theta / 2, phi, lambda), an entry that is a product of matrices
    // (plan.hpp FUSION) up to nine, and they were the longest stretch of this function -- 9 % of the one-launch route's kernel.
    // Thread 3 f + a takes angle a of angle-table entry f; the matrices are then put together from the table in LDS.
    double* trig = fm + 8 * kPrepMaxFold;
    const bool trig_staged = staged && n_factors + n_fold <= kPrepMaxTrig;
    if (trig_staged) {
        for (uint32_t i = threadIdx.x; i < 3u * (n_factors + n_fold); i += blockDim.x) {
            const uint32_t f = i / 3u, which = i - 3u * f;
            const uint32_t* __restrict__ e = table + size_t(f) * kAngleEntryWords;
            if (int32_t(e[0]) < -1) continue;  // (a fixed matrix)
            const int32_t pidx = int32_t(e[which]);
            double angle = pidx >= 0 ? p[pidx] : __hiloint2double(int(e[4 + 2 * which]), int(e[3 + 2 * which]));
            if (which == 0) angle *= 0.5;
            double sn, cs;
#ifndef QSV_ABL_PREP_TRIG
            sincos(angle, &sn, &cs);
#else
            sn = angle, cs = 1.0 - angle;  // (measurement: what the sines and cosines cost)
#endif
            trig[size_t(f) * 6 + 2 * which] = sn;
            trig[size_t(f) * 6 + 2 * which + 1] = cs;
        }
        __syncthreads();
    }
    for (uint32_t j = threadIdx.x; j < n_real + (staged ? n_fold : 0u); j += blockDim.x) {
        double m[8];
        if (j < n_real)
            chain_matrix(table, chains[j], p, m, trig_staged ? trig : nullptr);
        else if (trig_staged)
            entry_matrix_trig(table + size_t(n_factors + j - n_real) * kAngleEntryWords, trig + size_t(n_factors + j - n_real) * 6, m);
        else
            entry_matrix(table + size_t(n_factors + j - n_real) * kAngleEntryWords, p, m);
        double* dst = j < n_real ? out + size_t(j) * 8 : fm + size_t(j - n_real) * 8;
        if (float_mats && j < n_real) {
            float* dstf = reinterpret_cast<float*>(dst);
#pragma unroll
            for (int i = 0; i < 8; ++i) dstf[i] = float(m[i]);
#pragma unroll
            for (int i = 8; i < 16; ++i) dstf[i] = 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[i] = m[i];
        }
    }
    if (staged) __syncthreads();
    QSV_PSTAMP(4);  // matrices
    for (uint32_t q = threadIdx.x; q < n_qubits; q += blockDim.x) {
        const uint32_t first = fold[2 * q], count = fold[2 * q + 1];
        double v0r = 1.0, v0i = 0.0, v1r = 0.0, v1i = 0.0;
        for (uint32_t i = 0; i < count; ++i) {
            double m[8];
            if (staged) {
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fm[size_t(first + i - n_factors) * 8 + e];
            } else {
                entry_matrix(table + size_t(first + i) * kAngleEntryWords, p, m);
            }
            const double a0r = v0r, a0i = v0i, a1r = v1r, a1i = v1i;
            v0r = m[0] * a0r - m[1] * a0i + m[2] * a1r - m[3] * a1i;
            v0i = m[0] * a0i + m[1] * a0r + m[2] * a1i + m[3] * a1r;
            v1r = m[4] * a0r - m[5] * a0i + m[6] * a1r - m[7] * a1i;
            v1i = m[4] * a0i + m[5] * a0r + m[6] * a1i + m[7] * a1r;
        }
        double* o = out + size_t(n_real) * 8 + size_t(q) * 4;
        o[0] = v0r; o[1] = v0i; o[2] = v1r; o[3] = v1i;
        sv[4 * q] = v0r; sv[4 * q + 1] = v0i; sv[4 * q + 2] = v1r; sv[4 * q + 3] = v1i;
    }
    double* pad = out + size_t(n_real) * 8 + size_t(n_qubits) * 4;
    // zero the padding the pass kernel's one-gate-ahead prefetch may read
    for (uint32_t i = threadIdx.x; i < kMatPadDoubles; i += blockDim.x) pad[i] = 0.0;
    if (n_passes == 0) return;
    __syncthreads();
    QSV_PSTAMP(5);  // initial factors

    // Synthesis tables for pass 0 (the pass that writes the initial product state, amplitude(i) = prod_q v_q[i_q]):
    //   thread_factor[tid] = product over the tile qubits that pass 0's load layout keeps on thread bits
    //   tile_factor[tile]  = product over the qubits outside the tile
    // The pass kernel multiplies the two and expands the register-held qubits itself.
    // Pass 0's header block comes into SCALAR registers with wide loads (every use below has a compile-time subscript), and a
    // thread's factor is the product of ITS t factors, read from LDS all at once: the version before this one copied the block
    // to LDS behind a barrier and walked every qubit of the register per thread with run-time subscripts -- one dependent LDS
    // round trip after the other, 2.7 us of a one-launch evaluation's 36 (measured by leaving the tables out).
    uint32_t hw[kPassLoadColsOffset + kMaxThreadBits + 2];  // header, tile positions, thread columns of the load layout
    load_words<int(kPassLoadColsOffset + kMaxThreadBits + 2)>(as_constant(cp) + cp[kCircuitHeaderWords], hw);
    const uint32_t hdr = hw[0];
    const int k = hdr & 0xff, t = (hdr >> 16) & 0xff;
    uint32_t tile_mask = 0;
#pragma unroll
    for (int j = 0; j < int(kMaxTileBits); ++j)
        if (j < k) tile_mask |= 1u << hw[kPassHeaderWords + j];
    double* thread_factor = pad + kMatPadDoubles;
    double* tile_factor = thread_factor + (size_t(2) << t);
#ifndef QSV_ABL_PREP_TABLES  // (measurement: the launch without the synthesis tables -- wrong results, the time they cost)
    for (uint32_t i = threadIdx.x; i < (1u << t); i += blockDim.x) {
        // (a thread column is one bit: the position of the qubit thread bit u holds; ascending u, in every launch alike)
        double vr[kMaxThreadBits], vi[kMaxThreadBits];
#pragma unroll
        for (int u = 0; u < int(kMaxThreadBits); ++u) {
            vr[u] = 1.0;
            vi[u] = 0.0;
            if (u < t) {
                const double* v = sv + 4 * uint32_t(__builtin_ctz(hw[kPassLoadColsOffset + u])) + 2 * ((i >> u) & 1u);
                vr[u] = v[0];
                vi[u] = v[1];
            }
        }
        double fr = 1.0, fi = 0.0;
#pragma unroll
        for (int u = 0; u < int(kMaxThreadBits); ++u) {
            const double nr = fr * vr[u] - fi * vi[u];
            fi = fr * vi[u] + fi * vr[u];
            fr = nr;
        }
        thread_factor[2 * size_t(i)] = fr;
        thread_factor[2 * size_t(i) + 1] = fi;
    }
#endif
    const uint32_t all_qubits = n_qubits >= 32 ? 0xffffffffu : ((1u << n_qubits) - 1u);
    const uint32_t n_tiles = 1u << (n_qubits - uint32_t(k));
    // tile number -> the index with the tile's own bits clear (a zero inserted at every tile position, ascending)
    auto tile_base = [&](uint64_t base, const uint32_t (&w)[kPassLoadColsOffset + kMaxThreadBits + 2]) {
#pragma unroll
        for (int j = 0; j < int(kMaxTileBits); ++j)
            if (j < k) {
                const uint32_t ps = w[kPassHeaderWords + j];
                base = ((base >> ps) << (ps + 1)) | (base & ((uint64_t(1) << ps) - 1));
            }
        return base;
    };
#ifndef QSV_ABL_PREP_TABLES
    const uint32_t outside = all_qubits & ~tile_mask;
    for (uint32_t tile = threadIdx.x; tile < n_tiles; tile += blockDim.x) {
        double fr = 1.0, fi = 0.0;
        if (outside) {  // (a register of one tile has nothing outside it: its one factor is 1)
            const uint64_t base = tile_base(tile, hw);
            for (uint32_t m = outside; m; m &= m - 1u) {  // ascending qubits
                const uint32_t q = uint32_t(__builtin_ctz(m));
                const double* v = sv + 4 * q + 2 * ((base >> q) & 1u);
                const double nr = fr * v[0] - fi * v[1];
                fi = fr * v[1] + fi * v[0];
                fr = nr;
            }
        }
        tile_factor[2 * size_t(tile)] = fr;
        tile_factor[2 * size_t(tile) + 1] = fi;
    }
#endif
    QSV_PSTAMP(6);  // synthesis tables
    // Per pass and tile: what the pass kernel needs to know about its tile number (kernels.hpp TileInfo).
    TileInfo* info_all = reinterpret_cast<TileInfo*>(tile_factor + (size_t(2) << (n_qubits - uint32_t(k))));
    for (uint32_t p = 0; p < n_passes; ++p) {
        const uint32_t* __restrict__ ph = cp + cp[kCircuitHeaderWords + p];
        uint32_t pw[kPassLoadColsOffset + kMaxThreadBits + 2];  // (the pass's header and tile positions: scalar registers again)
        if (p == 0) {
#pragma unroll
            for (int i = 0; i < int(kPassLoadColsOffset + kMaxThreadBits + 2); ++i) pw[i] = hw[i];
        } else {
            load_words<int(kPassLoadColsOffset + kMaxThreadBits + 2)>(as_constant(ph), pw);
        }
        const uint32_t flags = pw[2];
        const bool cstore = flags & kPassCompactStore, cload = flags & kPassCompactLoad;
        const uint32_t count = cstore ? 1u << ((flags >> 8) & 0xffu) : n_tiles;
        TileInfo* info = info_all + size_t(p) * n_tiles;
        for (uint32_t tile = threadIdx.x; tile < count; tile += blockDim.x) {
            uint64_t base;
            if (cstore) {
                base = 0;
                for (uint32_t b = 0; b < kMaxCompactBits; ++b) base |= uint64_t((tile >> b) & 1u) << ph[kPassCompactOffset + b];
            } else {
                base = tile_base(tile, pw);
            }
            uint32_t wbase = 0, fbase = 0;
            if (cload)
                for (uint32_t b = 0; b < kMaxOuterBits; ++b)
                    if ((tile >> b) & 1u) {
                        wbase ^= ph[kPassCompactWBase + b];
                        fbase ^= ph[kPassCompactFBase + b];
                    }
            info[tile] = TileInfo{uint32_t(base), uint32_t(base >> 32), wbase, fbase};
        }
    }
    QSV_PSTAMP(7);  // tile info
}

// (defined with the factor kernels below) the tail of a pass-0 launch under kModeFusedFactor
template <typename real>
__device__ __forceinline__ void fused_factor_tail(const uint32_t* __restrict__ plan_arena, const EvalDesc& ev,
                                                  const cx<real>* __restrict__ slot_tables, uint64_t side_stride,
                                                  const double* __restrict__ diag, const PassScalars& a, unsigned char* lds,
                                                  uint32_t gram_waves, uint32_t lds_table, uint32_t halves_tile QSV_PSTAMP_PARAMS);

// FIRST = the pass synthesises its input (pass 0 of an evaluation from |0..0>): two instantiations, so that neither
// carries the other's load path through register allocation.
// FUSED (only with FIRST and R = 4): the instantiation the one-launch route of split evaluations runs (kModeFusedFactor) -- the
// only one that carries the sides' Gram sums and the combination behind the virtual circuits (fused_factor_tail): the other
// first-pass kernels are the leaner for it (a third fewer scalar spills), this one has registers to spare.
template <typename real, int R, int XMODE, bool FIRST, bool FUSED = false>
__global__ void __launch_bounds__(512, (Occupancy<R, XMODE, FIRST, FUSED>::waves_per_simd))
    pass_kernel(const uint32_t* __restrict__ plan_arena, const double* __restrict__ mats_all,
                const EvalDesc* __restrict__ evals, cx<real>* __restrict__ states, cx<real>* __restrict__ wtabs,
                const double* __restrict__ diag, double* __restrict__ partials, const PassScalars a) {
    using cxr = cx<real>;
    constexpr int NR = 1 << R;
    constexpr int ASH = Log2Size<real>::value + 1;         // log2 of an amplitude's bytes
    constexpr int LSH = XMODE == 0 ? ASH : ASH - 1;        // log2 of an LDS element's bytes
    extern __shared__ __align__(16) unsigned char lds_raw[];

    QSV_STAMP_DECL
#ifdef QSV_ABL_EMPTY  // (measurement: what a launch of the one-launch route costs with nothing in it)
    if constexpr (FUSED) return;
#endif
    // (kModeTileMajor: the two grid dimensions trade places)
    const bool tile_major = !FIRST && (a.mode & kModeTileMajor);
    const uint32_t block_x = tile_major ? blockIdx.y : blockIdx.x, grid_x = tile_major ? gridDim.y : gridDim.x;
    const uint32_t block_y = tile_major ? blockIdx.x : blockIdx.y;
    EvalDesc ev;
    const double* mats_base = mats_all;
    bool prepared_here = false;
    if constexpr (FIRST) prepared_here = a.mode & kModeFusedPrepare;
    if (prepared_here) {
        // This workgroup does prepare_kernel's work for its evaluation itself (virtual circuits of split evaluations:
        // one launch and its latency less in front of the contraction): descriptor and parameters from pinned host
        // memory, matrices and tables into the evaluation's region -- which this workgroup then reads back through the
        // scalar cache: its stores must have landed, stale lines must go, and no load below may be moved above this
        // point (the pointer the loads use is only known to the compiler from here on).
        const size_t slot = size_t(block_y) + size_t(blockIdx.z) * a.region_stride;
        ev = a.host_evals[slot];
        // (the device copy first, also of a null descriptor: a repeated batch reads its descriptors from that copy)
        if (threadIdx.x == 0 && block_x == 0) a.evals_out[slot] = ev;
        if (ev.flags & kEvalNull) return;
        {
            // a launch's grid is as wide as its largest evaluation: a workgroup beyond THIS evaluation's tiles leaves before
            // the preparation (its reads of parameters over PCIe, times the width of the grid, were most of such a launch)
            const uint32_t* c0 = plan_arena + ev.plan_base;
            const uint32_t* p0 = c0 + c0[kCircuitHeaderWords];
            const uint32_t f0 = p0[2];
            const uint32_t tiles0 = (f0 & kPassCompactStore) ? 1u << ((f0 >> 8) & 0xffu) : 1u << (c0[2] - (p0[0] & 0xffu));
            if (block_x >= tiles0) return;
            // (the one-launch route: a side's ONE workgroup sweeps all its tiles -- only a half side, kEvalHalves, has one per tile)
            if (FUSED && block_x > 0 && (ev.flags & kEvalFused) && !((ev.flags & kEvalHalves) && c0[2] == uint32_t(kFusedLdsRowsBits)) && (a.mode & kModeFusedFactor)) return;
        }
#ifdef QSV_STAMPS
        QSV_STAMP(0);  // descriptor
        prepare_eval(plan_arena, ev, a.host_params, a.mats_out, reinterpret_cast<double*>(lds_raw), std::is_same<real, float>::value, st_acc, &st_last);
#else
        prepare_eval(plan_arena, ev, a.host_params, a.mats_out, reinterpret_cast<double*>(lds_raw), std::is_same<real, float>::value);
#endif
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        __builtin_amdgcn_s_dcache_inv();
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("" : "+s"(mats_base)::"memory");
        QSV_STAMP(13);
#ifdef QSV_ABL_AFTER_PREP  // (measurement: descriptor + preparation alone)
        if constexpr (FUSED) return;
#endif
    } else {
        cu32p e = as_constant(reinterpret_cast<const uint32_t*>(evals + block_y + size_t(blockIdx.z) * a.region_stride));
        ev.plan_base = e[0];
        ev.mat_base = e[1];
        ev.state_slot = e[2];
        ev.out_index = e[3];
        ev.flags = e[6];
        ev.split_base = e[7];
    }
    if (ev.flags & kEvalNull) return;
    const bool side = ev.flags & kEvalSide;  // a virtual circuit of a split evaluation (split.hpp)
    cu32p cp = as_constant(plan_arena) + ev.plan_base;
    const uint32_t n_passes = cp[0];
    if (a.pass_index >= n_passes) return;
    const uint32_t n_real = cp[1], n_qubits = cp[2];
    cu32p pp = cp + cp[kCircuitHeaderWords + a.pass_index];
    const uint32_t hdr = pp[0];
    const int k = hdr & 0xff, t = (hdr >> 16) & 0xff, n_rounds = hdr >> 24;
    const uint32_t pass_flags = pp[2];
    const uint32_t tid = threadIdx.x;
    const uint32_t tid_ext = tid | (~tid & ((1u << kMaxThreadBits) - 1u)) << kMaxThreadBits;  // (gate predicates, plan.hpp)
    const bool all_active = blockDim.x == (1u << t);
    const bool active = tid < (1u << t);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wave_base = wave << 6;

    cu32p pos = pp + kPassHeaderWords;
    cu32p glr = pp + kPassLoadColsOffset + kMaxThreadBits;   // register columns of the load layout
    cu32p gsr = pp + kPassStoreColsOffset + kMaxThreadBits;  // ... of the store layout
    cu32p rounds0 = pp + kPassRoundsOffset;
    cf64p mats0 = as_constant(mats_base) + ev.mat_base + size_t(pp[1]) * 8;
    cf64p vecs = as_constant(mats_base) + ev.mat_base + size_t(n_real) * 8;

    constexpr bool synth = FIRST;  // the launcher picks FIRST = (pass_index == 0 && mode & kModeSynthFirst)
    const bool last = a.pass_index + 1 == n_passes;
    const bool do_probs = last && (a.mode & kModeFinalProbs) && !side;
    const bool do_store = (!last || (a.mode & kModeFinalStore) || side) && !do_probs;
    const bool do_diag = last && (a.mode & kModeFinalDiag) && !side;
    // the compact table of this state slot lives in its own buffer: pass 1 may already be storing the state while
    // other workgroups of the same launch still read the table
    cxr* __restrict__ wt0 = wtabs + uint64_t(ev.state_slot) * a.wtab_stride;
    // (a side of a split evaluation is one tile: its state goes to the side's half of the slot's table)
    cxr* __restrict__ st0 = side ? wt0 + ((ev.flags & kEvalSideB) ? a.wtab_stride >> 1 : 0)
                                 : states + uint64_t(ev.state_slot) * a.state_stride;
    // (a circuit that continues a kept state reads its first pass's input there, kernels.hpp kEvalPrefix)
    const cxr* __restrict__ ld0 = st0;
    if constexpr (!FIRST)
        if ((ev.flags & kEvalPrefix) && a.pass_index == 0)
            ld0 = static_cast<const cxr*>(a.prefix_states) + uint64_t(ev.split_base) * a.state_stride;
    // Global offsets inside a state are XORs of plan columns.  While a state's byte size fits 32 bits (n <= 28 in
    // fp64) they are kept as BYTE offsets in one 32-bit register per element next to a uniform tile pointer (one
    // v_xor per access, scalar-base addressing); larger states take the 64-bit path.
    const bool wide = n_qubits + uint32_t(ASH) > 32u;
    // COMPACT (plan.hpp): pass 0 computes one tile per pattern of its outer control qubits and stores them back to
    // back at the start of the state slot (the table W); pass 1 builds its input as W[..] * tile_factor[..].
    const bool cstore = synth && (pass_flags & kPassCompactStore);
    const bool cload = !FIRST && a.pass_index == 1 && (a.mode & kModeSynthFirst) && (pass_flags & kPassCompactLoad);
    const uint32_t total_tiles = cstore ? 1u << ((pass_flags >> 8) & 0xffu) : 1u << (n_qubits - uint32_t(k));
    // states larger than the Infinity Cache are streamed: non-temporal loads and stores of the state (measured on
    // single-gate sweeps: n = 26 / 27 +3.8 %, 5.23 -> 5.43 TB/s; at n = 24, where the state fits the cache, -28 %)
    const bool streaming = (a.mode & kModeStreaming) && !cload && !cstore && !side;
    // A side whose Gram matrices this workgroup forms itself (fused_factor_tail) stores its state WRITE-THROUGH: nobody but
    // this workgroup reads it, and left dirty in L2 the side tables of a launch (12 MB at 64 evaluations) are written back
    // when the kernel ends -- ten microseconds between the last workgroup and the host seeing the results.
#ifdef QSV_ABL_NO_THROUGH  // (measurement: the fused sides' states stored like any other)
    const bool through = false;
#else
    const bool through = FUSED && side && (ev.flags & kEvalFused) && (a.mode & kModeFusedFactor);
#endif
    // ... and a small enough side does not go to memory at all where the launch has the LDS for it (kModeFusedLdsTable): the
    // table is laid out in LDS exactly as it would be in its slot (a fused side is one tile: offsets inside the tile ARE table
    // indices), behind everything else this kernel keeps there
    // (a half side, kEvalHalves: a thirteen-qubit side of a circuit so flagged; each of its two workgroups holds a 12-qubit tile of it)
    const bool half_side = FUSED && through && (ev.flags & kEvalHalves) && n_qubits == uint32_t(kFusedLdsRowsBits);
    const bool table_in_lds = through && std::is_same<real, double>::value && (a.mode & kModeFusedLdsTable) &&
                              (n_qubits <= uint32_t(kFusedLdsTableBits) || (half_side && plan_arena[ev.split_base] < uint32_t(kFusedLdsRowsKeys)));
    // ... or, a three-key side of thirteen virtual qubits, as padded rows from offset 0 (kernels.hpp, kFusedLdsRowsBits)
    bool table_lds_rows = false;
    if constexpr (FUSED && std::is_same<real, double>::value)
        table_lds_rows = through && (a.mode & kModeFusedLdsTable) && n_qubits == uint32_t(kFusedLdsRowsBits) &&
                         plan_arena[ev.split_base] == uint32_t(kFusedLdsRowsKeys) &&
                         (uint32_t(k) == uint32_t(kFusedLdsRowsBits) || half_side);  // (ONE tile of thirteen qubits, or a half side's tile of twelve:
                                                                                    // a side swept as two tiles goes through its slot)

    const uint32_t tg = xor_columns(pp + kPassLoadColsOffset, tid, wave);
    const uint32_t sg = xor_columns(pp + kPassStoreColsOffset, tid, wave);

    // synthesis tables of this evaluation (see prepare_kernel): thread factors, then tile factors
    cf64p thread_factor = vecs + 4 * size_t(n_qubits) + kMatPadDoubles;
    cf64p tile_factor = thread_factor + (size_t(2) << t);
    // this pass's TileInfo table (prepare_kernel): one 16-byte record per tile number
    cu32p tile_info = reinterpret_cast<cu32p>(tile_factor + (size_t(2) << (n_qubits - uint32_t(k))) +
                                              size_t(a.pass_index) * (size_t(2) << (n_qubits - uint32_t(k))));
    // loaded once: a load inside the tile loop would make pass 0 wait for the previous tile's stores (same counter)
    double ttr = 1.0, tti = 0.0;
    if (synth && active) {
        ttr = thread_factor[2 * tid];
        tti = thread_factor[2 * tid + 1];
    }
    // Workgroup b sweeps tiles b, b + gridDim.x, b + 2 gridDim.x, ..: neighbouring workgroups (which run at the same
    // time) work on neighbouring tiles, and a compact pass 0 -- fewer tiles than the grid -- gives each working
    // workgroup a single tile instead of leaving half of them idle.
    // (the one-launch route's sides, half sides apart: workgroup 0 of the side takes every tile, whatever the grid's width)
    const bool sweeps = FUSED && side && (ev.flags & kEvalFused) && !half_side && (a.mode & kModeFusedFactor);
    const uint32_t tile0 = block_x, tile_step = sweeps ? 1u : grid_x;
    if (tile0 >= total_tiles) return;  // (uniform, before any barrier)
    const uint32_t n_tiles = sweeps ? total_tiles
                             : (total_tiles - tile0 + tile_step - 1) / tile_step < a.tiles_per_block
                                 ? (total_tiles - tile0 + tile_step - 1) / tile_step
                                 : a.tiles_per_block;
    // compact load: this thread's part of the W index and of the tile-factor index, the same for every tile
    uint32_t wthr = 0, fthr = 0;
    if (cload) {
        wthr = xor_columns(pp + kPassCompactWCols, tid, wave);
        fthr = xor_columns(pp + kPassCompactFCols, tid, wave);
    }
    cxr amp[NR];
    double acc = 0.0;
    uint32_t xflags = 0;         // the same two flags for the assembly round loop: bit 0 lds_dirty, bit 1 cross_pending
    const uint64_t active_mask = __ballot(active);
    bool lds_dirty = false;      // LDS holds exchange data some wave may still be reading
    bool cross_pending = false;  // ... and that wave may be another one (the last exchange crossed waves)

    if constexpr (!FIRST) {
        // (measurement: the two workgroups of a CU start half a tile apart, so that one loads while the other computes)
        if (a.dephase && (block_x & 1u))
            for (uint32_t i = 0; i < a.dephase; ++i) __builtin_amdgcn_s_sleep(127);
    }
#ifdef QSV_TIMELINE
    unsigned long long* tl_rec = nullptr;
    if constexpr (!FIRST) {
        uint32_t slot = 0;
        if (threadIdx.x == 0) slot = atomicAdd(&qsv_timeline_count, 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (threadIdx.x < 64 && slot < kTimelineWgs) {
            tl_rec = qsv_timeline + size_t(slot) * kTimelineWords;
            if (threadIdx.x == 0) {
                const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
                tl_rec[0] = uint64_t(hw) | uint64_t(xcc) << 32;
                tl_rec[1] = uint64_t(block_x) | uint64_t(a.pass_index) << 32 | uint64_t(block_y) << 40;
                tl_rec[2] = n_tiles;
                tl_rec[3] = qsv_realtime_now();
            }
        }
    }
#endif
    QSV_STAMP(0);
    for (uint32_t j = 0; j < n_tiles; ++j) {
        QSV_TL(0);
        // what depends on the tile number comes from prepare_kernel's table: one scalar load
        uint32_t ti[4];
        load_words<4>(tile_info + 4 * size_t(tile0 + j * tile_step), ti);
        const uint64_t base = uint64_t(ti[0]) | uint64_t(ti[1]) << 32;
        // Per-element offsets do not depend on the tile, so hipcc would compute all of them once, ahead of the tile
        // loop, and keep (in fact spill) 3 * 2^R registers for them.  Recomputing them costs one v_xor per access:
        // the opaque copies below stop the hoisting.
        uint32_t tgv = tg, sgv = sg;
        asm volatile("" : "+v"(tgv), "+v"(sgv));
        if constexpr (synth) {
            // initial product state: amplitude(i) = prod_q v_q[bit q of i].  prepare_kernel has multiplied out the
            // factors of the qubits outside the tile (one value per tile: tile_factor) and of the tile qubits held
            // by thread bits (one value per thread: thread_factor); the register qubits are expanded here.
            // (a compact pass 0 leaves the tile factor to pass 1)
            const double tfr = cstore ? 1.0 : tile_factor[2 * size_t(tile0 + j * tile_step)];
            const double tfi = cstore ? 0.0 : tile_factor[2 * size_t(tile0 + j * tile_step) + 1];
            amp[0].re = real(tfr * ttr - tfi * tti);
            amp[0].im = real(tfr * tti + tfi * ttr);
#pragma unroll
            for (int v = 0; v < R; ++v) {
                const uint32_t q = __builtin_ctz(glr[v]);
                const real v0r = real(vecs[4 * q]), v0i = real(vecs[4 * q + 1]);
                const real v1r = real(vecs[4 * q + 2]), v1i = real(vecs[4 * q + 3]);
#pragma unroll
                for (int e = 0; e < (1 << v); ++e) {
                    const cxr x = amp[e];
                    amp[e | (1 << v)].re = x.re * v1r - x.im * v1i;
                    amp[e | (1 << v)].im = x.re * v1i + x.im * v1r;
                    amp[e].re = x.re * v0r - x.im * v0i;
                    amp[e].im = x.re * v0i + x.im * v0r;
                }
            }
        } else {
            // Input of a later pass: the tile's amplitudes from the state -- or, behind a compact pass 0, the table
            // entry W[..] (same load code, other base / thread offset / register columns) times the tile factor F[..].
            const uint32_t wbase = ti[2], fbase = ti[3];  // (zero unless this is a COMPACT_LOAD pass)
            if (active) {
                if (!wide || cload) {
                    const unsigned char* tile = reinterpret_cast<const unsigned char*>(cload ? wt0 : ld0 + base);
                    cu32p rcols = cload ? pp + kPassCompactWCols + kMaxThreadBits : glr;
                    uint32_t ob = (cload ? wbase ^ wthr : tgv) << ASH;
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        if (i) ob ^= rcols[__builtin_ctz(i)] << ASH;
#ifndef QSV_ABL_NOLOAD
                        if (streaming) {
                            typedef real vec2 __attribute__((ext_vector_type(2)));
                            const vec2 v = __builtin_nontemporal_load(reinterpret_cast<const vec2*>(tile + ob));
                            amp[gray_index(i)].re = v.x;
                            amp[gray_index(i)].im = v.y;
                        } else {
                            amp[gray_index(i)] = *reinterpret_cast<const cxr*>(tile + ob);
                        }
#else
                        amp[gray_index(i)].re = real(ob);
                        amp[gray_index(i)].im = real(i);
#endif
                    }
                } else {
                    uint32_t off = tgv;
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        off = gray_step(i, off, glr);
                        amp[gray_index(i)] = ld0[base + off];
                    }
                }
#ifndef QSV_ABL_NOF
                if (cload) {
                    // times F: two halves (register budget), each walking its 2^(R-1) elements in Gray-code order
                    const unsigned char* ftab = reinterpret_cast<const unsigned char*>(
                        mats_base + ev.mat_base + size_t(n_real) * 8 + 4 * size_t(n_qubits) + kMatPadDoubles + (size_t(2) << t));
                    cu32p frc = pp + kPassCompactFCols + kMaxThreadBits;
                    const uint32_t fo = (fbase ^ fthr) << 4;
                    constexpr int HB = NR > 1 ? NR / 2 : 1;
#pragma unroll
                    for (int h = 0; h < NR / HB; ++h) {
                        uint32_t fh = h ? fo ^ (frc[R - 1] << 4) : fo;
                        double fr[HB], fi[HB];
#pragma unroll
                        for (int i = 0; i < HB; ++i) {
                            if (i) fh ^= frc[__builtin_ctz(i)] << 4;
                            fr[gray_index(i)] = *reinterpret_cast<const double*>(ftab + fh);
                            fi[gray_index(i)] = *reinterpret_cast<const double*>(ftab + fh + 8);
                        }
#pragma unroll
                        for (int e = 0; e < HB; ++e) {
                            const double wr = double(amp[h * HB + e].re), wi = double(amp[h * HB + e].im);
                            amp[h * HB + e].re = real(wr * fr[e] - wi * fi[e]);
                            amp[h * HB + e].im = real(wr * fi[e] + wi * fr[e]);
                        }
                    }
                }
#endif
                // Wait for the loads HERE, inside the branch: at the join with the synthesis path hipcc would otherwise
                // place this wait before the first use for both paths, and on the synthesis path (pass 0, which loads
                // nothing) it would then wait for the previous tile's STORES, which share the counter.
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            }
        }

        QSV_STAMP(1);
        QSV_TL(1);
        cu32p rp = rounds0;
        cf64p mp = mats0;
#if !defined(QSV_STAMPS) || defined(QSV_STAMPS_ASM)  // (QSV_STAMPS_ASM: stamps around the production block)
        // fp64, exchange mode 2 (the production configuration): every round of the tile is ONE generated assembly block
        // (gate_loop_gen.inc, RoundLoopF64).  The C++ loop below states the same thing and serves fp32, the other
        // exchange modes and the stamped diagnostic build.
        // (fp32 in exchange mode 0 likewise since round 4: RoundLoopF32, packed arithmetic)
        constexpr bool kAsmRounds = (std::is_same<real, double>::value && XMODE == 2 && R <= 4) ||
                                    (std::is_same<real, float>::value && XMODE == 0 && R <= 4);
#else
        constexpr bool kAsmRounds = false;
#endif
        if constexpr (kAsmRounds) {
            if (n_rounds > 0) {
                if constexpr (std::is_same<real, double>::value)
                    RoundLoopF64<R>::run(amp, rp, mp, uint32_t(n_rounds), uint32_t(base), tid_ext, wave, active_mask,
                                         uint32_t(uintptr_t(lds_raw)), xflags);
                else
                    RoundLoopF32<R>::run(amp, rp, mp, uint32_t(n_rounds), uint32_t(base), tid_ext, wave, active_mask,
                                         uint32_t(uintptr_t(lds_raw)), xflags);
            }
            QSV_STAMP(10);
        } else
        for (int m = 0; m < n_rounds; ++m) {
            const uint32_t rh = rp[0];
            const int n_gates = rh & 0xffff;
            rp += 1;
            if ((rh >> 16) & 1u) {
                cu32p wc = rp;
                cu32p rc = rp + kColumnWords;
                rp += kExchangeWords;
                // LDS byte offsets: the thread part once per exchange; the register part is walked in Gray-code
                // order in every phase (one v_xor with a scalar per element) instead of being kept in 2 * 2^R
                // registers, which the gate loop needs more
                const uint32_t wt = xor_columns(wc, tid, wave) << LSH, rt = xor_columns(rc, tid, wave) << LSH;
                cu32p wrc = wc + kMaxThreadBits;
                cu32p rrc = rc + kMaxThreadBits;
                // An intra-wave exchange (plan.hpp) moves data only inside each wave, through the LDS region that
                // wave owns: no barrier inside it, and none before it unless the previous exchange was a cross-wave
                // one whose readers may still be busy in this wave's region.
                const bool intra = (rh >> 17) & 1u;
                if (intra ? cross_pending : lds_dirty) {
                    lds_barrier();
                    cross_pending = false;
                }
                if constexpr (XMODE == 0) {
                    QSV_STAMP(2);
                    if (active) {
{
                            uint32_t o = wt;
#pragma unroll
                            for (int i = 0; i < NR; ++i) {
                                if (i) o ^= wrc[__builtin_ctz(i)] << LSH;
                                const int e = gray_index(i);
                                *reinterpret_cast<cxr*>(lds_raw + o) = amp[e];
                            }
                        }
                    }
                    if (!intra) lds_barrier();
                    {  // (every thread reads: see the gate loop below)
{
                            uint32_t o = rt;
#pragma unroll
                            for (int i = 0; i < NR; ++i) {
                                if (i) o ^= rrc[__builtin_ctz(i)] << LSH;
                                const int e = gray_index(i);
                                amp[e] = *reinterpret_cast<const cxr*>(lds_raw + o);
                            }
                        }
                    }
                } else if constexpr (XMODE == 1) {
                    QSV_STAMP(2);
                    unsigned char* pim = lds_raw + (size_t(sizeof(real)) << (hdr & 0xff));
                    if (active) {
{
                            uint32_t o = wt;
#pragma unroll
                            for (int i = 0; i < NR; ++i) {
                                if (i) o ^= wrc[__builtin_ctz(i)] << LSH;
                                const int e = gray_index(i);
                                *reinterpret_cast<real*>(lds_raw + o) = amp[e].re; *reinterpret_cast<real*>(pim + o) = amp[e].im;
                            }
                        }
                    }
                    if (!intra) lds_barrier();
                    {  // (every thread reads: see the gate loop below)
{
                            uint32_t o = rt;
#pragma unroll
                            for (int i = 0; i < NR; ++i) {
                                if (i) o ^= rrc[__builtin_ctz(i)] << LSH;
                                const int e = gray_index(i);
                                amp[e].re = *reinterpret_cast<const real*>(lds_raw + o); amp[e].im = *reinterpret_cast<const real*>(pim + o);
                            }
                        }
                    }
                } else {
                    QSV_STAMP(2);
                    if (active) {
{
                            uint32_t o = wt;
#pragma unroll
                            for (int i = 0; i < NR; ++i) {
                                if (i) o ^= wrc[__builtin_ctz(i)] << LSH;
                                const int e = gray_index(i);
                                *reinterpret_cast<real*>(lds_raw + o) = amp[e].re;
                            }
                        }
                    }
                    QSV_STAMP(3);
                    if (!intra) lds_barrier();
                    QSV_STAMP(4);
                    {  // (every thread reads: see the gate loop below)
{
                            uint32_t o = rt;
#pragma unroll
                            for (int i = 0; i < NR; ++i) {
                                if (i) o ^= rrc[__builtin_ctz(i)] << LSH;
                                const int e = gray_index(i);
                                amp[e].re = *reinterpret_cast<const real*>(lds_raw + o);
                            }
                        }
                    }
                    QSV_STAMP(5);
                    if (!intra) lds_barrier();
                    QSV_STAMP(6);
                    if (active) {
{
                            uint32_t o = wt;
#pragma unroll
                            for (int i = 0; i < NR; ++i) {
                                if (i) o ^= wrc[__builtin_ctz(i)] << LSH;
                                const int e = gray_index(i);
                                *reinterpret_cast<real*>(lds_raw + o) = amp[e].im;
                            }
                        }
                    }
                    QSV_STAMP(7);
                    if (!intra) lds_barrier();
                    QSV_STAMP(8);
                    {  // (every thread reads: see the gate loop below)
{
                            uint32_t o = rt;
#pragma unroll
                            for (int i = 0; i < NR; ++i) {
                                if (i) o ^= rrc[__builtin_ctz(i)] << LSH;
                                const int e = gray_index(i);
                                amp[e].im = *reinterpret_cast<const real*>(lds_raw + o);
                            }
                        }
                    }
                    QSV_STAMP(9);
                }
                lds_dirty = true;
                cross_pending = cross_pending || !intra;
            } else if ((rh >> 18) & 1u) {
                // relayout by lane swaps (plan.hpp): no LDS, no barrier
                QSV_STAMP(2);
                // (one dispatch site: the loop must not be unrolled, every case is a few dozen instructions)
#pragma nounroll
                for (int i = 0; i < int(kMaxSwaps); ++i) {
                    const uint32_t w = rp[i];
                    if (w == kSwapPad) break;
                    const uint32_t sel = (w & 0xffu) * 6u + ((w >> 8) & 0xffu);
#ifndef QSV_ABL_NOSWAP
                    if constexpr (std::is_same<real, double>::value && R <= 4)
                        SwapF64<R>::run(amp, sel, tid & 63u);  // generated assembly (gate_loop_gen.inc)
                    else
                        SwapDispatch<real, R, 6 * R - 1>::run(int(sel), amp);
#else
                    (void)sel;
#endif
                }
                rp += kMaxSwaps;
                QSV_STAMP(9);
            }
            if constexpr (std::is_same<real, double>::value && R <= 4) {
                // fp64: the gate loop is the generated assembly block (gate_loop_gen.inc); amplitudes never move
                if (n_gates > 0) {
#ifndef QSV_ABL_NOGATES  // (ablation builds, scripts/ablate.py: timing experiments with wrong results)
                    // (not under `if (active)`: threads beyond 2^t only exist in tiles smaller than a wave, their
                    // registers hold nothing anyone reads, and a conditional here makes hipcc carry the amplitudes
                    // through temporaries -- 16 v_mov_b64 into the block's fixed registers and 16 out, every round)
                    GateLoopF64<R>::run(amp, rp, mp, uint32_t(n_gates), uint32_t(base), tid_ext);
#endif
                    rp += size_t(n_gates) * kGateWords;
                    mp += size_t(n_gates) * 8;
                }
            } else {
                // gate stream: descriptor and matrix of gate g+1 are fetched (scalar loads) while gate g runs
                // (single precision: the records hold eight floats, prepare_eval's float_mats)
                auto entry = [](cf64p rec, int i) -> double {
                    if constexpr (std::is_same<real, float>::value)
                        return double(reinterpret_cast<const QSV_CONST_AS float*>(rec)[i]);
                    else
                        return rec[i];
                };
                uint32_t w0 = rp[0], ct = rp[1], cg = rp[2], ncg = rp[3];
                double m0 = entry(mp, 0), mi = entry(mp, 1), m1 = entry(mp, 2), m2 = entry(mp, 3), m3 = entry(mp, 4), m4 = entry(mp, 5),
                       m5 = entry(mp, 6), m6 = entry(mp, 7);
                for (int g = 0; g < n_gates; ++g) {
                    rp += kGateWords;
                    mp += 8;
                    const uint32_t nw0 = rp[0], nct = rp[1], nxcg = rp[2], nxncg = rp[3];
                    const double n0 = entry(mp, 0), ni = entry(mp, 1), n1 = entry(mp, 2), n2 = entry(mp, 3), n3 = entry(mp, 4),
                                 n4 = entry(mp, 5), n5 = entry(mp, 6), n6 = entry(mp, 7);
                    if ((w0 & (kGateGeneral | kGateNegated)) != 0) {
                        // an entry of a multiplexed gate or a product of matrices (plan.hpp FUSION): predicates over the
                        // complemented bits too, pairs by the entry's mask, general matrix
                        if ((uint32_t(base) & cg) == cg && (~uint32_t(base) & ncg) == ncg && active && (tid_ext & ct) == ct) {
                            const real mm[8] = {real(m0), real(mi), real(m1), real(m2), real(m3), real(m4), real(m5), real(m6)};
                            GeneralDispatch<real, R, R - 1>::run(int(w0 & 0xffu), amp, mm, (w0 >> 16) & 0xffu);
                        }
                    } else
                    if ((uint32_t(base) & cg) == cg) {  // else: the control is a fixed bit of this tile and it is 0
                        const uint32_t creg = (w0 >> 8) & 0xffu;
                        const int sel = int(w0 & 0xffu) * (R + 1) + (creg == 0xffu ? 0 : int(creg) + 1);
                        const real mm[7] = {real(m0), real(m1), real(m2), real(m3), real(m4), real(m5), real(m6)};
                        if (all_active && (ct & 63u) == 0) {
                            // the control (if any) is a wave-index bit: whole waves either run the gate or skip it
                            if ((wave_base & ct) == ct) ButterflyDispatch<real, R, R*(R + 1) - 1>::run(sel, amp, mm);
                        } else if (active && ((tid & ct) == ct)) {
                            // per-lane control: lanes whose control bit is 0 sit the gate out under the exec mask
                            ButterflyDispatch<real, R, R*(R + 1) - 1>::run(sel, amp, mm);
                        }
                    }
                    w0 = nw0; ct = nct; cg = nxcg; ncg = nxncg;
                    m0 = n0; mi = ni; m1 = n1; m2 = n2; m3 = n3; m4 = n4; m5 = n5; m6 = n6;
                }
            }
            QSV_STAMP(10);
        }

        // (rows in LDS from offset 0: over whatever the last exchange left there for a slower wave to read)
        QSV_TL(2);
        if (table_lds_rows) __syncthreads();
        if (active && (do_store || do_diag || do_probs)) {
            if (!wide || cstore) {
                unsigned char* tile = reinterpret_cast<unsigned char*>(cstore ? wt0 + (uint64_t(tile0 + j * tile_step) << k) : st0 + base);
                const unsigned char* dtile = reinterpret_cast<const unsigned char*>(diag + base);
                if (do_probs) {
                    // the sampler only wants the probabilities: written here, the state is never stored or read again
                    unsigned char* ptile = reinterpret_cast<unsigned char*>(partials + uint64_t(ev.state_slot) * a.state_stride + base);
                    uint32_t ob = sgv << 3;
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        if (i) ob ^= gsr[__builtin_ctz(i)] << 3;
                        const double re = double(amp[gray_index(i)].re), im = double(amp[gray_index(i)].im);
                        *reinterpret_cast<double*>(ptile + ob) = re * re + im * im;
                    }
                }
                if (do_store) {
                    uint32_t ob = sgv << ASH;
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        if (i) ob ^= gsr[__builtin_ctz(i)] << ASH;
                        if (streaming) {
                            typedef real vec2 __attribute__((ext_vector_type(2)));
                            const vec2 v = {amp[gray_index(i)].re, amp[gray_index(i)].im};
                            __builtin_nontemporal_store(v, reinterpret_cast<vec2*>(tile + ob));
                        } else if (table_in_lds) {
                            *reinterpret_cast<cxr*>(lds_raw + kFusedLdsTableOffset + ob) = amp[gray_index(i)];
                            // (a half side: the other half of x of this row goes to the side's slot as well, for the partner)
                            if (half_side && ((ob >> (uint32_t(kFusedLdsRowsBits) - 1u - plan_arena[ev.split_base] + ASH)) & 1u) != tile0) store_through(tile + ob, amp[gray_index(i)]);
                        } else if (table_lds_rows) {
                            // (one amplitude of padding after every row of 2^10)
                            constexpr uint32_t row_shift = uint32_t(kFusedLdsRowsBits - kFusedLdsRowsKeys) + ASH;
                            *reinterpret_cast<cxr*>(lds_raw + ob + ((ob >> row_shift) << ASH)) = amp[gray_index(i)];
                            // (a half side: what the partner will want of this row -- the other half of x -- goes to the side's slot
                            // as well, where it would lie there; fused_factor_tail drains these stores)
                            if (half_side && ((ob >> (row_shift - 1)) & 1u) != tile0) store_through(tile + ob, amp[gray_index(i)]);
                        } else if (through) {
                            store_through(tile + ob, amp[gray_index(i)]);
                        } else {
                            *reinterpret_cast<cxr*>(tile + ob) = amp[gray_index(i)];
                        }
                    }
                }
                if (do_diag) {
                    // all D[i] loads first, so that they are in flight together (one latency, not 2^R in a row)
                    double dv[NR];
                    uint32_t ob = sgv << 3;
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        if (i) ob ^= gsr[__builtin_ctz(i)] << 3;
#ifndef QSV_ABL_NODIAG
                        dv[gray_index(i)] = *reinterpret_cast<const double*>(dtile + ob);
#else
                        dv[gray_index(i)] = double(ob);
#endif
                    }
#pragma unroll
                    for (int e = 0; e < NR; ++e) {
                        const double re = double(amp[e].re), im = double(amp[e].im);
                        acc += (re * re + im * im) * dv[e];
                    }
                }
            } else {
                uint32_t off = sgv;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    off = gray_step(i, off, gsr);
                    const cxr x = amp[gray_index(i)];
                    if (do_store) st0[base + off] = x;
                    if (do_diag) {
                        const double re = double(x.re), im = double(x.im);
                        acc += (re * re + im * im) * diag[base + off];
                    }
                }
            }
        }
        QSV_STAMP(11);
        QSV_TL(3);
    }
#ifdef QSV_TIMELINE
    if (tl_rec && threadIdx.x == 0) {
        tl_rec[4] = qsv_realtime_now();
        tl_rec[5] = qsv_time_now();
    }
#endif

    if constexpr (FUSED) {
        const bool halves_side = half_side && (table_lds_rows || table_in_lds);  // (its two workgroups: tile0 = 0, 1; R = 3: t = 9, eight waves)
        if (half_side && !halves_side) {
            // (a launch without the LDS for it: the host never makes one -- no value rather than a wrong one)
            if (threadIdx.x == 0) a.result_out[ev.out_index] = __builtin_nan("");
            return;
        }
        // split evaluations whose virtual circuits are one tile and one pass each, under a quadratic operator: this side's
        // workgroup goes straight on to its weighted Gram matrices, and the side that finishes second combines
        if (side && (ev.flags & kEvalFused) && (a.mode & kModeFusedFactor)) {
#ifdef QSV_STAMPS  // (diagnostic build: the virtual circuit's phases in row 0, the tail's in row 7: 0 drain, 1 Gram, 2 hand-off, 3 combine)
            QSV_STAMP_FLUSH(0u);
            for (int ph = 0; ph < kStampPhases; ++ph) st_acc[ph] = 0;
            st_last = qsv_stamp_now();
            fused_factor_tail<real>(plan_arena, ev, wt0, a.wtab_stride, diag, a, lds_raw, t >= 9 ? 8u : 4u, table_in_lds ? 1u : table_lds_rows ? 2u : 0u, halves_side ? tile0 : 0xffffffffu, st_acc, &st_last);
            QSV_STAMP_FLUSH(7u);
#else
            // (Gram waves: by the virtual circuit's own geometry, never by the launch's block size)
            fused_factor_tail<real>(plan_arena, ev, wt0, a.wtab_stride, diag, a, lds_raw, t >= 9 ? 8u : 4u, table_in_lds ? 1u : table_lds_rows ? 2u : 0u, halves_side ? tile0 : 0xffffffffu);
#endif
            return;
        }
    }
    if (do_diag) {
        // one partial sum per WAVE leaves the kernel (fixed shuffle tree, no LDS, no barrier); reduce_partials_kernel
        // adds them in a fixed order
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        const uint32_t n_waves = blockDim.x >> 6;
        if (a.mode & kModeDirectResult) {
            // the evaluation is this workgroup alone: its waves' sums are added here, in wave order, and the result goes
            // straight to the caller's (pinned) buffer -- no partial sums, no reduction launch
            double* sums = reinterpret_cast<double*>(lds_raw);
            __syncthreads();  // (whoever still reads the last exchange's data from LDS is done after this)
            if ((tid & 63u) == 0) sums[wave] = acc;
            __syncthreads();
            if (tid == 0) {
                double total = 0.0;
                for (uint32_t w = 0; w < n_waves; ++w) total += sums[w];
                a.result_out[ev.out_index] = total;
            }
            return;
        }
        // a launch with fewer workgroups than the reducer's shape also clears the slots nobody owns
        const uint32_t slots = a.partial_chunks ? a.partial_chunks : grid_x;
        if ((tid & 63u) == 0) {
            double* mine = partials + size_t(ev.out_index) * slots * n_waves + wave;
            mine[size_t(block_x) * n_waves] = acc;
            for (uint32_t b2 = block_x + grid_x; b2 < slots; b2 += grid_x) mine[size_t(b2) * n_waves] = 0.0;
        }
    }
#ifdef QSV_STAMPS
    QSV_STAMP(12);
    QSV_STAMP_FLUSH(a.pass_index);
#endif
}

template <typename real, int R, int XMODE>
static hipError_t launch_pass_t(dim3 grid, int threads, size_t lds_bytes, hipStream_t stream, const PassArgs& args) {
    // the block reduction at the end needs one double per wave (and the diagnostic build a table of counters)
    const size_t lds = lds_bytes < 256 ? 256 : lds_bytes;
    const PassScalars sc{args.state_stride, args.wtab_stride, args.pass_index, args.mode, args.tiles_per_block,
                         args.partial_chunks, args.region_stride, args.host_evals, args.evals_out, args.host_params,
                         args.mats_out, args.result_out, args.quad, args.factor_scratch, args.factor_counters, args.n_full,
                         args.prefix_states, args.dephase, args.side_diag};
    cx<real>* st = reinterpret_cast<cx<real>*>(args.states);
    const bool first = args.pass_index == 0 && (args.mode & kModeSynthFirst);
    if constexpr (R == 4 || R == 3) {
        if (first && (args.mode & kModeFusedFactor)) {
            hipLaunchKernelGGL((pass_kernel<real, R, XMODE, true, true>), grid, dim3(threads), lds, stream, args.plan, args.mats,
                               args.evals, st, reinterpret_cast<cx<real>*>(args.wtab), args.diag, args.partials, sc);
            return hipGetLastError();
        }
    }
    if (first)
        hipLaunchKernelGGL((pass_kernel<real, R, XMODE, true>), grid, dim3(threads), lds, stream, args.plan, args.mats,
                           args.evals, st, reinterpret_cast<cx<real>*>(args.wtab), args.diag, args.partials, sc);
    else
        hipLaunchKernelGGL((pass_kernel<real, R, XMODE, false>), grid, dim3(threads), lds, stream, args.plan, args.mats,
                           args.evals, st, reinterpret_cast<cx<real>*>(args.wtab), args.diag, args.partials, sc);
    return hipGetLastError();
}

template <typename real, int R, int XMODE>
static hipError_t configure_t(size_t lds_bytes) {
    const int bytes = int(lds_bytes < 256 ? 256 : lds_bytes);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_kernel<real, R, XMODE, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    if constexpr (R == 4 || R == 3) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_kernel<real, R, XMODE, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_kernel<real, R, XMODE, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// op = 0: launch, op = 1: configure
template <typename real, int XMODE>
static hipError_t pass_dispatch_r(int op, int r, dim3 grid, int threads, size_t lds_bytes, hipStream_t stream,
                                  const PassArgs* args) {
    switch (r) {
        case 1: return op ? configure_t<real, 1, XMODE>(lds_bytes) : launch_pass_t<real, 1, XMODE>(grid, threads, lds_bytes, stream, *args);
        case 2: return op ? configure_t<real, 2, XMODE>(lds_bytes) : launch_pass_t<real, 2, XMODE>(grid, threads, lds_bytes, stream, *args);
        case 3: return op ? configure_t<real, 3, XMODE>(lds_bytes) : launch_pass_t<real, 3, XMODE>(grid, threads, lds_bytes, stream, *args);
        case 4: return op ? configure_t<real, 4, XMODE>(lds_bytes) : launch_pass_t<real, 4, XMODE>(grid, threads, lds_bytes, stream, *args);
        default: return hipErrorInvalidValue;
    }
}

static hipError_t pass_dispatch(int op, int dtype, int r, int xmode, dim3 grid, int threads, size_t lds_bytes,
                                hipStream_t stream, const PassArgs* args) {
    if (threads > 512) return hipErrorInvalidValue;
#ifdef QSV_PROBE_ONLY  // scripts/isa_probe.sh: compile just one instantiation (-DQSV_PROBE_R=3|4) to read its ISA quickly
#ifndef QSV_PROBE_R
#define QSV_PROBE_R 4
#endif
    if (op) return hipSuccess;
    const PassScalars sc{args->state_stride, args->wtab_stride, args->pass_index, args->mode, args->tiles_per_block,
                         args->partial_chunks, args->region_stride, args->host_evals, args->evals_out, args->host_params,
                         args->mats_out, args->result_out, args->quad, args->factor_scratch, args->factor_counters, args->n_full,
                          args->prefix_states, args->dephase, args->side_diag};
    if (args->pass_index == 0 && (args->mode & kModeSynthFirst))
        hipLaunchKernelGGL((pass_kernel<double, QSV_PROBE_R, 2, true>), grid, dim3(threads), lds_bytes, stream, args->plan,
                           args->mats, args->evals, reinterpret_cast<cx<double>*>(args->states),
                           reinterpret_cast<cx<double>*>(args->wtab), args->diag, args->partials, sc);
    else
    hipLaunchKernelGGL((pass_kernel<double, QSV_PROBE_R, 2, false>), grid, dim3(threads), lds_bytes, stream, args->plan,
                       args->mats, args->evals, reinterpret_cast<cx<double>*>(args->states),
                       reinterpret_cast<cx<double>*>(args->wtab), args->diag, args->partials, sc);
    return hipGetLastError();
#else
    if (dtype == 0) {
        switch (xmode) {
            case 0: return pass_dispatch_r<double, 0>(op, r, grid, threads, lds_bytes, stream, args);
            case 1: return pass_dispatch_r<double, 1>(op, r, grid, threads, lds_bytes, stream, args);
            case 2: return pass_dispatch_r<double, 2>(op, r, grid, threads, lds_bytes, stream, args);
            default: return hipErrorInvalidValue;
        }
    }
    switch (xmode) {
        case 0: return pass_dispatch_r<float, 0>(op, r, grid, threads, lds_bytes, stream, args);
        case 2: return pass_dispatch_r<float, 2>(op, r, grid, threads, lds_bytes, stream, args);
        default: return hipErrorInvalidValue;
    }
#endif
}

hipError_t launch_pass(int dtype, int r, int xmode, dim3 grid, int threads, size_t lds_bytes, hipStream_t stream,
                       const PassArgs& args) {
    return pass_dispatch(0, dtype, r, xmode, grid, threads, lds_bytes, stream, &args);
}

hipError_t configure_pass_kernels(int dtype, int r, int xmode, size_t lds_bytes) {
    return pass_dispatch(1, dtype, r, xmode, dim3(1), 64, lds_bytes, nullptr, nullptr);
}

// `host_evals` and `params` point into PINNED HOST memory: the kernel fetches the few hundred bytes an evaluation
// needs over PCIe itself and leaves a device copy of the descriptor for the pass kernels.  Separate H2D copies in
// front of it cost two more dependent stream operations (~50 us before the first pass of a step could start).
__global__ void __launch_bounds__(256) prepare_kernel(const uint32_t* __restrict__ plan,
                                                      const EvalDesc* __restrict__ host_evals,
                                                      EvalDesc* __restrict__ evals,
                                                      const double* __restrict__ params, double* __restrict__ mats,
                                                      uint32_t region_stride, uint32_t float_mats) {
    __shared__ double scratch[kPrepScratchDoubles];
    const size_t slot = size_t(blockIdx.x) + size_t(blockIdx.y) * region_stride;
    const EvalDesc ev = host_evals[slot];
    if (threadIdx.x == 0) evals[slot] = ev;
    if (ev.flags & kEvalNull) return;
    prepare_eval(plan, ev, params, mats, scratch, float_mats != 0);
}

hipError_t launch_prepare(const uint32_t* plan, const EvalDesc* host_evals, EvalDesc* evals, const double* params,
                          double* mats, int n_evals, hipStream_t stream, int n_regions, uint32_t region_stride, int dtype) {
    hipLaunchKernelGGL(prepare_kernel, dim3(n_evals, n_regions), dim3(256), 0, stream, plan, host_evals, evals, params,
                       mats, region_stride, uint32_t(dtype != 0));
    return hipGetLastError();
}

// ---- split evaluations: contraction of the two side tables with the diagonal operator ---------------------------
// psi[i] = sum_j X_j[x(i)] * Y_j[y(i)]: a rank-J outer product.  A thread owns 32 amplitudes (five index bits, the same
// positions for every circuit): LX of them belong to side X, the others to side Y, so it loads 2^LX * J values of X,
// 2^(5-LX) * J of Y and 32 of D -- independent loads, a few memory latencies per thread instead of one per amplitude --
// and spends 8 J + 5 flops per amplitude.  Lanes are index bits 0 .. 5 (D is read in 512-byte runs), the workgroup
// number the top bits; workgroup w of the 1-D grid takes part w mod 8 of D for one evaluation after the other, so every
// part of D is read by ONE XCD, once from memory and then from its L2 (D used to be re-read from memory for every
// evaluation).  Everything circuit-dependent comes as ready-made pieces of the two table indices (split block,
// kernels.hpp): one level of loads behind the descriptor.  Partial sums as in the pass kernel's fused last pass.
template <typename real, int J, int LX, int LY, int YB>
__device__ __forceinline__ double contract_block(const unsigned char* __restrict__ bx, const unsigned char* __restrict__ by,
                                                 const unsigned char* __restrict__ bd, uint32_t ix, uint32_t iy, uint32_t i0,
                                                 uint32_t bits_x, uint32_t bits_y, const uint32_t* col, const uint32_t* pos) {
    // every offset is a 32-bit BYTE offset from a uniform base (tables and D are far below 4 GiB: n <= 28): one VGPR per
    // address and scalar-base loads; with 64-bit addresses the kernel needed 204 VGPRs and ran two waves per SIMD
    constexpr int NX = 1 << LX, NY = 1 << LY;
    constexpr int ASH = Log2Size<real>::value + 1;
    double xr[NX][J], xi[NX][J];
#pragma unroll
    for (int a = 0; a < NX; ++a) {
        uint32_t idx = ix;
#pragma unroll
        for (int b = 0; b < LX; ++b)
            if (a >> b & 1) idx ^= col[b];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const cx<real> v = *reinterpret_cast<const cx<real>*>(bx + (((uint32_t(j) << bits_x) + idx) << ASH));
            xr[a][j] = double(v.re);
            xi[a][j] = double(v.im);
        }
    }
    double acc = 0.0;
    // one chunk of YB values of y per trip, NOT unrolled: hoisted to the top, the loads of every chunk (32 + 12 J per
    // thread) cost 200+ VGPRs -- two waves per SIMD, or spills under a register cap
#pragma nounroll
    for (int y0 = 0; y0 < NY; y0 += YB) {
        double yr[YB][J], yi[YB][J], d[YB][NX];
#pragma unroll
        for (int c = 0; c < YB; ++c) {
            uint32_t sy = 0, si = 0;  // (uniform: the chunk's part of the offsets)
#pragma unroll
            for (int b = 0; b < LY; ++b)
                if ((y0 + c) >> b & 1) {
                    sy ^= col[LX + b];
                    si ^= pos[LX + b];
                }
            const uint32_t idx = iy ^ sy, i = i0 ^ si;
#pragma unroll
            for (int j = 0; j < J; ++j) {
#ifndef QSV_ABL_CT_NOTAB
                const cx<real> v = *reinterpret_cast<const cx<real>*>(by + (((uint32_t(j) << bits_y) + idx) << ASH));
                yr[c][j] = double(v.re);
                yi[c][j] = double(v.im);
#else
                yr[c][j] = double(idx + uint32_t(j));
                yi[c][j] = double(idx);
#endif
            }
#pragma unroll
            for (int a = 0; a < NX; ++a) {
                uint32_t ia = i;
#pragma unroll
                for (int b = 0; b < LX; ++b)
                    if (a >> b & 1) ia ^= pos[b];
#ifndef QSV_ABL_CT_NOD  // (ablation builds, scripts/ablate.py: timing experiments with wrong results)
                d[c][a] = *reinterpret_cast<const double*>(bd + (ia << 3));
#else
                d[c][a] = double(ia);
#endif
            }
        }
#pragma unroll
        for (int c = 0; c < YB; ++c)
#pragma unroll
            for (int a = 0; a < NX; ++a) {
                double pr = xr[a][0] * yr[c][0], pi = xr[a][0] * yi[c][0];
                pr = fma(-xi[a][0], yi[c][0], pr);
                pi = fma(xi[a][0], yr[c][0], pi);
#pragma unroll
                for (int j = 1; j < J; ++j) {
                    pr = fma(xr[a][j], yr[c][j], fma(-xi[a][j], yi[c][j], pr));
                    pi = fma(xr[a][j], yi[c][j], fma(xi[a][j], yr[c][j], pi));
                }
#ifndef QSV_ABL_CT_NOMATH
                acc = fma(fma(pr, pr, pi * pi), d[c][a], acc);
#else
                acc += d[c][a] + yr[c][0];
#endif
            }
    }
    return acc;
}

template <typename real, int J>
__device__ __forceinline__ double contract_by_shape(uint32_t lx, const unsigned char* bx, const unsigned char* by,
                                                    const unsigned char* bd, uint32_t ix, uint32_t iy, uint32_t i0,
                                                    uint32_t bits_x, uint32_t bits_y, const uint32_t (&col)[kSplitLoopBits],
                                                    const uint32_t (&pos)[kSplitLoopBits]) {
    if (lx == 2) {
        if constexpr (J <= 4) {
            return contract_block<real, J, 2, 3, (J == 1 ? 8 : 4 / J)>(bx, by, bd, ix, iy, i0, bits_x, bits_y, col, pos);
        } else {
            // eight terms: the values of X for two bits alone would fill the registers; the second X bit is walked outside
            double acc = 0.0;
#pragma nounroll
            for (int o = 0; o < 2; ++o)
                acc += contract_block<real, J, 1, 3, 1>(bx, by, bd, o ? ix ^ col[0] : ix, iy, o ? i0 ^ pos[0] : i0, bits_x, bits_y,
                                                        col + 1, pos + 1);
            return acc;
        }
    }
    if (lx == 1) return contract_block<real, J, 1, 4, (J == 1 ? 16 : 8 / J)>(bx, by, bd, ix, iy, i0, bits_x, bits_y, col, pos);
    return contract_block<real, J, 0, 5, (J == 1 ? 16 : 8 / J)>(bx, by, bd, ix, iy, i0, bits_x, bits_y, col, pos);
}

// (Measured: one launch per number of terms instead of the switch below costs more than it saves -- the classes with
// several keys hold a few evaluations each and a small launch still takes ten microseconds.)
template <typename real>
__global__ void __launch_bounds__(512, 4) contract_kernel(const uint32_t* __restrict__ plan_arena,
                                                          const EvalDesc* __restrict__ evals,
                                                          const cx<real>* __restrict__ wtabs, const double* __restrict__ diag,
                                                          double* __restrict__ partials, uint64_t wtab_stride,
                                                          uint32_t n_qubits, uint32_t partial_chunks, uint32_t n_chunks,
                                                          uint32_t n_evals, uint32_t evals_per_group) {
    // workgroup -> (group of evaluations, chunk of the index space): consecutive workgroups go to consecutive XCDs, so
    // chunk c is handled by XCD c mod 8 for every evaluation (when there are at least 8 chunks).  A workgroup takes
    // evals_per_group evaluations one after the other (1 unless QSV_CONTRACT_GROUP says otherwise, see launch_contract).
    const uint32_t n_groups = (n_evals + evals_per_group - 1) / evals_per_group;
    uint32_t chunk, group;
    if ((n_chunks & 7u) == 0) {
        const uint32_t xcd = blockIdx.x & 7u, rest = blockIdx.x >> 3;
        group = rest % n_groups;
        chunk = (rest / n_groups) * 8u + xcd;
    } else {
        group = blockIdx.x % n_groups;
        chunk = blockIdx.x / n_groups;
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wave_bits = 25u - uint32_t(__builtin_clz(blockDim.x));  // log2(blockDim / 64)
    const uint32_t i0 = tid | chunk << (6 + wave_bits + kSplitLoopBits);  // (the thread's own bits start at 0)
    const unsigned char* bd = reinterpret_cast<const unsigned char*>(diag);
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t slots = partial_chunks ? partial_chunks : n_chunks;
    // (a group's evaluations are taken with stride n_groups: the descriptors are sorted by number of keys, and
    // neighbours in that order cost the same -- a group of the expensive ones would be the launch's tail)
#pragma nounroll
    for (uint32_t which = group; which < n_evals; which += n_groups) {
        EvalDesc ev;
        {
            cu32p e = as_constant(reinterpret_cast<const uint32_t*>(evals + which));
            ev.state_slot = e[2];
            ev.out_index = e[3];
            ev.flags = e[6];
            ev.split_base = e[7];
        }
        if (!(ev.flags & kEvalSide)) continue;
        cu32p sp = as_constant(plan_arena) + ev.split_base;
        // everything circuit-dependent: the header, and this thread's pieces of the two table indices
        uint32_t hdr[16];
        load_words<16>(sp, hdr);
        const uint32_t* lane_entry = plan_arena + ev.split_base + kSplitLaneTable + 2 * (tid & 63u);
        const uint32_t lane_x = lane_entry[0], lane_y = lane_entry[1];
        cu32p we = sp + kSplitWaveTable + 2 * wave, c0 = sp + kSplitChunkLow + 2 * (chunk & 127u),
              c1 = sp + kSplitChunkHigh + 2 * (chunk >> 7);
        const uint32_t ix = lane_x | we[0] | c0[0] | c1[0], iy = lane_y | we[1] | c0[1] | c1[1];
        const uint32_t n_keys = hdr[0], bits_x = hdr[1], bits_y = hdr[2];
        const bool swap = hdr[3] & 1u;
        const uint32_t lx = hdr[3] >> 8;
        uint32_t col[kSplitLoopBits], pos[kSplitLoopBits];
#pragma unroll
        for (int b = 0; b < kSplitLoopBits; ++b) {
            col[b] = hdr[kSplitLoopCols + b];
            pos[b] = hdr[kSplitLoopPos + b];
        }
        const cx<real>* __restrict__ ta = wtabs + uint64_t(ev.state_slot) * wtab_stride;
        const unsigned char* bx = reinterpret_cast<const unsigned char*>(ta + (swap ? wtab_stride >> 1 : 0));
        const unsigned char* by = reinterpret_cast<const unsigned char*>(ta + (swap ? 0 : wtab_stride >> 1));
        // chunk sizes by number of terms: one term keeps every load of a thread in flight at once, eight walk the tables
        // in small steps; all variants fit the same 128 registers
        double acc;
        if (n_keys == 0)
            acc = contract_by_shape<real, 1>(lx, bx, by, bd, ix, iy, i0, bits_x, bits_y, col, pos);
        else if (n_keys == 1)
            acc = contract_by_shape<real, 2>(lx, bx, by, bd, ix, iy, i0, bits_x, bits_y, col, pos);
        else if (n_keys == 2)
            acc = contract_by_shape<real, 4>(lx, bx, by, bd, ix, iy, i0, bits_x, bits_y, col, pos);
        else
            acc = contract_by_shape<real, 8>(lx, bx, by, bd, ix, iy, i0, bits_x, bits_y, col, pos);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if ((tid & 63u) == 0) {
            double* mine = partials + size_t(ev.out_index) * slots * n_waves + wave;
            mine[size_t(chunk) * n_waves] = acc;
            for (uint32_t b2 = chunk + n_chunks; b2 < slots; b2 += n_chunks) mine[size_t(b2) * n_waves] = 0.0;
        }
    }
}

hipError_t launch_contract(int dtype, unsigned n_chunks, unsigned n_evals, int threads, hipStream_t stream, const PassArgs& a) {
    const uint32_t n = uint32_t(63 - __builtin_clzll(a.state_stride));
    if (threads < 64 || threads > 512 || (threads & (threads - 1))) return hipErrorInvalidValue;
    // every index once: n_chunks workgroups x threads x the 32 amplitudes of a thread's block
    if (uint64_t(n_chunks) * uint64_t(threads) << kSplitLoopBits != uint64_t(1) << n || n > 28) return hipErrorInvalidValue;
    // One evaluation per workgroup.  (QSV_CONTRACT_GROUP = e lets a workgroup take e evaluations in turn -- fewer
    // workgroups to dispatch; measured slower at n = 20: 44 us per launch of 32 evaluations for e = 1, 47 / 55 / 49 / 60
    // for e = 2 / 4 / 8 / 16: the evaluations of a workgroup run one after the other, each with its own chain of loads.)
    static const unsigned env_group = getenv("QSV_CONTRACT_GROUP") ? unsigned(atoi(getenv("QSV_CONTRACT_GROUP"))) : 0u;
    const unsigned per_group = std::min(std::max(1u, env_group), std::max(1u, n_evals));
    const unsigned n_groups = (n_evals + per_group - 1) / per_group;
    const dim3 grid(n_chunks * n_groups);
    if (dtype == 0)
        hipLaunchKernelGGL(contract_kernel<double>, grid, dim3(threads), 0, stream, a.plan, a.evals,
                           static_cast<const cx<double>*>(a.wtab), a.diag, a.partials, a.wtab_stride, n, a.partial_chunks,
                           n_chunks, n_evals, per_group);
    else
        hipLaunchKernelGGL(contract_kernel<float>, grid, dim3(threads), 0, stream, a.plan, a.evals,
                           static_cast<const cx<float>*>(a.wtab), a.diag, a.partials, a.wtab_stride, n, a.partial_chunks,
                           n_chunks, n_evals, per_group);
    return hipGetLastError();
}

// ---- diagonal table ------------------------------------------------------------------------------------
// D[i] = sum_k c_k (-1)^popcount(i & z_k), terms added in index order (same order as the oracle).
__global__ void __launch_bounds__(256) diag_table_kernel(uint64_t dim, int n_terms, const uint64_t* __restrict__ z,
                                                         const double* __restrict__ c, double* __restrict__ table) {
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < dim; i += stride) {
        double d = 0.0;
        for (int k = 0; k < n_terms; ++k) {
            const double ck = c[k];
            d += (__popcll(i & z[k]) & 1) ? -ck : ck;
        }
        table[i] = d;
    }
}

hipError_t launch_diag_table(int n_qubits, int n_terms, const uint64_t* z_mask, const double* coeff, double* table,
                             hipStream_t stream) {
    const uint64_t dim = uint64_t(1) << n_qubits;
    const uint64_t want = (dim + 255) / 256;
    const unsigned blocks = unsigned(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(diag_table_kernel, dim3(blocks), dim3(256), 0, stream, dim, n_terms, z_mask, coeff, table);
    return hipGetLastError();
}

// ---- reductions ----------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) reduce_partials_kernel(const double* __restrict__ partials, uint32_t blocks,
                                                              double* __restrict__ out, const EvalDesc* __restrict__ evals) {
    __shared__ double red[4];
    const uint32_t e = evals ? evals[blockIdx.x].out_index : blockIdx.x;
    const double* p = partials + size_t(e) * blocks;
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < blocks; i += 256) acc += p[i];
    const double total = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[e] = total;
}

// ---- qsv_spsa_step ----------------------------------------------------------------------------------------------------------
// The arithmetic of _SPSARun.accept / propose (queasars_amd/evqe/solver.py; constant-gain SPSA as the reference's notebook
// configures qiskit_algorithms' optimiser) for one run per workgroup, every product and sum rounded on its own (contraction off:
// the host computes x + eps * delta in two roundings), and the reference's termination rule (queasars/utility/
// spsa_termination.py:46-94 with accepted = True) in thread 0.  The norm of an update is a fixed-order sum over the run's row
// (strided partial sums, then a tree): deterministic, not the host's order.
__global__ void __launch_bounds__(256) spsa_step_kernel(const SpsaStepArgs a) {
#pragma clang fp contract(off)  // (every product and sum below is rounded on its own, as NumPy's are: x - u * lr must not become one fma)
    __shared__ double red[256];
    __shared__ double s_scale;
    const int r = blockIdx.x, tid = threadIdx.x;
    double* x = a.x + size_t(r) * size_t(a.width);
    if (a.values) {
        const double* delta = a.delta_accept + size_t(r) * size_t(a.width);
        const double f_plus = a.values[2 * r], f_minus = a.values[2 * r + 1];
        const double g = (f_plus - f_minus) / (2.0 * a.eps);
        const bool was_active = a.active[r] != 0;
        __syncthreads();  // (every wave has read the flag before thread 0 may store the new one further down)
        double scale = 1.0;
        if (a.trust_region) {
            double acc = 0.0;
            for (int j = tid; j < a.width; j += 256) {
                const double u = g * delta[j];
                acc = acc + u * u;
            }
            red[tid] = acc;
            __syncthreads();
            for (int half = 128; half > 0; half >>= 1) {
                if (tid < half) red[tid] = red[tid] + red[tid + half];
                __syncthreads();
            }
            if (tid == 0) {
                const double norm = sqrt(red[0]);
                s_scale = norm > 1.0 ? norm : 1.0;
            }
            __syncthreads();
            scale = s_scale;
        }
        if (was_active)
            for (int j = tid; j < a.width; j += 256) {
                double u = g * delta[j];
                if (a.trust_region) u = u / scale;
                u = u * a.lr;
                x[j] = x[j] - u;
            }
        if (tid == 0) {
            const long long it = a.iterations[r] + (was_active ? 1 : 0);
            bool stop = it >= a.maxiter;
            if (a.window > 0) {
                const bool over = a.maxfev >= 0 && 2 * it >= a.maxfev;
                stop = stop || over;
                const bool fed = was_active && !over;  // (the reference returns before it stores anything)
                if (fed) {
                    const double value = 0.5 * (f_plus + f_minus);
                    double* ch = a.changes + size_t(r) * size_t(a.window);
                    if (a.n_values[r] >= 1) {
                        const double prev = a.previous[r];
                        const double change = fabs(value - prev) / prev;
                        double most = change;
                        for (int w = 0; w + 1 < a.window; ++w) {
                            ch[w] = ch[w + 1];
                            most = ch[w] > most ? ch[w] : most;  // (+inf where the window is not full yet)
                        }
                        ch[a.window - 1] = change;
                        if (most < a.min_rel) stop = true;
                    }
                    a.previous[r] = value;
                    a.n_values[r] += 1;
                }
            }
            a.iterations[r] = it;
            a.active[r] = was_active && !stop;
        }
        __syncthreads();  // (x is complete before the proposal below reads it)
    }
    if (a.delta_propose) {
        const double* delta = a.delta_propose + size_t(r) * size_t(a.width);
        double* p_plus = a.points + size_t(2 * r) * size_t(a.width);
        double* p_minus = p_plus + a.width;
        for (int j = tid; j < a.width; j += 256) {
            const double shift = delta[j] * a.eps;
            p_plus[j] = x[j] + shift;
            p_minus[j] = x[j] - shift;
        }
    }
}

hipError_t launch_spsa_step(const SpsaStepArgs& args, hipStream_t stream) {
    if (args.n_runs <= 0 || args.width <= 0) return hipSuccess;
    hipLaunchKernelGGL(spsa_step_kernel, dim3(unsigned(args.n_runs)), dim3(256), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_reduce_partials(const double* partials, uint32_t blocks, int n_evals, double* out,
                                  hipStream_t stream, const EvalDesc* evals) {
    if (n_evals <= 0) return hipSuccess;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(n_evals), dim3(256), 0, stream, partials, blocks, out, evals);
    return hipGetLastError();
}

// ---- general Pauli terms -------------------------------------------------------------------------------
// Terms are grouped by x mask (host).  For a group with mask x != 0 and pivot bit b = highest set bit of x, every
// unordered pair (i, j = i ^ x) with bit b of i clear is visited once -- so each amplitude is read once per group --
// and contributes  2 * sgn_k(i) * coef_k * (Re or Im of conj(a_i) a_j)  for every term k of the group:
//   <P> = sum_pairs i^{ny} (-1)^{popcount(i & z)} [ (-1)^{ny} c + conj(c) ],  c = conj(a_i) a_j
//       = +-2 Re c (ny even) or +-2 Im c (ny odd); the sign (-1)^{floor(ny/2)} is folded into coef_k on the host.
// The x = 0 group (diagonal terms) does not come here: it uses the diagonal table inside the last gate pass.
template <typename real>
__global__ void __launch_bounds__(256) pauli_groups_kernel(const cx<real>* __restrict__ states, uint64_t state_stride,
                                                           uint64_t n_pairs, const PauliGroup* __restrict__ groups,
                                                           const uint64_t* __restrict__ term_z,
                                                           const double* __restrict__ term_coef,
                                                           const uint32_t* __restrict__ term_odd,
                                                           double* __restrict__ partials) {
    // The group's terms are staged in LDS once per workgroup, kChunk at a time (z mask, and the coefficient filed under
    // "weight of Re" or "weight of Im"): the pair loop then reads them by broadcast instead of going back to global
    // memory for every amplitude pair.
    constexpr uint32_t kChunk = 256;
    __shared__ double red[4];
    __shared__ uint64_t sz[kChunk];
    __shared__ double s_re[kChunk], s_im[kChunk];
    const PauliGroup g = groups[blockIdx.y];
    const int slot = blockIdx.z;
    const cx<real>* __restrict__ st = states + uint64_t(slot) * state_stride;
    const uint64_t low_mask = (uint64_t(1) << g.pivot) - 1;
    double acc = 0.0;
    const uint64_t stride = uint64_t(gridDim.x) * 256;
    for (uint32_t c0 = 0; c0 < g.count; c0 += kChunk) {
        const uint32_t nc = min(kChunk, g.count - c0);
        __syncthreads();  // (the previous chunk is no longer read)
        if (threadIdx.x < nc) {
            const uint32_t k = g.first + c0 + threadIdx.x;
            const double c = term_coef[k];
            sz[threadIdx.x] = term_z[k];
            s_re[threadIdx.x] = term_odd[k] ? 0.0 : c;
            s_im[threadIdx.x] = term_odd[k] ? c : 0.0;
        }
        __syncthreads();
        for (uint64_t p = uint64_t(blockIdx.x) * 256 + threadIdx.x; p < n_pairs; p += stride) {
            const uint64_t i = ((p & ~low_mask) << 1) | (p & low_mask);  // bit `pivot` of i is 0
            const uint64_t j = i ^ g.x;
            const cx<real> a = st[i], b = st[j];
            const double re = double(a.re) * double(b.re) + double(a.im) * double(b.im);
            const double im = double(a.re) * double(b.im) - double(a.im) * double(b.re);
            double w_re = 0.0, w_im = 0.0;
            for (uint32_t k = 0; k < nc; ++k) {
                const bool minus = __popcll(i & sz[k]) & 1;
                w_re += minus ? -s_re[k] : s_re[k];
                w_im += minus ? -s_im[k] : s_im[k];
            }
            acc += w_re * re + w_im * im;
        }
    }
    const double total = block_sum_256(2.0 * acc, red);
    if (threadIdx.x == 0) partials[(size_t(slot) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = total;
}

hipError_t launch_pauli_groups(int dtype, const void* states, uint64_t state_stride, int n_qubits, int n_slots,
                               int n_groups, const PauliGroup* groups, const uint64_t* term_z,
                               const double* term_coef, const uint32_t* term_odd, int nb, double* partials,
                               hipStream_t stream) {
    const uint64_t n_pairs = uint64_t(1) << (n_qubits - 1);
    dim3 grid(nb, n_groups, n_slots);
    if (dtype == 0)
        hipLaunchKernelGGL(pauli_groups_kernel<double>, grid, dim3(256), 0, stream,
                           reinterpret_cast<const cx<double>*>(states), state_stride, n_pairs, groups, term_z,
                           term_coef, term_odd, partials);
    else
        hipLaunchKernelGGL(pauli_groups_kernel<float>, grid, dim3(256), 0, stream,
                           reinterpret_cast<const cx<float>*>(states), state_stride, n_pairs, groups, term_z,
                           term_coef, term_odd, partials);
    return hipGetLastError();
}

// out[out_index[slot]] = (fixed-order sum of the slot's group partials) + (sum of its diagonal-table partials)
__global__ void __launch_bounds__(256) pauli_combine_kernel(const double* __restrict__ partials, uint32_t per_slot,
                                                            const double* __restrict__ diag_partials,
                                                            uint32_t diag_per_eval,
                                                            const EvalDesc* __restrict__ evals,
                                                            double* __restrict__ out) {
    __shared__ double red[4];
    const int slot = blockIdx.x;
    const uint32_t out_index = evals[slot].out_index;
    double acc = 0.0;
    const double* p = partials + size_t(slot) * per_slot;
    for (uint32_t i = threadIdx.x; i < per_slot; i += 256) acc += p[i];
    if (diag_partials) {
        const double* d = diag_partials + size_t(out_index) * diag_per_eval;
        for (uint32_t i = threadIdx.x; i < diag_per_eval; i += 256) acc += d[i];
    }
    const double total = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[out_index] = total;
}

hipError_t launch_pauli_combine(const double* partials, uint32_t per_slot, const double* diag_partials,
                                uint32_t diag_per_eval, int n_slots, const EvalDesc* evals, double* out,
                                hipStream_t stream) {
    hipLaunchKernelGGL(pauli_combine_kernel, dim3(n_slots), dim3(256), 0, stream, partials, per_slot, diag_partials,
                       diag_per_eval, evals, out);
    return hipGetLastError();
}

// ---- state read-out ------------------------------------------------------------------------------------
template <typename real>
__global__ void __launch_bounds__(256) probabilities_kernel(const cx<real>* __restrict__ st_all, uint64_t dim,
                                                            double* __restrict__ probs_all) {
    const cx<real>* __restrict__ st = st_all + uint64_t(blockIdx.y) * dim;
    double* __restrict__ probs = probs_all + uint64_t(blockIdx.y) * dim;
    const uint64_t stride = uint64_t(gridDim.x) * 256;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < dim; i += stride) {
        const double re = double(st[i].re), im = double(st[i].im);
        probs[i] = re * re + im * im;
    }
}

template <typename real>
__global__ void __launch_bounds__(256) state_to_f64_kernel(const cx<real>* __restrict__ st, uint64_t dim,
                                                           double* __restrict__ out) {
    const uint64_t stride = uint64_t(gridDim.x) * 256;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < dim; i += stride) {
        out[2 * i] = double(st[i].re);
        out[2 * i + 1] = double(st[i].im);
    }
}

// ---- sampling -------------------------------------------------------------------------------------------
// Inverse-CDF sampling without materialising a 2^n-entry CDF: sums of 64-entry chunks (one coalesced 512-byte read
// and a fixed-order shuffle tree per chunk), one inclusive scan over the chunk sums, then per shot a binary search
// over chunks and a walk of at most 64 entries inside the chosen chunk.  (With 4096-entry chunks the walk was a
// serial chain of ~2000 dependent loads per shot: 0.5 ms per batch at 12 qubits.)
constexpr uint32_t kSampleChunk = 64;

__global__ void __launch_bounds__(256) chunk_sums_kernel(const double* __restrict__ probs_all, uint64_t dim,
                                                         double* __restrict__ sums_all, uint32_t n_chunks) {
    const double* __restrict__ probs = probs_all + uint64_t(blockIdx.y) * dim;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t chunk = blockIdx.x * 4 + wave; chunk < n_chunks; chunk += gridDim.x * 4) {
        const uint64_t idx = uint64_t(chunk) * kSampleChunk + lane;
        double v = idx < dim ? probs[idx] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) sums_all[size_t(blockIdx.y) * n_chunks + chunk] = v;
    }
}

// in-place inclusive scan of `n` chunk sums by one workgroup per slot (n <= 2^20: a few microseconds)
__global__ void __launch_bounds__(256) scan_sums_kernel(double* __restrict__ sums_all, uint32_t n) {
    __shared__ double part[256];
    double* __restrict__ sums = sums_all + size_t(blockIdx.x) * n;
    const uint32_t per = (n + 255) / 256;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    double acc = 0.0;
    for (uint32_t i = lo; i < hi; ++i) acc += sums[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int i = 0; i < 256; ++i) {
            const double v = part[i];
            part[i] = run;
            run += v;
        }
    }
    __syncthreads();
    double run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) {
        run += sums[i];
        sums[i] = run;
    }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// the uniform number in [0, 1) of one shot of one evaluation: a stream per evaluation, a counter per shot
__device__ __forceinline__ double shot_uniform(uint64_t seed, uint32_t eval, uint32_t shot) {
    const uint64_t stream = splitmix64(seed + 0xD1B54A32D192ED03ull * (uint64_t(eval) + 1));
    const uint64_t bits = splitmix64(stream ^ splitmix64(uint64_t(shot) + 1));
    return double(bits >> 11) * (1.0 / 9007199254740992.0);
}

// blockIdx.y = slot; slot s samples with stream seed splitmix64(seed + stream_of[s]) and writes shots of
// evaluation out_index[s]; when `diag` is given each sample's diagonal value D[state] is gathered too.
__global__ void __launch_bounds__(256) sample_kernel(const double* __restrict__ probs_all, uint64_t dim,
                                                     const double* __restrict__ scanned_all, uint32_t n_chunks,
                                                     int shots, uint64_t seed, uint32_t first_eval,
                                                     const double* __restrict__ diag, uint64_t* __restrict__ out,
                                                     double* __restrict__ out_values, const EvalDesc* __restrict__ evals) {
    const int shot = blockIdx.x * blockDim.x + threadIdx.x;
    if (shot >= shots) return;
    const uint32_t slot = blockIdx.y, eval = evals ? evals[slot].out_index : first_eval + slot;
    const double* __restrict__ probs = probs_all + uint64_t(slot) * dim;
    const double* __restrict__ scanned = scanned_all + size_t(slot) * n_chunks;
    const double total = scanned[n_chunks - 1];
    const double u = shot_uniform(seed, eval, uint32_t(shot)) * total;  // [0, total)
    // first chunk whose inclusive prefix exceeds u
    uint32_t lo = 0, hi = n_chunks - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (scanned[mid] > u) hi = mid; else lo = mid + 1;
    }
    double run = lo ? scanned[lo - 1] : 0.0;
    const uint64_t first = uint64_t(lo) * kSampleChunk;
    const uint64_t last = min(dim, first + kSampleChunk) - 1;
    uint64_t idx = first;
    for (; idx < last; ++idx) {
        run += probs[idx];
        if (run > u) break;
    }
    // walk back over zero-probability states a rounding tie could have selected
    while (idx > first && probs[idx] == 0.0) --idx;
    const size_t o = size_t(eval) * size_t(shots) + size_t(shot);
    out[o] = idx;
    if (out_values) out_values[o] = diag[idx];
}

hipError_t launch_sample(const double* probs, uint64_t dim, int n_slots, double* chunk_sums, int shots, uint64_t seed,
                         uint32_t first_eval, const double* diag, uint64_t* out, double* out_values,
                         hipStream_t stream, const EvalDesc* evals) {
    const uint32_t n_chunks = uint32_t((dim + kSampleChunk - 1) / kSampleChunk);
    const uint32_t sum_blocks = (n_chunks + 3) / 4 < 4096 ? (n_chunks + 3) / 4 : 4096;
    hipLaunchKernelGGL(chunk_sums_kernel, dim3(sum_blocks, n_slots), dim3(256), 0, stream, probs, dim, chunk_sums, n_chunks);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(n_slots), dim3(256), 0, stream, chunk_sums, n_chunks);
    hipLaunchKernelGGL(sample_kernel, dim3((shots + 255) / 256, n_slots), dim3(256), 0, stream, probs, dim, chunk_sums,
                       n_chunks, shots, seed, first_eval, diag, out, out_values, evals);
    return hipGetLastError();
}

uint32_t sample_chunk_count(uint64_t dim) { return uint32_t((dim + kSampleChunk - 1) / kSampleChunk); }

// CVaR of one evaluation's sample values per workgroup (expectation_calculation.py:27-40 restated for equally weighted
// samples): sort ascending, take probability mass alpha from the low end.
__global__ void __launch_bounds__(1024) cvar_kernel(const double* __restrict__ values, int shots, int padded, double alpha,
                                                   double* __restrict__ out) {
    extern __shared__ double sorted[];
    const double* v = values + size_t(blockIdx.x) * size_t(shots);
    for (int i = threadIdx.x; i < padded; i += blockDim.x) sorted[i] = i < shots ? v[i] : __builtin_huge_val();
    __syncthreads();
    for (int k = 2; k <= padded; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < padded; i += blockDim.x) {
                const int partner = i ^ j;
                if (partner > i) {
                    const bool ascending = (i & k) == 0;
                    const double a = sorted[i], b = sorted[partner];
                    if ((a > b) == ascending) {
                        sorted[i] = b;
                        sorted[partner] = a;
                    }
                }
            }
            __syncthreads();
        }
    // mass alpha * shots from the low end: `whole` samples entirely, the next one with the remaining fraction
    const double mass = alpha * double(shots);
    int whole = int(floor(mass + 1e-12));
    if (whole > shots) whole = shots;
    // fixed-order sum: every thread adds its strided share, then a tree over the threads (in place of `sorted`'s tail)
    double part = 0.0;
    for (int i = threadIdx.x; i < whole; i += blockDim.x) part += sorted[i];
    const double boundary = whole < shots ? sorted[whole] : 0.0;
    __syncthreads();
    double* tree = sorted;  // (values no longer needed)
    tree[threadIdx.x] = part;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (int(threadIdx.x) < s) tree[threadIdx.x] += tree[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double total = tree[0];
        if (whole < shots && mass - double(whole) > 1e-12) total += (mass - double(whole)) * boundary;
        out[blockIdx.x] = total / mass;
    }
}

hipError_t launch_cvar(const double* values, int n_evals, int shots, double alpha, double* out, hipStream_t stream) {
    if (shots < 1 || shots > kCvarMaxShots || !(alpha > 0.0) || alpha > 1.0) return hipErrorInvalidValue;
    int padded = 256;  // (at least one value per thread: the reduction tree reuses the buffer)
    while (padded < shots) padded <<= 1;
    // a thread per value up to 1024 (measured at 1024 shots: 30 us per launch with 256 threads -- the sort is a chain of
    // 55 barrier-separated steps, and four values per thread and step made each of them longer)
    const int threads = padded < 1024 ? padded : 1024;
    hipLaunchKernelGGL(cvar_kernel, dim3(n_evals), dim3(threads), size_t(padded) * sizeof(double), stream, values, shots, padded,
                       alpha, out);
    return hipGetLastError();
}

// ---- sampling split evaluations (kernels.hpp: launch_split_tables / launch_split_sample) ---------------------------
// Scratch of one evaluation: running sums of the marginal of x (one per value of x), then the Gram table T[pi][y1]
// (pi < J^2, y1 < 2^(|Y| - 6)).
// T's rows: pi = j < J: G_jj; then for every pair j < j' two rows, 2 Re G_jj' and -2 Im G_jj', with
// G_jj'[y1] = sum over the block's y of Y_j[y] conj(Y_j'[y]) -- so that the probability of (x, block y1) is
//     sum_j T[j] |X_j|^2 + sum_{j<j'} T[a] Re(X_j conj X_j') + T[b] Im(X_j conj X_j')        (split_quad).
constexpr uint32_t kSplitSampleBlockBits = 6;
constexpr int kSplitSampleShotsPerBlock = 32;

size_t split_sample_slot_doubles(int side_bits) {
    // running sums: one per value of x; Gram table: J^2 2^(|Y| - 6) <= J 2^(side_bits - 6) <= 2^(side_bits - 3)
    const size_t table = std::max<size_t>(64, size_t(1) << (side_bits > 3 ? side_bits - 3 : 0));
    return (size_t(1) << side_bits) + table;
}
constexpr uint32_t kSplitSampleLdsDoubles = 6144;  // the Gram table is staged in LDS up to this size, read from L2 beyond

// v of the lane `shift` places down its row of 16 lanes / of lane 15 of the previous row / of lane 31 (DPP; 0.0 for a
// lane that has no such source or whose row is masked out)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

// inclusive sums over the lanes of a wave, fixed tree, no LDS traffic: four doubling steps inside the rows of 16, then
// the row totals
__device__ __forceinline__ double wave_inclusive(double v) {
    v += dpp_f64<0x111, 0xf>(v);  // row_shr:1
    v += dpp_f64<0x112, 0xf>(v);  // row_shr:2
    v += dpp_f64<0x114, 0xf>(v);  // row_shr:4
    v += dpp_f64<0x118, 0xf>(v);  // row_shr:8
    v += dpp_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_f64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ double read_lane(double v, uint32_t lane) {  // lane: uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), int(lane)),
                            __builtin_amdgcn_readlane(__double2loint(v), int(lane)));
}

// Which entry of a J x J Hermitian form lane pi < J^2 stands for (the rows of T above): part 0 = the diagonal entry ja,
// part 1 / 2 = the real / imaginary part of pair (ja, jb), ja < jb.
template <int J>
__device__ __forceinline__ void split_entry_of(uint32_t pi, uint32_t* ja_out, uint32_t* jb_out, uint32_t* part_out) {
    uint32_t ja = pi, jb = pi, part = 0;
    if (pi >= uint32_t(J)) {
        uint32_t o = (pi - J) >> 1;
        part = 1 + ((pi - J) & 1u);
        ja = 0;
        while (o >= uint32_t(J) - 1 - ja) {
            o -= uint32_t(J) - 1 - ja;
            ++ja;
        }
        jb = ja + 1 + o;
    }
    *ja_out = ja;
    *jb_out = jb;
    *part_out = part;
}
// ... and the value that goes with it for the vector X: |X_ja|^2, Re(X_ja conj X_jb), Im(X_ja conj X_jb)
__device__ __forceinline__ double split_entry_value(uint32_t part, double ar, double ai, double br, double bi) {
    return part == 2 ? fma(ai, br, -ar * bi) : fma(ar, br, ai * bi);
}

// sum_pi T[pi][col] w_pi, lane pi holding w_pi: the uniform factor of every term comes out of its lane as a scalar
// (v_readlane), so the form costs J^2 loads and J^2 fused multiply-adds per lane and no registers for X or w
template <int J>
__device__ __forceinline__ double split_quad_table(const double* __restrict__ T, uint32_t stride, uint32_t col, double w_of_lane) {
    double t = 0.0;
#pragma unroll
    for (uint32_t pi = 0; pi < uint32_t(J * J); ++pi) t = fma(T[pi * stride + col], read_lane(w_of_lane, pi), t);
    return t;
}
// the same form with the table entries uniform (lane pi holds M_pi) and the vector per lane
template <int J>
__device__ __forceinline__ double split_quad_vector(double m_of_lane, const double (&xr)[J], const double (&xi)[J]) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) t = fma(read_lane(m_of_lane, uint32_t(j)), fma(xr[j], xr[j], xi[j] * xi[j]), t);
    uint32_t row = J;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int jp = j + 1; jp < J; ++jp) {
            t = fma(read_lane(m_of_lane, row), fma(xr[j], xr[jp], xi[j] * xi[jp]), t);
            t = fma(read_lane(m_of_lane, row + 1), fma(xi[j], xr[jp], -xr[j] * xi[jp]), t);
            row += 2;
        }
    return t;
}

// Gram blocks of one evaluation (blockIdx.y), spread over the waves of gridDim.x workgroups.  A wave takes a block of 64
// y: its J rows of Y go to the wave's own LDS region (coalesced loads; y-major with one entry of padding per y, so that
// neither the writes nor the reads below collide on banks); lane (sub, pi) adds row pi's products over the sub-th run
// of J^2 values of y, the runs are added across lanes.
template <typename real, int J>
__device__ void split_gram_body(const cx<real>* __restrict__ Y, uint32_t bits_y, double* __restrict__ T, cx<real>* stage_all) {
    constexpr uint32_t NQ = J * J;  // rows of T
    constexpr uint32_t PITCH = J + 1;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n_waves = blockDim.x >> 6;
    const uint32_t ly2 = bits_y < kSplitSampleBlockBits ? bits_y : kSplitSampleBlockBits;
    const uint32_t ny2 = 1u << ly2, ny1 = 1u << (bits_y - ly2);
    cx<real>* stage = stage_all + size_t(wave) * PITCH * 64;
    const uint32_t pi = lane % NQ, sub = lane / NQ;
    uint32_t ja, jb, part;
    split_entry_of<J>(pi, &ja, &jb, &part);
    for (uint32_t y1 = blockIdx.x * n_waves + wave; y1 < ny1; y1 += gridDim.x * n_waves) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            cx<real> v{real(0), real(0)};
            if (lane < ny2) v = Y[(size_t(j) << bits_y) + size_t(y1) * ny2 + lane];
            stage[lane * PITCH + uint32_t(j)] = v;
        }
        double acc = 0.0;
#pragma unroll 4
        for (uint32_t i = 0; i < NQ; ++i) {
            const uint32_t y2 = sub * NQ + i;
            const cx<real> a = stage[y2 * PITCH + ja], b = stage[y2 * PITCH + jb];
            acc += split_entry_value(part, double(a.re), double(a.im), double(b.re), double(b.im));
        }
        for (uint32_t off = NQ; off < 64; off <<= 1) acc += __shfl_xor(acc, int(off));
        acc = part == 0 ? acc : part == 1 ? 2.0 * acc : -2.0 * acc;
        if (lane < NQ) T[pi * ny1 + y1] = acc;
    }
}

constexpr unsigned kSplitGramParts = 4;  // workgroups per evaluation

template <typename real>
__global__ void __launch_bounds__(256, 2) split_gram_kernel(const uint32_t* __restrict__ plan_arena, const EvalDesc* __restrict__ evals,
                                                            const cx<real>* __restrict__ sides, uint64_t side_stride,
                                                            double* __restrict__ scratch, uint32_t slot_doubles, uint32_t cum_doubles) {
    __shared__ cx<real> stage[4 * 9 * 64];
    const EvalDesc ev = evals[blockIdx.y];
    if (!(ev.flags & kEvalSide)) return;
    const uint32_t* sp = plan_arena + ev.split_base;
    const uint32_t n_keys = sp[0], bits_y = sp[2];
    const bool swap = sp[3] & 1u;
    const cx<real>* Y = sides + uint64_t(ev.state_slot) * side_stride + (swap ? 0 : side_stride >> 1);
    double* T = scratch + size_t(blockIdx.y) * slot_doubles + cum_doubles;
    if (n_keys == 0)
        split_gram_body<real, 1>(Y, bits_y, T, stage);
    else if (n_keys == 1)
        split_gram_body<real, 2>(Y, bits_y, T, stage);
    else if (n_keys == 2)
        split_gram_body<real, 4>(Y, bits_y, T, stage);
    else
        split_gram_body<real, 8>(Y, bits_y, T, stage);
}

// The marginal of x and its running sums, one workgroup per evaluation: the Gram matrix of the whole of Y (the sum of
// the blocks, in a fixed order), its quadratic form for every x, an inclusive scan.
template <typename real, int J>
__device__ void split_marginal_body(const cx<real>* __restrict__ X, uint32_t bits_x, uint32_t bits_y, double* __restrict__ cum,
                                    const double* __restrict__ T, double* lds) {
    constexpr uint32_t NQ = J * J;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n_threads = blockDim.x;
    const uint32_t ly2 = bits_y < kSplitSampleBlockBits ? bits_y : kSplitSampleBlockBits;
    const uint32_t ny1 = 1u << (bits_y - ly2);
    // thread (piece, pi) adds its piece of row pi, then the pieces in order
    const uint32_t pi = tid % NQ, piece = tid / NQ, n_pieces = n_threads / NQ;
    double* partial = lds;  // [n_threads]
    {
        const uint32_t per = (ny1 + n_pieces - 1) / n_pieces;
        const uint32_t lo = min(ny1, piece * per), hi = min(ny1, lo + per);
        double m = 0.0;
        for (uint32_t y1 = lo; y1 < hi; ++y1) m += T[pi * ny1 + y1];
        partial[tid] = m;
    }
    __syncthreads();
    double m_of_lane = 0.0;
    for (uint32_t q = 0; q < n_pieces; ++q) m_of_lane += partial[q * NQ + (lane % NQ)];
    const uint32_t nx = 1u << bits_x, per = (nx + n_threads - 1) / n_threads;
    const uint32_t lo = min(nx, tid * per), hi = min(nx, lo + per);
    double mine = 0.0;
    // (every lane makes every trip, a lane without an x with a stand-in: the form reads its uniform factors out of lanes,
    // and a lane that sits out a divergent loop need not hold its value any more)
    for (uint32_t i = 0; i < per; ++i) {
        const bool valid = lo + i < hi;
        const uint32_t x = valid ? lo + i : 0u;
        double xr[J], xi[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const cx<real> v = X[(size_t(j) << bits_x) + x];
            xr[j] = double(v.re);
            xi[j] = double(v.im);
        }
        double p = split_quad_vector<J>(m_of_lane, xr, xi);
        p = valid && p > 0.0 ? p : 0.0;  // (a quadratic form of a Gram matrix: negative only by rounding)
        if (valid) cum[x] = p;
        mine += p;
    }
    const double inc = wave_inclusive(mine);
    __syncthreads();  // (partial[] is reused)
    double* wave_total = lds;  // [n_waves]
    if (lane == 63) wave_total[wave] = inc;
    __syncthreads();
    double run = inc - mine;
    for (uint32_t w = 0; w < wave; ++w) run += wave_total[w];
    for (uint32_t x = lo; x < hi; ++x) {
        run += cum[x];
        cum[x] = run;
    }
}

template <typename real>
__global__ void __launch_bounds__(256, 2) split_marginal_kernel(const uint32_t* __restrict__ plan_arena, const EvalDesc* __restrict__ evals,
                                                                const cx<real>* __restrict__ sides, uint64_t side_stride,
                                                                double* __restrict__ scratch, uint32_t slot_doubles, uint32_t cum_doubles) {
    __shared__ double lds[256];
    const EvalDesc ev = evals[blockIdx.x];
    if (!(ev.flags & kEvalSide)) return;
    const uint32_t* sp = plan_arena + ev.split_base;
    const uint32_t n_keys = sp[0], bits_x = sp[1], bits_y = sp[2];
    const bool swap = sp[3] & 1u;
    const cx<real>* X = sides + uint64_t(ev.state_slot) * side_stride + (swap ? side_stride >> 1 : 0);
    double* cum = scratch + size_t(blockIdx.x) * slot_doubles;
    const double* T = cum + cum_doubles;
    if (n_keys == 0)
        split_marginal_body<real, 1>(X, bits_x, bits_y, cum, T, lds);
    else if (n_keys == 1)
        split_marginal_body<real, 2>(X, bits_x, bits_y, cum, T, lds);
    else if (n_keys == 2)
        split_marginal_body<real, 4>(X, bits_x, bits_y, cum, T, lds);
    else
        split_marginal_body<real, 8>(X, bits_x, bits_y, cum, T, lds);
}

// One inverse-CDF step over the lanes of a wave: the first lane of positive weight whose running sum exceeds r, and what
// is left of r inside that lane's weight.  Rounding may leave no such lane (the caller then takes the last lane of
// positive weight).
struct LanePick {
    bool crossed;
    uint32_t lane;
    double rest;
};
__device__ __forceinline__ LanePick select_lane(double weight, double inclusive, double r, uint64_t positive) {
    const uint64_t crossed = __ballot(inclusive > r) & positive;
    LanePick p{crossed != 0, 0u, 0.0};
    if (crossed) {
        p.lane = uint32_t(__builtin_ctzll(crossed));
        const double left = r - read_lane(inclusive - weight, p.lane);
        p.rest = left > 0.0 ? left : 0.0;
    }
    return p;
}

__device__ __forceinline__ uint32_t deposit_bits(uint32_t v, uint32_t mask) {
    uint32_t out = 0;
    while (mask) {
        const uint32_t low = mask & (0u - mask);
        if (v & 1u) out |= low;
        v >>= 1;
        mask ^= low;
    }
    return out;
}

// |psi(x, y)|^2 for this lane's y; lane j < J holds X_j[x] (re, im)
template <typename real, int J>
__device__ __forceinline__ double split_probability(const cx<real>* __restrict__ Y, uint32_t bits_y, uint32_t y, double x_re_of_lane,
                                                    double x_im_of_lane) {
    double pr = 0.0, pi = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const cx<real> v = Y[(size_t(j) << bits_y) + y];
        const double yr = double(v.re), yi = double(v.im);
        const double xr = read_lane(x_re_of_lane, uint32_t(j)), xi = read_lane(x_im_of_lane, uint32_t(j));
        pr = fma(xr, yr, fma(-xi, yi, pr));
        pi = fma(xr, yi, fma(xi, yr, pi));
    }
    return fma(pr, pr, pi * pi);
}

// One wave per shot, three inverse-CDF levels: x (running sums of its marginal, 64 candidates per step), the block y1
// of 64 values of y (quadratic forms of the block's Gram matrix, a candidate per lane), y inside the block (the
// amplitudes themselves, a value of y per lane).  The levels above the last work with sums that carry rounding errors;
// the last one only ever selects a lane whose amplitude is not zero, and when rounding has led to a block without any
// (probability ~1e-16) the blocks are walked until one has: a state of probability zero is never drawn.
// What is uniform over the wave (the J values of X at the shot's x, the J^2 products of the quadratic form) lives in
// ONE register, entry pi in lane pi, and is read out as scalars where it is used.
template <typename real, int J>
__device__ void split_sample_body(const cx<real>* __restrict__ X, const cx<real>* __restrict__ Y, uint32_t bits_x, uint32_t bits_y,
                                  uint32_t mask_x, uint32_t mask_y, const double* __restrict__ cum, const double* __restrict__ Tg,
                                  double* lds_table, uint32_t lds_doubles, int shots, uint64_t seed, uint32_t eval,
                                  const double* __restrict__ diag, uint64_t* __restrict__ out, double* __restrict__ out_values) {
    constexpr int SPB = kSplitSampleShotsPerBlock;
    constexpr uint32_t NQ = J * J;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ly2 = bits_y < kSplitSampleBlockBits ? bits_y : kSplitSampleBlockBits;
    const uint32_t ny2 = 1u << ly2, ny1 = 1u << (bits_y - ly2);
    const uint32_t nx = 1u << bits_x;
    const double* T = Tg;  // (a table too large for the LDS share is read where it is: L2)
    if (ny1 > 1 && NQ * ny1 <= lds_doubles) {  // (uniform: the workgroup's evaluation)
        for (uint32_t i = tid; i < NQ * ny1; i += blockDim.x) lds_table[i] = Tg[i];
        __syncthreads();
        T = lds_table;
    }
    uint32_t ja, jb, part;
    split_entry_of<J>(lane % NQ, &ja, &jb, &part);
    const double total = cum[nx - 1];
    const double below_total = __longlong_as_double(__double_as_longlong(total) - 1);
    const int shot0 = int(blockIdx.x) * SPB;
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t my_y2 = lane < ny2 ? lane : 0u;
    // (the first step of the search for x looks at the same entries for every shot)
    const uint32_t step0 = bits_x < 6u ? bits_x : 6u;
    const double top = cum[(((lane < (1u << step0) ? lane : (1u << step0) - 1u) + 1u) << (bits_x - step0)) - 1u];
    for (uint32_t s = wave; s < uint32_t(SPB) && shot0 + int(s) < shots; s += n_waves) {
        double r = shot_uniform(seed, eval, uint32_t(shot0) + s) * total;
        if (!(r < total)) r = below_total;  // (so that every step below finds a running sum above r)
        // x: six bits per step, lane c looks at the last entry of the c-th part of what is left; the sum in front of
        // the chosen part is the neighbour's entry (or what the step before left in front)
        double before = 0.0;
        uint32_t x;
        {
            const uint64_t crossed = __ballot(top > r && lane < (1u << step0));
            x = crossed ? uint32_t(__builtin_ctzll(crossed)) : (1u << step0) - 1u;
            if (x) before = read_lane(top, x - 1u);
        }
        for (uint32_t left = bits_x - step0; left > 0;) {
            const uint32_t step = left < 6u ? left : 6u;
            left -= step;
            const uint32_t c = lane < (1u << step) ? lane : (1u << step) - 1u;
            const double v = cum[((((x << step) | c) + 1u) << left) - 1u];
            const uint64_t crossed = __ballot(v > r && lane < (1u << step));
            const uint32_t pick = crossed ? uint32_t(__builtin_ctzll(crossed)) : (1u << step) - 1u;
            if (pick) before = read_lane(v, pick - 1u);
            x = (x << step) | pick;
        }
        x = __builtin_amdgcn_readfirstlane(x);
        r = r > before ? r - before : 0.0;
        cx<real> xa = X[(size_t(ja) << bits_x) + x], xb = X[(size_t(jb) << bits_x) + x];
        // the block of y
        uint32_t y1 = 0;
        if (ny1 > 1) {
            const double w = split_entry_value(part, double(xa.re), double(xa.im), double(xb.re), double(xb.im));
            bool found = false;
            uint32_t last_positive = 0;
            double carry = 0.0, rest = __builtin_huge_val();
            for (uint32_t base = 0; base < ny1; base += 64) {
                const uint32_t cand = base + lane;
                double t = split_quad_table<J>(T, ny1, cand < ny1 ? cand : ny1 - 1u, w);
                t = cand < ny1 && t > 0.0 ? t : 0.0;
                const double inc = wave_inclusive(t);
                const uint64_t positive = __ballot(t > 0.0);
                if (!found) {
                    const LanePick p = select_lane(t, inc, r - carry, positive);
                    if (p.crossed) {
                        found = true;
                        y1 = base + p.lane;
                        rest = p.rest;
                    }
                }
                if (positive) last_positive = base + 63u - uint32_t(__builtin_clzll(positive));
                carry += read_lane(inc, 63u);
            }
            if (!found) y1 = last_positive;  // (rest = +inf: its last value of y)
            r = rest;
        }
        y1 = __builtin_amdgcn_readfirstlane(y1);
        // y inside the block
        double q = split_probability<real, J>(Y, bits_y, y1 * ny2 + my_y2, double(xa.re), double(xa.im));
        q = lane < ny2 ? q : 0.0;
        uint64_t positive = __ballot(q > 0.0);
        uint32_t y2;
        if (positive) {
            const double inc = wave_inclusive(q);
            const LanePick p = select_lane(q, inc, r, positive);
            y2 = p.crossed ? p.lane : 63u - uint32_t(__builtin_clzll(positive));
        } else {
            // rounding has led to a block in which every amplitude is zero: walk on, block by block
            for (uint32_t tries = 0; !positive && tries < nx * ny1; ++tries) {
                if (++y1 == ny1) {
                    y1 = 0;
                    x = (x + 1u) & (nx - 1u);
                    xa = X[(size_t(ja) << bits_x) + x];
                }
                q = split_probability<real, J>(Y, bits_y, y1 * ny2 + my_y2, double(xa.re), double(xa.im));
                positive = __ballot(q > 0.0 && lane < ny2);
            }
            y2 = positive ? uint32_t(__builtin_ctzll(positive)) : 0u;
        }
        if (lane == 0) {
            const uint32_t index = deposit_bits(x, mask_x) | deposit_bits(y1 * ny2 + y2, mask_y);
            const size_t o = size_t(eval) * size_t(shots) + size_t(shot0) + s;
            out[o] = index;
            if (out_values) out_values[o] = diag[index];
        }
    }
}

template <typename real>
__global__ void __launch_bounds__(256, 8) split_sample_kernel(const uint32_t* __restrict__ plan_arena, const EvalDesc* __restrict__ evals,
                                                              const cx<real>* __restrict__ sides, uint64_t side_stride,
                                                              const double* __restrict__ scratch, uint32_t slot_doubles,
                                                              uint32_t cum_doubles, uint32_t lds_doubles, int shots, uint64_t seed,
                                                              const double* __restrict__ diag, uint64_t* __restrict__ out,
                                                              double* __restrict__ out_values) {
    extern __shared__ __align__(16) double split_lds[];
    const EvalDesc ev = evals[blockIdx.y];
    if (!(ev.flags & kEvalSide)) return;
    const uint32_t* sp = plan_arena + ev.split_base;
    const uint32_t n_keys = sp[0], bits_x = sp[1], bits_y = sp[2];
    const bool swap = sp[3] & 1u;
    const uint32_t mask_x = sp[kSplitMaskX], mask_y = sp[kSplitMaskY];
    const cx<real>* ta = sides + uint64_t(ev.state_slot) * side_stride;
    const cx<real>* X = ta + (swap ? side_stride >> 1 : 0);
    const cx<real>* Y = ta + (swap ? 0 : side_stride >> 1);
    const double* cum = scratch + size_t(blockIdx.y) * slot_doubles;
    const double* T = cum + cum_doubles;
    if (n_keys == 0)
        split_sample_body<real, 1>(X, Y, bits_x, bits_y, mask_x, mask_y, cum, T, split_lds, lds_doubles, shots, seed, ev.out_index, diag, out, out_values);
    else if (n_keys == 1)
        split_sample_body<real, 2>(X, Y, bits_x, bits_y, mask_x, mask_y, cum, T, split_lds, lds_doubles, shots, seed, ev.out_index, diag, out, out_values);
    else if (n_keys == 2)
        split_sample_body<real, 4>(X, Y, bits_x, bits_y, mask_x, mask_y, cum, T, split_lds, lds_doubles, shots, seed, ev.out_index, diag, out, out_values);
    else
        split_sample_body<real, 8>(X, Y, bits_x, bits_y, mask_x, mask_y, cum, T, split_lds, lds_doubles, shots, seed, ev.out_index, diag, out, out_values);
}

hipError_t launch_split_tables(int dtype, int side_bits, unsigned n_evals, double* scratch, hipStream_t stream, const PassArgs& a) {
    if (n_evals == 0) return hipSuccess;
    const uint32_t slot = uint32_t(split_sample_slot_doubles(side_bits)), cum = 1u << side_bits;
    const dim3 gram_grid(kSplitGramParts, n_evals);
    if (dtype == 0) {
        hipLaunchKernelGGL(split_gram_kernel<double>, gram_grid, dim3(256), 0, stream, a.plan, a.evals,
                           static_cast<const cx<double>*>(a.wtab), a.wtab_stride, scratch, slot, cum);
        hipLaunchKernelGGL(split_marginal_kernel<double>, dim3(n_evals), dim3(256), 0, stream, a.plan, a.evals,
                           static_cast<const cx<double>*>(a.wtab), a.wtab_stride, scratch, slot, cum);
    } else {
        hipLaunchKernelGGL(split_gram_kernel<float>, gram_grid, dim3(256), 0, stream, a.plan, a.evals,
                           static_cast<const cx<float>*>(a.wtab), a.wtab_stride, scratch, slot, cum);
        hipLaunchKernelGGL(split_marginal_kernel<float>, dim3(n_evals), dim3(256), 0, stream, a.plan, a.evals,
                           static_cast<const cx<float>*>(a.wtab), a.wtab_stride, scratch, slot, cum);
    }
    return hipGetLastError();
}

hipError_t launch_split_sample(int dtype, int side_bits, unsigned n_evals, const double* scratch, int shots, uint64_t seed,
                               const double* diag, uint64_t* out, double* out_values, hipStream_t stream, const PassArgs& a,
                               uint32_t table_doubles) {
    if (n_evals == 0 || shots <= 0) return hipSuccess;
    const uint32_t slot = uint32_t(split_sample_slot_doubles(side_bits)), cum = 1u << side_bits;
    // LDS: the Gram table of the workgroup's evaluation (the caller may know that the launch's tables are smaller
    // than the bound: more workgroups per CU); a table beyond kSplitSampleLdsDoubles stays in memory
    const uint32_t lds_doubles = std::min(kSplitSampleLdsDoubles, table_doubles ? std::min(table_doubles, slot - cum) : slot - cum);
    const size_t lds = size_t(lds_doubles) * sizeof(double);
    const dim3 grid(unsigned((shots + kSplitSampleShotsPerBlock - 1) / kSplitSampleShotsPerBlock), n_evals);
    if (dtype == 0)
        hipLaunchKernelGGL(split_sample_kernel<double>, grid, dim3(256), lds, stream, a.plan, a.evals,
                           static_cast<const cx<double>*>(a.wtab), a.wtab_stride, scratch, slot, cum, lds_doubles, shots, seed, diag, out, out_values);
    else
        hipLaunchKernelGGL(split_sample_kernel<float>, grid, dim3(256), lds, stream, a.plan, a.evals,
                           static_cast<const cx<float>*>(a.wtab), a.wtab_stride, scratch, slot, cum, lds_doubles, shots, seed, diag, out, out_values);
    return hipGetLastError();
}

// ---- split evaluations under a quadratic diagonal operator (kernels.hpp: launch_factor) -----------------------------
// Weighted Gram matrices of one side: entry pi of  sum_x w(x) X(x) X(x)^dagger  for the weights w = 1, D(x, 0) and the
// bits of x.  Lanes as in split_gram_body: (run, pi); a wave takes blocks of 64 table entries (its rows staged in LDS
// together with the block's 64 values of D; the next block's rows are already on their way while this one is added up).
constexpr uint32_t kFactorWeights = 18;  // 1, D, and up to 16 bits
constexpr uint32_t kFactorPitch = 65;    // doubles per row of a Gram table in LDS (64 entries + one: rows on different banks)

// out (LDS, [weight][64]) may overlap other waves' staging regions: it is written after a workgroup barrier.
// AHEAD: the next block's rows are fetched while this one is added up (32 more registers at eight terms; the tail of the pass
// kernel, which has none to spare, fetches each block when it needs it).
// The values of D for a wave's first kFactorDAhead blocks (first, first + step, ..; below `end`): scattered reads of a table
// the size of the state -- the longest latency of the whole Gram phase and independent of the side's state, so the callers
// ask for them BEFORE they wait for the side's stores; the bodies keep the queue kFactorDAhead blocks ahead.
constexpr int kFactorDAhead = 4;
__device__ __forceinline__ double factor_d_of_block(const double* __restrict__ diag, uint32_t bits, uint32_t mask, uint32_t blk, uint32_t end) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_local = bits < 6u ? 1u << bits : 64u;
    double d = 0.0;
    // (a side's own table -- mask = its low `bits` bits -- is read in place: the deposit's loop over the mask's bits, taken by every
    // block of every wave, was 5 of a 12-qubit side's 39 us)
    const uint32_t x = blk * 64u + lane;
    if (lane < n_local && blk < end) d = diag[mask == (1u << bits) - 1u ? x : deposit_bits(x, mask)];
    return d;
}
template <int DA>
__device__ __forceinline__ void factor_prefetch_d(double (&dq)[DA], const double* __restrict__ diag, uint32_t bits, uint32_t mask,
                                                  uint32_t first_block, uint32_t block_step, uint32_t end) {
#pragma unroll
    for (int k = 0; k < DA; ++k) dq[k] = factor_d_of_block(diag, bits, mask, first_block + uint32_t(k) * block_step, end);
}
constexpr int kFactorDAheadPairs = 2;  // (the eight-term body has no registers to spare)
__device__ __forceinline__ uint32_t factor_block_count(uint32_t bits) { return bits < 6u ? 1u : 1u << (bits - 6u); }

// (a half side's rows: where each starts, in bytes from the table, so that its entry x lies at x; by value: registers)
struct HalfRows {
    int32_t at[4] = {0, 0, 0, 0};
    bool on = false;
};
template <typename real, int J, bool AHEAD = true>
__device__ __forceinline__ void factor_side_body(const cx<real>* __restrict__ tab, uint32_t bits, uint32_t mask, const double* __restrict__ diag,
                                 uint32_t first_block, uint32_t block_step, cx<real>* stage, double* dstage, double* out,
                                 double (&dq)[kFactorDAhead] QSV_PSTAMP_PARAMS, HalfRows row_bytes = HalfRows{}, uint32_t end_block = 0xffffffffu) {
    // (row_bytes / end_block: a half side -- where each row starts, in bytes from `tab`, so that its entry x lies at x; the blocks end early)
    constexpr uint32_t NQ = J * J;
    constexpr uint32_t LQ = J == 1 ? 0 : J == 2 ? 2 : J == 4 ? 4 : 6;  // log2(NQ)
    constexpr uint32_t PITCH = J + 1;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t pi = lane % NQ, sub = lane / NQ;
    uint32_t ja, jb, part;
    split_entry_of<J>(pi, &ja, &jb, &part);
    const uint32_t n_local = bits < 6u ? 1u << bits : 64u, all_blocks = bits < 6u ? 1u : 1u << (bits - 6u);
    const uint32_t n_blocks = end_block < all_blocks ? end_block : all_blocks;
    double acc_one = 0.0, acc_d = 0.0, acc_low[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, acc_high[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) acc_high[q] = 0.0;
    cx<real> rows[J];
    auto fetch = [&](uint32_t blk) {
        const bool live = lane < n_local && blk < n_blocks;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            rows[j] = cx<real>{real(0), real(0)};
            // (a 32-bit byte offset from the uniform base: one address register per row instead of a 64-bit pointer each)
            const int32_t row = row_bytes.on ? row_bytes.at[j < 4 ? j : 0] : int32_t((uint32_t(j) << bits) * uint32_t(sizeof(cx<real>)));
            const int32_t off = row + int32_t((blk * 64u + lane) * uint32_t(sizeof(cx<real>)));
            if (live) rows[j] = *reinterpret_cast<const cx<real>*>(reinterpret_cast<const char*>(tab) + off);
        }
    };
    if constexpr (AHEAD) fetch(first_block);
    QSV_PSTAMP(4);  // (diagnostic build: the first block's rows)
    for (uint32_t blk = first_block; blk < n_blocks; blk += block_step) {
        if constexpr (!AHEAD) fetch(blk);
        // (ONE product term: a lane's only entry is |its own amplitude|^2 times its own value of D -- nothing goes through LDS)
        const double own_re = double(rows[0].re), own_im = double(rows[0].im), own_d = dq[0];
        if constexpr (J > 1) {
#pragma unroll
            for (int j = 0; j < J; ++j) stage[lane * PITCH + uint32_t(j)] = rows[j];
            dstage[lane] = dq[0];
        }
        // (the queue moves up; the block kFactorDAhead steps on is asked for AFTER this block's rows: loads return in order)
#pragma unroll
        for (int k = 0; k + 1 < kFactorDAhead; ++k) dq[k] = dq[k + 1];
        dq[kFactorDAhead - 1] = factor_d_of_block(diag, bits, mask, blk + uint32_t(kFactorDAhead) * block_step, n_blocks);
        if constexpr (AHEAD) fetch(blk + block_step);
        double s_one = 0.0, s_d = 0.0;
        if constexpr (J == 1) {
            const double p = split_entry_value(0u, own_re, own_im, own_re, own_im);
            s_one += p;
            s_d = fma(own_d, p, s_d);
        } else
#pragma unroll
        for (uint32_t i = 0; i < NQ; ++i) {
            const uint32_t xl = sub * NQ + i;
            const cx<real> a = stage[xl * PITCH + ja], b = stage[xl * PITCH + jb];
            const double p = split_entry_value(part, double(a.re), double(a.im), double(b.re), double(b.im));
            s_one += p;
            s_d = fma(dstage[xl], p, s_d);
#pragma unroll
            for (uint32_t q = 0; q < LQ; ++q)  // (bits of i: known when the loop is unrolled)
                if (i >> q & 1u) acc_low[q] += p;
        }
#pragma unroll
        for (uint32_t q = LQ; q < 6; ++q)  // (bits of the run number: the same for the whole run)
            acc_low[q] += (sub >> (q - LQ) & 1u) ? s_one : 0.0;
        acc_one += s_one;
        acc_d += s_d;
#pragma unroll
        for (int q = 0; q < 10; ++q)  // (bits 6 and up are the block number's)
            if (blk >> q & 1u) acc_high[q] += s_one;
    }
    QSV_PSTAMP(5);  // the blocks
    // The runs of one entry are added INSIDE each row of 16 lanes only (rotations by DPP: no LDS traffic); the four rows leave
    // as four partial sums which whoever adds the waves' partials adds as well (factor_sum_partials).  The butterflies over
    // all 64 lanes this replaces were 18 accumulators x 6 steps x 2 ds_bpermute per wave: the largest piece of a zero-key
    // side's Gram phase (5.4 k of 9.5 k cycles), every wave of the CU queueing at the same LDS crossbar.
    auto across = [&](double v) {
        if constexpr (NQ < 2) v += dpp_f64<0x121, 0xf>(v);   // row_ror:1
        if constexpr (NQ < 4) v += dpp_f64<0x122, 0xf>(v);   // row_ror:2
        if constexpr (NQ < 8) v += dpp_f64<0x124, 0xf>(v);   // row_ror:4
        if constexpr (NQ < 16) v += dpp_f64<0x128, 0xf>(v);  // row_ror:8
        return v;
    };
    acc_one = across(acc_one);
    acc_d = across(acc_d);
#pragma unroll
    for (int q = 0; q < 6; ++q) acc_low[q] = across(acc_low[q]);
#pragma unroll
    for (int q = 0; q < 10; ++q) acc_high[q] = across(acc_high[q]);
    QSV_PSTAMP(6);  // sums across lanes
    __syncthreads();  // (every wave of the workgroup is here: nobody reads a staging region any more)
    if ((lane & 15u) < NQ && out) {  // (out = null: a wave that only keeps the others company at the barrier)
        // slot row * 16 + pi of every weight: the row's partial sum of entry pi
        out[0 * 64 + lane] = acc_one;
        out[1 * 64 + lane] = acc_d;
#pragma unroll
        for (int q = 0; q < 6; ++q) out[(2 + q) * 64 + lane] = acc_low[q];
#pragma unroll
        for (int q = 0; q < 10; ++q) out[(8 + q) * 64 + lane] = acc_high[q];
    }
}

// Entry pi of weight w, added over the partial matrices [wave][weight][64] the bodies above leave in LDS: the waves in
// order, and for up to four product terms each wave's four row partials in order (eight terms: one value per wave).
__device__ __forceinline__ double factor_sum_partials(const double* partial, uint32_t n_waves, uint32_t w, uint32_t pi, uint32_t n_keys) {
    double v = 0.0;
    if (n_keys < 3) {
        for (uint32_t g = 0; g < n_waves; ++g)
#pragma unroll
            for (uint32_t row = 0; row < 4; ++row) v += partial[(size_t(g) * kFactorWeights + w) * 64 + row * 16 + pi];
    } else {
        for (uint32_t g = 0; g < n_waves; ++g) v += partial[(size_t(g) * kFactorWeights + w) * 64 + pi];
    }
    return v;
}

// EIGHT product terms (three cut keys): with a lane per entry every (entry, x) reads its two amplitudes from LDS -- 2 KiB per x
// and wave, and eight waves of a workgroup doing that was most of a three-key side's Gram phase (the longest workgroup of the
// benchmark's launch).  Here a lane owns TWO entries that share their amplitudes -- the real and the imaginary part of one
// pair (28 lanes), or two diagonal entries (4 lanes) -- so 32 lanes cover the 64 entries and the wave's halves take two x
// at a time: half the LDS traffic per x.  x = 64 blk + 2 i + half: the weight of bit 0 comes from the half, bits 1 .. 5 from the
// step number (three of them known inside an unrolled group of eight steps).  Worker `worker` of `n_workers` (a power of two) takes the CONTIGUOUS blocks [worker m, (worker + 1) m),
// m = n_blocks / n_workers, so only the low log2(m) <= HB bits of the block number need accumulators of their own (two entries
// per lane doubled the accumulators; ten more pairs of them for the block number did not fit the pass kernel's registers):
// the bits above are the worker's.  Sums ascend in x per lane, the two halves are added last, in every batch alike.
// (the contiguous blocks of a worker of the eight-term body: [begin, end))
__device__ __forceinline__ void factor_pairs_blocks(uint32_t bits, uint32_t worker, uint32_t n_workers, uint32_t* begin, uint32_t* end) {
    const uint32_t n_blocks = factor_block_count(bits);
    const uint32_t per_worker = n_blocks > n_workers ? n_blocks / n_workers : 1u;
    *begin = worker * per_worker;
    *end = worker < n_blocks ? *begin + per_worker : *begin;
    if (worker >= n_workers) *end = *begin;
}
// (LDS_ROWS: where row j of the side's state starts -- rows one amplitude apart by default; a half side's own and imported rows)
struct RowsOnePitch {
    template <typename real>
    __device__ __forceinline__ const cx<real>* operator()(const cx<real>* tab, uint32_t j, uint32_t row_pitch) const { return tab + j * row_pitch; }
};
template <typename real, int HB, bool LDS_ROWS, class RowOf = RowsOnePitch>
__device__ __forceinline__ void factor_side_body_pairs(const cx<real>* __restrict__ tab, uint32_t bits, uint32_t mask, const double* __restrict__ diag,
                                 uint32_t worker, uint32_t n_workers, cx<real>* stage, double* dstage, double* out,
                                 double (&dq)[kFactorDAheadPairs] QSV_PSTAMP_PARAMS, RowOf row_of = RowOf{}) {
    constexpr int J = 8;
    constexpr uint32_t PITCH = J + 1;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t half = lane >> 5, L = lane & 31u;
    const bool diagonal = L < uint32_t(J / 2);
    uint32_t ja = 2 * L, jb = 2 * L + 1, pi0 = 2 * L;
    if (!diagonal) {
        uint32_t part;
        pi0 = uint32_t(J) + 2 * (L - uint32_t(J / 2));
        split_entry_of<J>(pi0, &ja, &jb, &part);
    }
    const uint32_t n_local = bits < 6u ? 1u << bits : 64u, n_blocks = bits < 6u ? 1u : 1u << (bits - 6u);
    const uint32_t per_worker = n_blocks > n_workers ? n_blocks / n_workers : 1u;  // m
    const uint32_t lm = uint32_t(__builtin_ctz(per_worker));
    if (lm > uint32_t(HB)) __builtin_trap();  // (the callers' sides are smaller: kernels.hpp)
    uint32_t blk_begin, blk_end;
    factor_pairs_blocks(bits, worker, n_workers, &blk_begin, &blk_end);
    double acc_one[2] = {0.0, 0.0}, acc_d[2] = {0.0, 0.0}, acc_low[2][6], acc_high[2][HB];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int q = 0; q < 6; ++q) acc_low[e][q] = 0.0;
#pragma unroll
        for (int q = 0; q < HB; ++q) acc_high[e][q] = 0.0;
    }
    for (uint32_t blk = blk_begin; blk < blk_end; ++blk) {
        if constexpr (!LDS_ROWS) {
            const bool live = lane < n_local;
#pragma unroll
            for (int j = 0; j < J; ++j) {  // (a table of fewer than 64 x: the other lanes read lane 0's entry and stage zeros)
                const uint32_t off = ((uint32_t(j) << bits) + blk * 64u + (live ? lane : 0u)) * uint32_t(sizeof(cx<real>));
                cx<real> v = *reinterpret_cast<const cx<real>*>(reinterpret_cast<const char*>(tab) + off);
                v.re = live ? v.re : real(0);
                v.im = live ? v.im : real(0);
                stage[lane * PITCH + uint32_t(j)] = v;
            }
        }
        dstage[lane] = dq[0];
#pragma unroll
        for (int k = 0; k + 1 < kFactorDAheadPairs; ++k) dq[k] = dq[k + 1];
        dq[kFactorDAheadPairs - 1] = factor_d_of_block(diag, bits, mask, blk + uint32_t(kFactorDAheadPairs), blk_end);
        QSV_PSTAMP(4);  // (diagnostic build: a block's rows fetched and staged)
        double s_one[2] = {0.0, 0.0}, s_d[2] = {0.0, 0.0};
        // Four groups of eight steps: bits 1 .. 3 of x are known inside the unrolled group, bits 4 and 5 are the group's.  The
        // reads of step i + 1 are ISSUED before step i is added up (and kept there: the compiler's own order waited for every
        // pair of reads right after asking for it -- two waves per SIMD do not hide an LDS round trip per step).
        // (LDS_ROWS: `tab` IS the side's state in LDS, rows of 2^bits amplitudes one amplitude apart: read in place)
        const uint32_t row_pitch = (1u << bits) + 1u;
        constexpr uint32_t XS = LDS_ROWS ? 1u : PITCH;  // amplitudes from one x to the next
        const cx<real>* pa = LDS_ROWS ? row_of(tab, ja, row_pitch) + blk * 64u + half : stage + half * PITCH + ja;
        const cx<real>* pb = LDS_ROWS ? row_of(tab, jb, row_pitch) + blk * 64u + half : stage + half * PITCH + jb;
        const double* pd = dstage + half;
        real ar_next = pa[0].re, ai_next = pa[0].im, br_next = pb[0].re, bi_next = pb[0].im;
        double d_next = pd[0];
#ifdef QSV_ABL_PAIR_STEPS  // (measurement: the eight-term body without its steps)
#define QSV_PAIR_GROUPS 0
#else
#define QSV_PAIR_GROUPS 4
#endif
#pragma unroll 1
        for (uint32_t g = 0; g < QSV_PAIR_GROUPS; ++g) {
            double t_one[2] = {0.0, 0.0};
#pragma unroll
            for (uint32_t ii = 0; ii < 8; ++ii) {
                const double ar = double(ar_next), ai = double(ai_next), br = double(br_next), bi = double(bi_next);
                const double d = d_next;
                {
                    // (the step after the block's last one reads the last rows again: never used)
                    const uint32_t nxt = 8 * g + ii + 1 < 32 ? 8 * g + ii + 1 : 31;
                    const cx<real> an = pa[2 * nxt * XS], bn = pb[2 * nxt * XS];
                    ar_next = an.re;
                    ai_next = an.im;
                    br_next = bn.re;
                    bi_next = bn.im;
                    d_next = pd[2 * nxt];
                }
                if constexpr (sizeof(real) == 8) __builtin_amdgcn_sched_barrier(0);  // (single precision: the fence costs the kernel a stack slot)
                // a pair: Re, Im of a conj(b) as split_entry_value has them; two diagonal entries: |a|^2, |b|^2
#ifdef QSV_ABL_PAIR_NODIAG  // (measurement: the steps without the diagonal entries' second form and the selects)
                const double p0 = fma(ar, br, ai * bi);
                const double p1 = fma(ai, br, -ar * bi);
#else
                const double p0 = diagonal ? fma(ar, ar, ai * ai) : fma(ar, br, ai * bi);
                const double p1 = diagonal ? fma(br, br, bi * bi) : fma(ai, br, -ar * bi);
#endif
                t_one[0] += p0;
                t_one[1] += p1;
                s_d[0] = fma(d, p0, s_d[0]);
                s_d[1] = fma(d, p1, s_d[1]);
#pragma unroll
                for (uint32_t q = 1; q < 4; ++q)
                    if (ii >> (q - 1) & 1u) {
                        acc_low[0][q] += p0;
                        acc_low[1][q] += p1;
                    }
                if constexpr (sizeof(real) == 8) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                acc_low[e][4] += (g & 1u) ? t_one[e] : 0.0;
                acc_low[e][5] += (g & 2u) ? t_one[e] : 0.0;
                s_one[e] += t_one[e];
            }
        }
        QSV_PSTAMP(7);  // (diagnostic build: a block's steps)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            acc_low[e][0] += half ? s_one[e] : 0.0;
            acc_one[e] += s_one[e];
            acc_d[e] += s_d[e];
#pragma unroll
            for (int q = 0; q < HB; ++q)
                if (blk >> q & 1u) acc_high[e][q] += s_one[e];
        }
    }
    QSV_PSTAMP(5);  // the blocks' sums
    // (the two halves: v_permlane32_swap of the value with a copy of itself leaves (low, low) and (high, high) -- no LDS)
    auto across = [&](double v) {
        uint32_t lo0 = uint32_t(__double2loint(v)), lo1 = lo0, hi0 = uint32_t(__double2hiint(v)), hi1 = hi0;
        swap_words<5>(lo0, lo1);
        swap_words<5>(hi0, hi1);
        return __hiloint2double(int(hi0), int(lo0)) + __hiloint2double(int(hi1), int(lo1));
    };
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        acc_one[e] = across(acc_one[e]);
        acc_d[e] = across(acc_d[e]);
#pragma unroll
        for (int q = 0; q < 6; ++q) acc_low[e][q] = across(acc_low[e][q]);
#pragma unroll
        for (int q = 0; q < HB; ++q) acc_high[e][q] = across(acc_high[e][q]);
    }
    QSV_PSTAMP(6);  // the halves added
    __syncthreads();  // (every wave of the workgroup is here: nobody reads a staging region any more)
    if (lane < 32 && out) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t pi = pi0 + uint32_t(e);
            out[0 * 64 + pi] = acc_one[e];
            out[1 * 64 + pi] = acc_d[e];
#pragma unroll
            for (int q = 0; q < 6; ++q) out[(2 + q) * 64 + pi] = acc_low[e][q];
#pragma unroll
            for (uint32_t q = 0; q < 10; ++q) {  // (block-number bit q: one of this worker's own, or the same for all its blocks)
                double v = 0.0;
                if (q < uint32_t(HB) && q < lm) v = acc_high[e][q < uint32_t(HB) ? q : 0];
                if (q >= lm && (worker >> (q - lm) & 1u)) v = acc_one[e];
                out[(8 + q) * 64 + pi] = v;
            }
        }
    }
}

// sum_{j'j} A[j'j] B[j'j] for two Hermitian matrices in the entry representation (split_entry_of): the diagonal entries,
// and for every pair twice the real part of the product
// (unrolled per J: the reads of a table row go out together instead of one dependent LDS round trip per entry -- at eight
// terms the loop form was 36 round trips, most of the three-key combination; the order of the additions is the loop's)
template <uint32_t J>
__device__ __forceinline__ double factor_pairing_unrolled(const double* a, const double* b) {
    constexpr uint32_t NQ = J * J;
    double t = 0.0;
#pragma unroll
    for (uint32_t j = 0; j < J; ++j) t = fma(a[j], b[j], t);
#pragma unroll
    for (uint32_t e = J; e < NQ; e += 2) t += 2.0 * fma(a[e], b[e], -a[e + 1] * b[e + 1]);
    return t;
}
__device__ __forceinline__ double factor_pairing(const double* a, const double* b, uint32_t n_keys) {
    switch (n_keys) {
        case 0: return factor_pairing_unrolled<1>(a, b);
        case 1: return factor_pairing_unrolled<2>(a, b);
        case 2: return factor_pairing_unrolled<4>(a, b);
        default: return factor_pairing_unrolled<8>(a, b);
    }
}

constexpr unsigned kFactorParts = 8;
constexpr uint32_t kFactorSlices = kFactorParts / 2;

// ---- the same for sixteen and thirty-two product terms (four and five cut keys) ------------------------------------------
// J^2 = 256 / 1024 entries no longer fit the lanes of a wave: here a workgroup's 256 threads each own ONE entry (of a
// group of 256) and every wave walks ALL rows of the workgroup's blocks, which the workgroup stages in LDS together
// (J x 64 amplitudes, x-major with one amplitude of padding per x, and the block's 64 values of D).  Grid: entry groups x
// kFactorSlices (a slice takes blocks slice, slice + 4, ..) x the two sides, per evaluation; a slice's sums leave as
// partial sums [side][slice][weight][entry], added in slice order by factor_combine_big_kernel.  Bound by the LDS reads
// (two 16-byte reads per entry and row): about 1.7 us per block and workgroup.
constexpr uint32_t kFactorBigEntries = 1024;  // J = 32
constexpr uint32_t kFactorBigSlices = 8;      // (a slice takes blocks slice, slice + 8, ..: at most four of a 2^11-row table)
// per side-table slot: the slices' partial sums [side][slice][weight][entry], then the finished sums [side][weight][entry]
constexpr size_t kFactorBigFinal = size_t(2) * kFactorBigSlices * kFactorWeights * kFactorBigEntries;
constexpr size_t factor_big_slot_doubles_c() { return kFactorBigFinal + size_t(2) * kFactorWeights * kFactorBigEntries; }
size_t factor_big_slot_doubles() { return factor_big_slot_doubles_c(); }
// ... and one counter per (side, entry group): the workgroup of a (side, group) that finishes LAST adds the slices' partial sums
// (in slice order: which one is last does not enter any sum); a counter only ever grows, by kFactorBigSlices per evaluation
constexpr uint32_t kFactorBigCounters = 8;
size_t factor_big_slot_counters() { return kFactorBigCounters; }

// blockIdx.x = side + 2 * (slice + kFactorBigSlices * entry group).  The rows of the NEXT block are fetched (into registers)
// before this block's sums are formed: unpipelined, a block cost one memory latency plus its arithmetic, 6 - 7 us of which
// 1.5 were arithmetic -- few workgroups, one to a CU, nobody else to hide it.
template <typename real, int J>
__device__ __forceinline__ void factor_moments_big_body(const EvalDesc& ev, const uint32_t* __restrict__ sp,
                                                        const cx<real>* __restrict__ sides, uint64_t side_stride,
                                                        const double* __restrict__ diag, const double* __restrict__ side_diag,
                                                        double* __restrict__ scratch, uint32_t* __restrict__ counters, cx<real>* stage, double* dstage) {
    constexpr uint32_t NQ = J * J, PITCH = J + 1, PER = uint32_t(J) * 64u / 256u;  // amplitudes a thread stages per block
    const bool swap = sp[3] & 1u;
    const uint32_t side = blockIdx.x & 1u, slice = (blockIdx.x >> 1) & (kFactorBigSlices - 1), group = blockIdx.x / (2 * kFactorBigSlices);
    if (group * 256u >= NQ) return;  // (uniform: before any barrier)
    const uint32_t bits = sp[1 + side];
    const uint32_t own_d = sp[kSplitSideDiag + side];  // (the side's own table of D, read in place)
    const bool in_place = own_d != kNoSideDiag;
    const uint32_t mask = sp[kSplitMaskX + side];
    diag = in_place ? side_diag + own_d : diag;
    const uint32_t tid = threadIdx.x;
    const cx<real>* ta = sides + uint64_t(ev.state_slot) * side_stride;
    const cx<real>* tab = ta + ((side == 0) == swap ? side_stride >> 1 : 0);  // X is side B's half when swapped
    const uint32_t pi = group * 256u + tid;
    uint32_t ja, jb, part;
    split_entry_of<J>(pi, &ja, &jb, &part);
    const uint32_t n_local = bits < 6u ? 1u << bits : 64u, n_blocks = bits < 6u ? 1u : 1u << (bits - 6u);
    double acc_one = 0.0, acc_d = 0.0, acc_low[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, acc_high[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) acc_high[q] = 0.0;
    cx<real> next[PER];
    double d_next = 0.0;
    auto fetch = [&](uint32_t blk) {
#pragma unroll
        for (uint32_t i = 0; i < PER; ++i) {
            const uint32_t idx = tid + i * 256u, j = idx >> 6, xl = idx & 63u;
            next[i] = cx<real>{real(0), real(0)};
            if (blk < n_blocks && xl < n_local) next[i] = tab[(size_t(j) << bits) + size_t(blk) * 64 + xl];
        }
        d_next = (tid < 64 && blk < n_blocks && tid < n_local) ? diag[in_place ? blk * 64u + tid : deposit_bits(blk * 64u + tid, mask)] : 0.0;
    };
    fetch(slice);
    for (uint32_t blk = slice; blk < n_blocks; blk += kFactorBigSlices) {
        __syncthreads();  // (the block before is no longer read)
#pragma unroll
        for (uint32_t i = 0; i < PER; ++i) {
            const uint32_t idx = tid + i * 256u, j = idx >> 6, xl = idx & 63u;
            stage[xl * PITCH + j] = next[i];
        }
        if (tid < 64) dstage[tid] = d_next;
        fetch(blk + kFactorBigSlices);
        __syncthreads();
        double s_one = 0.0, s_d = 0.0;
#pragma unroll
        for (uint32_t i = 0; i < 64; ++i) {
            const cx<real> a = stage[i * PITCH + ja], b = stage[i * PITCH + jb];
            const double p = split_entry_value(part, double(a.re), double(a.im), double(b.re), double(b.im));
            s_one += p;
            s_d = fma(dstage[i], p, s_d);
#pragma unroll
            for (uint32_t q = 0; q < 6; ++q)  // (bits of i: known when the loop is unrolled)
                if (i >> q & 1u) acc_low[q] += p;
        }
        acc_one += s_one;
        acc_d += s_d;
#pragma unroll
        for (int q = 0; q < 10; ++q)  // (bits 6 and up are the block number's)
            if (blk >> q & 1u) acc_high[q] += s_one;
    }
    // The slice's sums leave by write-through (agent-scope relaxed) stores, every storing wave drains them, workgroup barrier,
    // one lane adds to the counter of this (side, entry group) at agent scope; the workgroup whose add completes the
    // evaluation's kFactorBigSlices is the last: it reads all slices' sums back (agent-scope loads: the other workgroups may
    // sit on other XCDs, whose L2s are not coherent -- the hand-off of fused_factor_tail) and adds them in slice order.
    double* slot = scratch + size_t(ev.state_slot) * factor_big_slot_doubles_c();
    double* mine = slot + (size_t(side) * kFactorBigSlices + slice) * kFactorWeights * kFactorBigEntries;
    auto put = [&](uint32_t w, double v) { __hip_atomic_store(mine + size_t(w) * kFactorBigEntries + pi, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    put(0, acc_one);
    put(1, acc_d);
#pragma unroll
    for (int q = 0; q < 6; ++q)
        if (uint32_t(q) < bits) put(2 + q, acc_low[q]);
#pragma unroll
    for (int q = 0; q < 10; ++q)
        if (uint32_t(6 + q) < bits) put(8 + q, acc_high[q]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t* flag = reinterpret_cast<uint32_t*>(dstage);
    if (tid == 0) {
        const uint32_t before = __hip_atomic_fetch_add(counters + size_t(ev.state_slot) * kFactorBigCounters + side * 4u + group, 1u,
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = (before & (kFactorBigSlices - 1)) == kFactorBigSlices - 1 ? 1u : 0u;  // (the add has returned: its value is used)
    }
    __syncthreads();
    if (!*flag) return;
    double* fin = slot + kFactorBigFinal + size_t(side) * kFactorWeights * kFactorBigEntries;
#pragma unroll
    for (uint32_t w = 0; w < kFactorWeights; ++w) {
        if (w >= 2 + bits) break;
        double v = 0.0;
#pragma unroll
        for (uint32_t g = 0; g < kFactorBigSlices; ++g)
            v += __hip_atomic_load(slot + ((size_t(side) * kFactorBigSlices + g) * kFactorWeights + w) * kFactorBigEntries + pi,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fin[size_t(w) * kFactorBigEntries + pi] = v;
    }
}

// One workgroup of 1024 threads per evaluation, a thread per entry (J = 16: the first 256); per entry the slices' partial sums
// in slice order -- every load of a thread independent of the others: they are all in flight together --, then
//   sigma_e [ 4 sum_{ab} J_ab FX_{2+a}[e] FY_{2+b}[e] + FX_D[e] FY_1[e] + FX_1[e] FY_D[e] - D(0,0) FX_1[e] FY_1[e] ]
// (sigma = 1 / 2 / -2 for a diagonal entry / the real / the imaginary part of a pair: factor_pairing entry by entry), and the
// entries' terms are added in a fixed order (waves in order after the fixed shuffle tree of each).
template <int J>
__device__ __forceinline__ void factor_combine_big_body(const EvalDesc& ev, const uint32_t* __restrict__ sp, const double* __restrict__ scratch,
                                                        const double* __restrict__ quad, uint32_t n_qubits, const double* __restrict__ diag,
                                                        double* __restrict__ result_out, double* coupling, double* red) {
    constexpr uint32_t NQ = J * J;
    const uint32_t bx = sp[1], by = sp[2];
    const uint32_t masks[2] = {sp[kSplitMaskX], sp[kSplitMaskY]};
    const uint32_t tid = threadIdx.x;
    if (tid < 256) coupling[tid] = 0.0;
    __syncthreads();
    if (tid < bx * by) {
        const uint32_t a = tid / by, b = tid % by;
        uint32_t qa = 0, qb = 0;
        for (uint32_t m = masks[0], k = 0; m; m &= m - 1, ++k)
            if (k == a) qa = uint32_t(__builtin_ctz(m));
        for (uint32_t m = masks[1], k = 0; m; m &= m - 1, ++k)
            if (k == b) qb = uint32_t(__builtin_ctz(m));
        coupling[a * 16 + b] = quad[qa * n_qubits + qb];
    }
    const double d00 = diag[0];
    const double* fin = scratch + size_t(ev.state_slot) * factor_big_slot_doubles_c() + kFactorBigFinal;  // [side][weight][entry]
    const uint32_t pi = tid;
    double fx[kFactorWeights], fy[kFactorWeights];  // (fixed trip counts: the arrays stay in registers)
#pragma unroll
    for (uint32_t w = 0; w < kFactorWeights; ++w) {
        fx[w] = (pi < NQ && w < 2 + bx) ? fin[(size_t(0) * kFactorWeights + w) * kFactorBigEntries + pi] : 0.0;
        fy[w] = (pi < NQ && w < 2 + by) ? fin[(size_t(1) * kFactorWeights + w) * kFactorBigEntries + pi] : 0.0;
    }
    __syncthreads();  // (the couplings)
    double acc = 0.0;
    if (pi < NQ) {
        uint32_t ja, jb, part;
        split_entry_of<J>(pi, &ja, &jb, &part);
        const double sigma = part == 0 ? 1.0 : part == 1 ? 2.0 : -2.0;
        double t = fma(fx[1], fy[0], fx[0] * fy[1]) - d00 * fx[0] * fy[0];
#pragma unroll
        for (uint32_t a = 0; a < 16; ++a) {
            double row = 0.0;
#pragma unroll
            for (uint32_t b = 0; b < 16; ++b) row = fma(coupling[a * 16 + b], fy[2 + b], row);  // (zero beyond bx x by)
            t = fma(4.0 * fx[2 + a], row, t);
        }
        acc = sigma * t;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((tid & 63u) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double total = 0.0;
        for (uint32_t w = 0; w < NQ / 64u; ++w) total += red[w];
        result_out[ev.out_index] = total;
    }
}

// Two launches.  factor_moments_kernel: kFactorParts workgroups of four waves per evaluation, the even ones on side X,
// the odd ones on side Y; wave (slice, w) takes blocks 4 slice + w, + 16, ..; a workgroup's four partial matrices are added
// in LDS (wave order) and leave as ONE partial per weight and entry.  factor_combine_kernel: one workgroup per evaluation
// adds the slices' partials in order, then one thread per pair (a, b) of qubits across the cut -- and three for the
// separable parts -- forms its term, and the terms are added in a fixed order.
static_assert(factor_slot_doubles() == size_t(2) * kFactorSlices * kFactorWeights * 64, "scratch of one evaluation");

template <typename real>
__global__ void __launch_bounds__(256, 2) factor_moments_kernel(const uint32_t* __restrict__ plan_arena, const EvalDesc* __restrict__ evals,
                                                                const cx<real>* __restrict__ sides, uint64_t side_stride,
                                                                const double* __restrict__ diag, double* __restrict__ scratch,
                                                                double* __restrict__ scratch_big, uint32_t* __restrict__ big_counters,
                                                                const double* __restrict__ side_diag) {
    constexpr uint32_t kWaves = 4;
    // the waves' staging regions (9 x 64 amplitudes each) and, afterwards, their partial matrices (18 x 64 doubles each)
    __shared__ __align__(16) unsigned char raw[kWaves * kFactorWeights * 64 * sizeof(double)];
    static_assert(sizeof(raw) >= kWaves * 9 * 64 * sizeof(cx<real>), "staging regions");
    static_assert(sizeof(raw) >= 64 * 33 * sizeof(cx<real>), "block of 32 product terms");
    __shared__ double dstage[kWaves * 64];
    const EvalDesc ev = evals[blockIdx.y];
    if (!(ev.flags & kEvalSide)) return;
    const uint32_t* sp = plan_arena + ev.split_base;
    const uint32_t n_keys = sp[0], NQ = 1u << (2 * n_keys);
    if (n_keys > 3) {  // sixteen / thirty-two product terms: a thread per entry (grid: up to four entry groups)
        if (n_keys == 4)
            factor_moments_big_body<real, 16>(ev, sp, sides, side_stride, diag, side_diag, scratch_big, big_counters, reinterpret_cast<cx<real>*>(raw), dstage);
        else
            factor_moments_big_body<real, 32>(ev, sp, sides, side_stride, diag, side_diag, scratch_big, big_counters, reinterpret_cast<cx<real>*>(raw), dstage);
        return;
    }
    if (blockIdx.x >= kFactorParts) return;  // (a grid widened for an evaluation of 32 terms)
    const bool swap = sp[3] & 1u;
    const uint32_t side = blockIdx.x & 1u, slice = blockIdx.x >> 1;
    const uint32_t bits = sp[1 + side];
    // (the side's own table of D where the block names one, read in place: kernels.hpp kSplitSideDiag)
    const uint32_t own_d = sp[kSplitSideDiag + side];
    const uint32_t mask = own_d != kNoSideDiag ? (1u << bits) - 1u : sp[kSplitMaskX + side];
    diag = own_d != kNoSideDiag ? side_diag + own_d : diag;
    const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const cx<real>* ta = sides + uint64_t(ev.state_slot) * side_stride;
    const cx<real>* tab = ta + ((side == 0) == swap ? side_stride >> 1 : 0);  // X is side B's half when swapped
    cx<real>* stage = reinterpret_cast<cx<real>*>(raw) + size_t(wave) * 9 * 64;
    double* partial = reinterpret_cast<double*>(raw);  // [wave][weight][64]
    double* out = partial + size_t(wave) * kFactorWeights * 64;
    const uint32_t first = slice * kWaves + wave, step = kFactorSlices * kWaves;
    double dq[kFactorDAhead], dq_pairs[kFactorDAheadPairs];
    if (n_keys < 3) {
        factor_prefetch_d(dq, diag, bits, mask, first, step, factor_block_count(bits));
    } else {
        uint32_t b0, b1;
        factor_pairs_blocks(bits, first, step, &b0, &b1);
        factor_prefetch_d(dq_pairs, diag, bits, mask, b0, 1u, b1);
    }
    if (n_keys == 0)
        factor_side_body<real, 1>(tab, bits, mask, diag, first, step, stage, dstage + wave * 64, out, dq);
    else if (n_keys == 1)
        factor_side_body<real, 2>(tab, bits, mask, diag, first, step, stage, dstage + wave * 64, out, dq);
    else if (n_keys == 2)
        factor_side_body<real, 4>(tab, bits, mask, diag, first, step, stage, dstage + wave * 64, out, dq);
    else
        factor_side_body_pairs<real, 6, false>(tab, bits, mask, diag, first, step, stage, dstage + wave * 64, out, dq_pairs);
    __syncthreads();
    double* mine = scratch + size_t(ev.state_slot) * factor_slot_doubles() +
                   (size_t(side) * kFactorSlices + slice) * kFactorWeights * 64;
    for (uint32_t idx = tid; idx < (2u + bits) * NQ; idx += blockDim.x) {
        const uint32_t w = idx / NQ, pi = idx % NQ;
        mine[w * 64 + pi] = factor_sum_partials(partial, kWaves, w, pi, n_keys);
    }
}

// (256 threads; 1024 when the launch holds evaluations of sixteen or thirty-two product terms, whose combination takes a
// thread per entry -- the others' code does not care: its loops stride by the block size, its terms sit in the first 172 threads)
__global__ void __launch_bounds__(1024) factor_combine_kernel(const uint32_t* __restrict__ plan_arena, const EvalDesc* __restrict__ evals,
                                                              const double* __restrict__ scratch, const double* __restrict__ scratch_big,
                                                              const double* __restrict__ quad, uint32_t n_qubits,
                                                              const double* __restrict__ diag, double* __restrict__ result_out) {
    __shared__ double gram[2][kFactorWeights * kFactorPitch];
    __shared__ double red[16];
    const EvalDesc ev = evals[blockIdx.x];
    if (!(ev.flags & kEvalSide)) return;
    const uint32_t* sp = plan_arena + ev.split_base;
    const uint32_t n_keys = sp[0], NQ = 1u << (2 * n_keys);
    if (n_keys > 3) {
        static_assert(sizeof(gram) >= 16 * 16 * sizeof(double), "couplings of the cut");
        if (n_keys == 4)
            factor_combine_big_body<16>(ev, sp, scratch_big, quad, n_qubits, diag, result_out, &gram[0][0], red);
        else
            factor_combine_big_body<32>(ev, sp, scratch_big, quad, n_qubits, diag, result_out, &gram[0][0], red);
        return;
    }
    const uint32_t bits[2] = {sp[1], sp[2]}, mask[2] = {sp[kSplitMaskX], sp[kSplitMaskY]};
    const double* mine = scratch + size_t(ev.state_slot) * factor_slot_doubles();
    const uint32_t tid = threadIdx.x;
    // (the coupling of this thread's pair of qubits: its load goes out together with the partials')
    const uint32_t bx = bits[0], by = bits[1];
    double coupling = 0.0;
    uint32_t a = 0, b = 0;
    if (tid < bx * by) {
        a = tid / by;
        b = tid % by;
        uint32_t qa = 0, qb = 0;  // the a-th qubit of side X, the b-th of side Y
        for (uint32_t m = mask[0], k = 0; m; m &= m - 1, ++k)
            if (k == a) qa = uint32_t(__builtin_ctz(m));
        for (uint32_t m = mask[1], k = 0; m; m &= m - 1, ++k)
            if (k == b) qb = uint32_t(__builtin_ctz(m));
        coupling = quad[qa * n_qubits + qb];
    }
    const double d00 = diag[0];
    for (uint32_t side = 0; side < 2; ++side)
        for (uint32_t idx = tid; idx < (2u + bits[side]) * NQ; idx += blockDim.x) {
            const uint32_t w = idx / NQ, pi = idx % NQ;
            double v = 0.0;
#pragma unroll
            for (uint32_t g = 0; g < kFactorSlices; ++g) v += mine[(size_t(side) * kFactorSlices + g) * kFactorWeights * 64 + w * 64 + pi];
            gram[side][w * kFactorPitch + pi] = v;
        }
    __syncthreads();
    double v = 0.0;
    if (tid < bx * by) {
        if (coupling != 0.0)
            v = 4.0 * coupling * factor_pairing(gram[0] + (2 + a) * kFactorPitch, gram[1] + (2 + b) * kFactorPitch, n_keys);
    } else if (tid == bx * by) {
        v = factor_pairing(gram[0] + kFactorPitch, gram[1], n_keys);  // D(x, 0)
    } else if (tid == bx * by + 1) {
        v = factor_pairing(gram[0], gram[1] + kFactorPitch, n_keys);  // D(0, y)
    } else if (tid == bx * by + 2) {
        v = -d00 * factor_pairing(gram[0], gram[1], n_keys);          // - D(0, 0) <psi|psi>
    }
    const double total = block_sum_256(v, red);
    if (tid == 0) result_out[ev.out_index] = total;
}

// The two launches above as the tail of the kernel that ran the virtual circuits (pass_kernel, kModeFusedFactor): the side's
// final state has just been stored to its half of the slot; four waves -- eight for a virtual circuit whose own tile needs
// 512 threads; never a function of the launch's block size: the assignment of table blocks to waves, and with it the order
// of every sum, must not depend on the batch -- form the weighted Gram matrices of this side, the workgroup leaves them in the slot's scratch, and the workgroup of an evaluation that gets there second
// combines the two sets.  The hand-off between the two workgroups (which may sit on different XCDs, whose L2s are not
// coherent): the Gram matrices leave by write-through (agent-scope relaxed) stores, every storing wave drains them, workgroup
// barrier, one lane adds to the slot's counter at agent scope; the workgroup whose add returns an odd value is the second:
// workgroup barrier, then agent-scope loads of both sets (MI355X_MICROARCH.md, hand-offs with sc1 stores and loads in place
// of a release / acquire pair, first row: one unsharded counter, the last adder told by its add's return value).  The order in
// which the two sides finish does not enter any sum.
template <typename real>
__device__ __forceinline__ void fused_factor_tail(const uint32_t* __restrict__ plan_arena, const EvalDesc& ev,
                                                  const cx<real>* __restrict__ slot_tables, uint64_t side_stride,
                                                  const double* __restrict__ diag, const PassScalars& a, unsigned char* lds,
                                                  uint32_t gram_waves, uint32_t lds_table, uint32_t halves_tile QSV_PSTAMP_PARAMS_DEF) {
    constexpr uint32_t kMaxWaves = 8;
    // (lds_table: 0 the side's state is in its slot; 1 in LDS behind the tail's scratch, laid out like the slot; 2 in LDS from
    // offset 0 as padded rows, kernels.hpp)
    const bool table_in_lds = lds_table == 1u, table_lds_rows = lds_table == 2u;
    const bool halves = halves_tile != 0xffffffffu;  // (kEvalHalves: this workgroup has the rows whose third key bit is halves_tile)
    const uint32_t hh = halves ? halves_tile : 0u;
    const uint32_t kWaves = gram_waves;
    const uint32_t* sp = plan_arena + ev.split_base;
    const uint32_t n_keys = sp[0], NQ = 1u << (2 * n_keys);
    const bool swap = sp[3] & 1u, is_b = ev.flags & kEvalSideB;
    const uint32_t xy = (is_b == swap) ? 0u : 1u;  // this side's name in the contraction's terms (X is B's half when swapped)
    const uint32_t bits = sp[1 + xy];
    // (the side's own table of D where the block names one -- entry x, one run -- else D itself, entry x deposited in the side's qubits)
    const uint32_t own_d = sp[kSplitSideDiag + xy];
    const uint32_t mask = own_d != kNoSideDiag ? (1u << bits) - 1u : sp[kSplitMaskX + xy];
    const double* whole_diag = diag;  // (the combination's D(0, 0))
    diag = own_d != kNoSideDiag ? a.side_diag + own_d : diag;
    const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (the side's state: in its half of the slot -- or still in LDS, where the pass left it for this tail alone)
    const cx<real>* tab = table_in_lds ? reinterpret_cast<const cx<real>*>(lds + kFusedLdsTableOffset) : slot_tables + (is_b ? side_stride >> 1 : 0);
    // (the values of D this wave's blocks will want: asked for before the wait below, whose stores they do not depend on)
    double dq[kFactorDAhead], dq_pairs[kFactorDAheadPairs];
    {
        const uint32_t worker = wave < kWaves ? wave : 0xffffffu;
        if (n_keys < 3) {
            if (halves)  // (a half side: the blocks of its half of x)
                factor_prefetch_d(dq, diag, bits, mask, wave < kWaves ? (factor_block_count(bits) >> 1) * hh + wave : 0xffffffu, kWaves, (factor_block_count(bits) >> 1) * (hh + 1u));
            else
                factor_prefetch_d(dq, diag, bits, mask, worker, kWaves, factor_block_count(bits));
        } else {
            uint32_t b0, b1;
            if (halves)  // (a half side: this workgroup's waves are workers 8 h .. 8 h + 7 of the side's sixteen)
                factor_pairs_blocks(bits, wave < kWaves ? kWaves * hh + wave : 0xffffffu, 2u * kWaves, &b0, &b1);
            else
                factor_pairs_blocks(bits, worker, kWaves, &b0, &b1);
            factor_prefetch_d(dq_pairs, diag, bits, mask, b0, 1u, b1);
        }
    }
    // the side's state was stored by this workgroup: its waves' stores have to be in L2 before anybody reads them back (a state
    // left in LDS has no stores to wait for -- and the wait would be for the values of D just asked for)
    if (lds_table == 0u || halves) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    QSV_PSTAMP(0);  // the side's stores drained
    if (halves) {
        // The two halves of a side trade half rows through the side's half of the slot (rows of 2^10 as they would lie there):
        // mine for the OTHER half of x went out with the pass's own stores (write-through, drained above, by every wave, before the
        // barrier), one lane adds to the exchange counter (it grows by two per launch: the first to add waits for the next even
        // value, bounded), barrier; the partner's rows for MY half of x come in by agent-scope loads (the partner may sit on
        // another XCD) and lie behind mine.
        // (three keys: four rows of 2^10 a workgroup, the partner's half rows behind mine, one amplitude apart; fewer: 2^(keys - 1)
        // rows of 2^bits as they lie in the tile, the partner's half of each row where my own other half was -- that went out)
        const uint32_t own_rows = (1u << n_keys) >> 1, row_bits = bits, half_x = 1u << (bits - 1u);
        const cx<real>* gtab = slot_tables + (is_b ? side_stride >> 1 : 0);
        if (tid == 0) {
            uint32_t* exchange = a.factor_counters + size_t(kFactorCountersPerSlot) * ev.state_slot + 1u + (is_b ? 1u : 0u);
            const uint32_t before = __hip_atomic_fetch_add(exchange, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t want = (before | 1u) + 1u;
            for (uint32_t spins = 0; spins < (1u << 18); ++spins) {  // (bounded: a partner that never comes costs a wrong value, not a hang)
                if (int32_t(__hip_atomic_load(exchange, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        cx<real>* imported = reinterpret_cast<cx<real>*>(lds + kFusedHalvesImport);
        cx<real>* table = reinterpret_cast<cx<real>*>(lds + kFusedLdsTableOffset);
        // (all of a thread's loads first -- they are trips to memory --, then its LDS writes; 2^11 amplitudes whatever the keys)
        constexpr uint32_t kCount = 1u << (kFusedLdsRowsBits - 2), kMost = kCount / 256u;  // (a workgroup of at least four waves)
        double got[kMost][2];
#pragma unroll
        for (uint32_t it = 0; it < kMost; ++it) {
            const uint32_t i = tid + it * blockDim.x, r = i >> (row_bits - 1u), xl = i & (half_x - 1u);
            got[it][0] = got[it][1] = 0.0;
            if (i < kCount) {
                const double* src = reinterpret_cast<const double*>(gtab + (size_t(own_rows * (1u - hh) + r) << row_bits) + hh * half_x + xl);
                got[it][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                got[it][1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (uint32_t it = 0; it < kMost; ++it) {
            const uint32_t i = tid + it * blockDim.x, r = i >> (row_bits - 1u), xl = i & (half_x - 1u);
            if (i < kCount) {
                const cx<real> v{real(got[it][0]), real(got[it][1])};
                if (table_lds_rows)
                    imported[r * uint32_t(kFusedHalvesImportPitch) + xl] = v;
                else
                    table[(size_t(r) << row_bits) + (1u - hh) * half_x + xl] = v;
            }
        }
        __syncthreads();
    }
    double* partial = reinterpret_cast<double*>(lds);                              // [wave][weight][64]
    // [wave][64]; (rows in LDS: behind them -- the partial matrices lie over the rows, and are written when nobody reads those any more)
    double* dstage_all = table_lds_rows ? reinterpret_cast<double*>(lds + kFusedLdsRowsDstage) : partial + size_t(kMaxWaves) * kFactorWeights * 64;
    uint32_t* flag = reinterpret_cast<uint32_t*>(dstage_all + size_t(kMaxWaves) * 64);
    {
        // (waves beyond the fourth take no blocks -- first block past the end -- but join the body's barrier)
        const uint32_t w = wave < kWaves ? wave : 0u;
        cx<real>* stage = reinterpret_cast<cx<real>*>(lds) + size_t(w) * 9 * 64;
        double* out = partial + size_t(w) * kFactorWeights * 64;
        const uint32_t first = wave < kWaves ? wave : 0xffffffu, step = kWaves;
        double* sink = wave < kWaves ? out : nullptr;
#ifdef QSV_ABL_TAIL_GRAM  // (measurement: the tail without its Gram sums)
        if (sink)
            for (uint32_t i = tid & 63u; i < kFactorWeights * 64; i += 64) sink[i] = 0.0;
        __syncthreads();
        if (n_keys > 99)
#else
        // (a half side of one or two keys: row j = (its last key bit, the others): mine where they lie in the tile, the partner's in my
        // rows' other halves of x, so that entry x of each lies at x; blocks of my half of x only)
        HalfRows half_rows;
        uint32_t half_first = first, half_end = 0xffffffffu;
        if (halves && n_keys < 3) {
            const uint32_t own_rows = (1u << n_keys) >> 1, half_x = 1u << (bits - 1u), blocks = factor_block_count(bits) >> 1;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {  // (unrolled: the array stays in registers)
                const int32_t local = int32_t(((j & (own_rows - 1u)) << bits) * uint32_t(sizeof(cx<real>)));
                const int32_t shift = int32_t(half_x * uint32_t(sizeof(cx<real>)));
                half_rows.at[j] = (j >> (n_keys - 1u)) == hh ? local : local + (hh ? -shift : shift);
            }
            half_rows.on = true;
            half_first = wave < kWaves ? blocks * hh + wave : 0xffffffu;
            half_end = blocks * (hh + 1u);
        }
        if (n_keys == 0)
#endif
            factor_side_body<real, 1, true>(tab, bits, mask, diag, first, step, stage, dstage_all + w * 64, sink, dq QSV_PSTAMP_ARGS);
#ifndef QSV_ABL_TAIL_GRAM
        else if (n_keys == 1)
            factor_side_body<real, 2, true>(tab, bits, mask, diag, half_first, step, stage, dstage_all + w * 64, sink, dq QSV_PSTAMP_ARGS,
                                            half_rows, half_end);
        else if (n_keys == 2)
            factor_side_body<real, 4, false>(tab, bits, mask, diag, half_first, step, stage, dstage_all + w * 64, sink, dq QSV_PSTAMP_ARGS,
                                             half_rows, half_end);
        else if (halves) {
            // rows 4 h .. 4 h + 3 are mine (pitch kFusedLdsRowPitch, every x), the others the partner's (this half of x only, so that
            // its x = 2^9 h lies at the row's start); blocks 8 h .. 8 h + 7 of the side's sixteen, one per wave
            const uint32_t mine_first = 4u * hh, x0 = hh << (kFusedLdsRowsBits - kFusedLdsRowsKeys - 1);
            auto row_of = [=](const cx<real>* base, uint32_t j, uint32_t) -> const cx<real>* {
                return (j >> 2) == (mine_first >> 2) ? base + (j & 3u) * uint32_t(kFusedLdsRowPitch)
                                                    : base + uint32_t(kFusedHalvesImport / sizeof(cx<real>)) + (j & 3u) * uint32_t(kFusedHalvesImportPitch) - x0;
            };
            factor_side_body_pairs<real, 2, true>(reinterpret_cast<const cx<real>*>(lds), bits, mask, diag, wave < kWaves ? kWaves * hh + wave : 0xffffffu, 2u * kWaves, stage,
                                                  dstage_all + w * 64, sink, dq_pairs QSV_PSTAMP_ARGS, row_of);
        } else if (table_lds_rows)
            factor_side_body_pairs<real, 2, true>(reinterpret_cast<const cx<real>*>(lds), bits, mask, diag, wave < kWaves ? wave : 0xffffffu, step, stage,
                                                  dstage_all + w * 64, sink, dq_pairs QSV_PSTAMP_ARGS);
        else
            factor_side_body_pairs<real, 2, false>(tab, bits, mask, diag, wave < kWaves ? wave : 0xffffffu, step, stage, dstage_all + w * 64, sink, dq_pairs QSV_PSTAMP_ARGS);
#endif
    }
    __syncthreads();
    QSV_PSTAMP(1);  // Gram matrices
#ifdef QSV_ABL_NO_HANDOFF  // (measurement: no hand-off between the sides, no combination)
    return;
#endif
    double* slot = a.factor_scratch + size_t(ev.state_slot) * factor_slot_doubles();
    double* mine = slot + (size_t(xy) * kFactorSlices + hh) * kFactorWeights * 64;  // (a half side: slice h)
    // (write-through stores: the few hundred bytes the other side will read must not wait for a write-back of everything
    // this XCD's L2 holds dirty -- the side tables of sixteen workgroups; an agent-scope release did that: 46 us per launch)
    for (uint32_t idx = tid; idx < (2u + bits) * NQ; idx += blockDim.x) {
        const uint32_t w = idx / NQ, pi = idx % NQ;
        __hip_atomic_store(mine + w * 64 + pi, factor_sum_partials(partial, kWaves, w, pi, n_keys), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave, before the barrier the signalling lane joins
    __syncthreads();
    if (tid == 0) {
        // (four per evaluation: a side's one workgroup adds two, a half side's one -- whoever completes the four combines)
        const uint32_t add = halves ? 1u : 2u;
        const uint32_t before = __hip_atomic_fetch_add(a.factor_counters + size_t(kFactorCountersPerSlot) * ev.state_slot, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = ((before + add) & 3u) == 0u;  // (the add has returned: its value is used)
    }
    __syncthreads();
    QSV_PSTAMP(2);  // partial matrices out, counter
    if (!*flag) return;
    // ---- the combination (factor_combine_kernel's, with one slice per side) ----
    const uint32_t bx = sp[1], by = sp[2];
    const uint32_t masks[2] = {sp[kSplitMaskX], sp[kSplitMaskY]};
    double coupling = 0.0;
    uint32_t qa_i = 0, qb_i = 0;
    if (tid < bx * by) {
        qa_i = tid / by;
        qb_i = tid % by;
        uint32_t qa = 0, qb = 0;
        for (uint32_t m = masks[0], k = 0; m; m &= m - 1, ++k)
            if (k == qa_i) qa = uint32_t(__builtin_ctz(m));
        for (uint32_t m = masks[1], k = 0; m; m &= m - 1, ++k)
            if (k == qb_i) qb = uint32_t(__builtin_ctz(m));
        coupling = a.quad[qa * a.n_full + qb];
    }
    const double d00 = whole_diag[0];
    double* gram = reinterpret_cast<double*>(lds);  // [side][weight][kFactorPitch] (the partial matrices are no longer needed)
    const uint32_t side_bits[2] = {bx, by};
    {
        // (all of a thread's loads first, then its LDS writes: at eight terms a thread has up to five values per side to fetch,
        // and a loop that stored each before asking for the next paid a memory round trip per value)
        constexpr uint32_t kMaxPerThread = (kFactorWeights * 64 + 255) / 256;
        double fetched[2][kMaxPerThread];
#pragma unroll
        for (uint32_t s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (uint32_t it = 0; it < kMaxPerThread; ++it) {
                const uint32_t idx = tid + it * blockDim.x;
                fetched[s2][it] = 0.0;
                if (idx < (2u + side_bits[s2]) * NQ) {
                    const double* entry = slot + size_t(s2) * kFactorSlices * kFactorWeights * 64 + (idx / NQ) * 64 + idx % NQ;
                    fetched[s2][it] = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // (half sides: the sums over the two halves of x, the lower first)
                    if ((ev.flags & kEvalHalves) && side_bits[s2] + n_keys == uint32_t(kFusedLdsRowsBits))
                        fetched[s2][it] += __hip_atomic_load(entry + kFactorWeights * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
        for (uint32_t s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (uint32_t it = 0; it < kMaxPerThread; ++it) {
                const uint32_t idx = tid + it * blockDim.x;
                if (idx < (2u + side_bits[s2]) * NQ)
                    gram[size_t(s2) * kFactorWeights * kFactorPitch + (idx / NQ) * kFactorPitch + idx % NQ] = fetched[s2][it];
            }
    }
    __syncthreads();
    const double* g0 = gram;
    const double* g1 = gram + size_t(kFactorWeights) * kFactorPitch;
    double v = 0.0;
    if (tid < bx * by) {
        if (coupling != 0.0) v = 4.0 * coupling * factor_pairing(g0 + (2 + qa_i) * kFactorPitch, g1 + (2 + qb_i) * kFactorPitch, n_keys);
    } else if (tid == bx * by) {
        v = factor_pairing(g0 + kFactorPitch, g1, n_keys);  // D(x, 0)
    } else if (tid == bx * by + 1) {
        v = factor_pairing(g0, g1 + kFactorPitch, n_keys);  // D(0, y)
    } else if (tid == bx * by + 2) {
        v = -d00 * factor_pairing(g0, g1, n_keys);          // - D(0, 0) <psi|psi>
    }
    // fixed-order sum over the first 256 threads (bx by + 3 <= 172 of them carry a term)
    double* red = reinterpret_cast<double*>(flag) + 1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((tid & 63u) == 0 && wave < 4) red[wave] = v;
    __syncthreads();
    if (tid == 0) a.result_out[ev.out_index] = red[0] + red[1] + red[2] + red[3];
    QSV_PSTAMP(3);  // combination
}

hipError_t launch_factor(int dtype, unsigned n_evals, double* scratch, const double* quad, int n_qubits, hipStream_t stream,
                         const PassArgs& a, double* scratch_big, uint32_t* big_counters, int most_keys) {
    if (n_evals == 0) return hipSuccess;
    if (most_keys > 3 && (!scratch_big || !big_counters)) return hipErrorInvalidValue;
    // (an evaluation of 32 product terms has four groups of 256 entries: four times the workgroups)
    const dim3 grid(most_keys >= 5 ? 2 * kFactorBigSlices * 4 : most_keys == 4 ? 2 * kFactorBigSlices : kFactorParts, n_evals);
    if (dtype == 0)
        hipLaunchKernelGGL(factor_moments_kernel<double>, grid, dim3(256), 0, stream, a.plan, a.evals,
                           static_cast<const cx<double>*>(a.wtab), a.wtab_stride, a.diag, scratch, scratch_big, big_counters, a.side_diag);
    else
        hipLaunchKernelGGL(factor_moments_kernel<float>, grid, dim3(256), 0, stream, a.plan, a.evals,
                           static_cast<const cx<float>*>(a.wtab), a.wtab_stride, a.diag, scratch, scratch_big, big_counters, a.side_diag);
    hipLaunchKernelGGL(factor_combine_kernel, dim3(n_evals), dim3(most_keys > 3 ? 1024 : 256), 0, stream, a.plan, a.evals, scratch,
                       scratch_big, quad, uint32_t(n_qubits), a.diag, a.result_out);
    return hipGetLastError();
}

// ---- split evaluations under a general Pauli operator (kernels.hpp: launch_factor_terms) ----------------------------
__device__ __forceinline__ uint32_t extract_bits(uint32_t v, uint32_t mask) {  // the bits of v under mask, packed
    uint32_t out = 0, pos = 0;
    while (mask) {
        const uint32_t low = mask & (0u - mask);
        if (v & low) out |= 1u << pos;
        ++pos;
        mask ^= low;
    }
    return out;
}

// M[j'][j] = sum_u conj(T_j'[u ^ f]) (-1)^popcount(u & z) T_j[u] for one side table; lane (run, j', j) adds its run of
// every block of 64 values of u (the block and its partner block u ^ f staged in the wave's LDS region), the runs are
// added across lanes: every lane of the first run ends up with its entry (re, im).
template <typename real, int J>
__device__ __forceinline__ void factor_term_side(const cx<real>* __restrict__ tab, uint32_t bits, uint32_t f, uint32_t z,
                                                 cx<real>* stage, double* out_re, double* out_im) {
    constexpr uint32_t NQ = J * J;
    constexpr uint32_t PITCH = J + 1;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t pi = lane % NQ, sub = lane / NQ;
    const uint32_t jr = pi / J, jc = pi % J;  // row j' (conjugated), column j
    const uint32_t n_local = bits < 6u ? 1u << bits : 64u, n_blocks = bits < 6u ? 1u : 1u << (bits - 6u);
    const uint32_t f_low = f & 63u, f_high = f >> 6, z_low = z & 63u, z_high = z >> 6;
    cx<real>* here = stage;
    cx<real>* there = stage + 64 * PITCH;
    double acc_re = 0.0, acc_im = 0.0;
    for (uint32_t blk = 0; blk < n_blocks; ++blk) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            cx<real> v{real(0), real(0)}, w{real(0), real(0)};
            if (lane < n_local) {
                v = tab[(size_t(j) << bits) + size_t(blk) * 64 + lane];
                w = tab[(size_t(j) << bits) + size_t(blk ^ f_high) * 64 + lane];
            }
            here[lane * PITCH + uint32_t(j)] = v;
            there[lane * PITCH + uint32_t(j)] = w;
        }
        const uint32_t block_sign = uint32_t(__builtin_popcount(blk & z_high)) & 1u;
        double s_re = 0.0, s_im = 0.0;
#pragma unroll 4
        for (uint32_t i = 0; i < NQ; ++i) {
            const uint32_t ul = sub * NQ + i;
            const cx<real> a = there[(ul ^ f_low) * PITCH + jr], b = here[ul * PITCH + jc];
            const double ar = double(a.re), ai = double(a.im), br = double(b.re), bi = double(b.im);
            // conj(a) b, with the sign of this u
            const double pr = fma(ar, br, ai * bi), pim = fma(ar, bi, -ai * br);
            const bool minus = (uint32_t(__builtin_popcount(ul & z_low)) & 1u) != 0;
            s_re += minus ? -pr : pr;
            s_im += minus ? -pim : pim;
        }
        acc_re += block_sign ? -s_re : s_re;
        acc_im += block_sign ? -s_im : s_im;
    }
    for (uint32_t off = NQ; off < 64; off <<= 1) {
        acc_re += __shfl_xor(acc_re, int(off));
        acc_im += __shfl_xor(acc_im, int(off));
    }
    *out_re = acc_re;
    *out_im = acc_im;
}

template <typename real, int J>
__device__ double factor_terms_body(const cx<real>* __restrict__ X, const cx<real>* __restrict__ Y, const uint32_t (&bits)[2],
                                    const uint32_t (&mask)[2], const FactorTerm* __restrict__ terms, uint32_t n_terms,
                                    uint32_t first, uint32_t step, cx<real>* stage) {
    constexpr uint32_t NQ = J * J;
    const uint32_t lane = threadIdx.x & 63u;
    double total = 0.0;
    for (uint32_t k = first; k < n_terms; k += step) {
        const FactorTerm t = terms[k];
        const uint32_t fx = extract_bits(t.x, mask[0]), zx = extract_bits(t.z, mask[0]);
        const uint32_t fy = extract_bits(t.x, mask[1]), zy = extract_bits(t.z, mask[1]);
        double ar, ai, br, bi;
        factor_term_side<real, J>(X, bits[0], fx, zx, stage, &ar, &ai);
        factor_term_side<real, J>(Y, bits[1], fy, zy, stage, &br, &bi);
        // sum over the entries of A[j'j] B[j'j] (complex), then the power of i of the string's Y factors
        double sr = lane < NQ ? fma(ar, br, -ai * bi) : 0.0, si = lane < NQ ? fma(ar, bi, ai * br) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sr += __shfl_xor(sr, off);
            si += __shfl_xor(si, off);
        }
        const uint32_t ny = uint32_t(__builtin_popcount(t.x & t.z)) & 3u;
        const double value = ny == 0 ? sr : ny == 1 ? -si : ny == 2 ? -sr : si;  // Re(i^ny (sr + i si))
        total = fma(t.coeff, value, total);
    }
    return total;
}

template <typename real>
__global__ void __launch_bounds__(256, 2) factor_terms_kernel(const uint32_t* __restrict__ plan_arena, const EvalDesc* __restrict__ evals,
                                                              const cx<real>* __restrict__ sides, uint64_t side_stride,
                                                              const FactorTerm* __restrict__ terms, uint32_t n_terms,
                                                              double* __restrict__ partials) {
    __shared__ cx<real> stage_all[4 * 2 * 9 * 64];
    const EvalDesc ev = evals[blockIdx.y];
    if (!(ev.flags & kEvalSide)) return;
    const uint32_t* sp = plan_arena + ev.split_base;
    const uint32_t n_keys = sp[0];
    const bool swap = sp[3] & 1u;
    const uint32_t bits[2] = {sp[1], sp[2]}, mask[2] = {sp[kSplitMaskX], sp[kSplitMaskY]};
    const cx<real>* ta = sides + uint64_t(ev.state_slot) * side_stride;
    const cx<real>* X = ta + (swap ? side_stride >> 1 : 0);
    const cx<real>* Y = ta + (swap ? 0 : side_stride >> 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = blockDim.x >> 6;
    const uint32_t gw = blockIdx.x * n_waves + wave, step = gridDim.x * n_waves;
    cx<real>* stage = stage_all + size_t(wave) * 2 * 9 * 64;
    double total;
    if (n_keys == 0)
        total = factor_terms_body<real, 1>(X, Y, bits, mask, terms, n_terms, gw, step, stage);
    else if (n_keys == 1)
        total = factor_terms_body<real, 2>(X, Y, bits, mask, terms, n_terms, gw, step, stage);
    else if (n_keys == 2)
        total = factor_terms_body<real, 4>(X, Y, bits, mask, terms, n_terms, gw, step, stage);
    else
        total = factor_terms_body<real, 8>(X, Y, bits, mask, terms, n_terms, gw, step, stage);
    if ((threadIdx.x & 63u) == 0) partials[size_t(ev.out_index) * step + gw] = total;
}

hipError_t launch_factor_terms(int dtype, unsigned n_evals, const FactorTerm* terms, uint32_t n_terms, double* partials,
                               hipStream_t stream, const PassArgs& a) {
    if (n_evals == 0) return hipSuccess;
    const dim3 grid(kFactorTermWaves / 4, n_evals);
    if (dtype == 0)
        hipLaunchKernelGGL(factor_terms_kernel<double>, grid, dim3(256), 0, stream, a.plan, a.evals,
                           static_cast<const cx<double>*>(a.wtab), a.wtab_stride, terms, n_terms, partials);
    else
        hipLaunchKernelGGL(factor_terms_kernel<float>, grid, dim3(256), 0, stream, a.plan, a.evals,
                           static_cast<const cx<float>*>(a.wtab), a.wtab_stride, terms, n_terms, partials);
    return hipGetLastError();
}

// ---- exact-probability CVaR (kernels.hpp: launch_cvar_exact) -------------------------------------------------------------
// The probability of one basis state of one evaluation: an entry of the probabilities a gate pass left, or -- a split
// circuit -- |sum_j X_j[x(i)] Y_j[y(i)]|^2 from the two side tables.
template <typename real>
struct ExactSource {
    const double* probs = nullptr;
    const cx<real>* X = nullptr;
    const cx<real>* Y = nullptr;
    uint32_t bits_x = 0, bits_y = 0, mask_x = 0, mask_y = 0, terms = 0;
    __device__ __forceinline__ double operator()(uint32_t i) const {
        if (terms == 0) return probs[i];
        const uint32_t x = extract_bits(i, mask_x), y = extract_bits(i, mask_y);
        double pr = 0.0, pi = 0.0;
        for (uint32_t j = 0; j < terms; ++j) {
            const cx<real> a = X[(size_t(j) << bits_x) + x], b = Y[(size_t(j) << bits_y) + y];
            const double ar = double(a.re), ai = double(a.im), br = double(b.re), bi = double(b.im);
            pr = fma(ar, br, fma(-ai, bi, pr));
            pi = fma(ar, bi, fma(ai, br, pi));
        }
        return fma(pr, pr, pi * pi);
    }
};

template <typename real>
__device__ __forceinline__ ExactSource<real> exact_source(const uint32_t* __restrict__ plan_arena, const EvalDesc& ev, uint32_t position,
                                                          const double* __restrict__ probs_all, uint64_t dim,
                                                          const cx<real>* __restrict__ sides, uint64_t side_stride) {
    ExactSource<real> src;
    if (ev.flags & kEvalSide) {
        const uint32_t* sp = plan_arena + ev.split_base;
        const bool swap = sp[3] & 1u;
        const cx<real>* ta = sides + uint64_t(ev.state_slot) * side_stride;
        src.X = ta + (swap ? side_stride >> 1 : 0);
        src.Y = ta + (swap ? 0 : side_stride >> 1);
        src.bits_x = sp[1];
        src.bits_y = sp[2];
        src.mask_x = sp[kSplitMaskX];
        src.mask_y = sp[kSplitMaskY];
        src.terms = 1u << sp[0];
    } else {
        src.probs = probs_all + uint64_t(ev.state_slot) * dim;  // (the group's states: slot = position in the group's buffers)
    }
    (void)position;
    return src;
}

// chunk c of evaluation e: mass and mass x value of ranks [c kCvarChunk, (c + 1) kCvarChunk), every thread its strided share
// in ascending order, then the fixed-order block sums
template <typename real>
__global__ void __launch_bounds__(256) cvar_exact_chunks_kernel(const uint32_t* __restrict__ plan_arena, const EvalDesc* __restrict__ evals,
                                                                const double* __restrict__ probs_all, uint64_t dim,
                                                                const cx<real>* __restrict__ sides, uint64_t side_stride,
                                                                const uint32_t* __restrict__ order, const double* __restrict__ sorted,
                                                                uint32_t n_chunks, double* __restrict__ scratch) {
    __shared__ double red[4];
    const EvalDesc ev = evals[blockIdx.y];
    const ExactSource<real> src = exact_source<real>(plan_arena, ev, blockIdx.y, probs_all, dim, sides, side_stride);
    const uint64_t base = uint64_t(blockIdx.x) * kCvarChunk;
    double m = 0.0, w = 0.0;
#pragma unroll 4
    for (uint32_t k = 0; k < kCvarChunk / 256; ++k) {
        const uint64_t pos = base + uint64_t(k) * 256 + threadIdx.x;
        if (pos < dim) {
            const double p = src(order[pos]);
            m += p;
            w = fma(p, sorted[pos], w);
        }
    }
    const double mass = block_sum_256(m, red);
    const double wsum = block_sum_256(w, red);
    if (threadIdx.x == 0) {
        scratch[(size_t(blockIdx.y) * 2 + 0) * n_chunks + blockIdx.x] = mass;
        scratch[(size_t(blockIdx.y) * 2 + 1) * n_chunks + blockIdx.x] = wsum;
    }
}

// one workgroup per evaluation: the chunk in which the gathered mass comes within numpy.isclose of alpha, the rank inside it,
// and the accumulation up to there (see kernels.hpp for the rule being restated)
template <typename real>
__global__ void __launch_bounds__(256) cvar_exact_finish_kernel(const uint32_t* __restrict__ plan_arena, const EvalDesc* __restrict__ evals,
                                                                const double* __restrict__ probs_all, uint64_t dim,
                                                                const cx<real>* __restrict__ sides, uint64_t side_stride,
                                                                const uint32_t* __restrict__ order, const double* __restrict__ sorted,
                                                                uint32_t n_chunks, const double* __restrict__ scratch, double alpha,
                                                                double* __restrict__ out) {
    __shared__ double sh_a[256], sh_b[256];
    __shared__ uint32_t sh_i[256];
    __shared__ double pick[4];
    __shared__ uint32_t pick_i[2];
    const EvalDesc ev = evals[blockIdx.x];
    const uint32_t t = threadIdx.x;
    const double* mass = scratch + (size_t(blockIdx.x) * 2 + 0) * n_chunks;
    const double* wsum = scratch + (size_t(blockIdx.x) * 2 + 1) * n_chunks;
    const double target = alpha - (1e-8 + 1e-5 * fabs(alpha));  // isclose(gathered, alpha) with gathered <= alpha
    // ---- which chunk ----
    const uint32_t per = (n_chunks + 255) / 256;
    const uint32_t lo = min(n_chunks, t * per), hi = min(n_chunks, lo + per);
    double local = 0.0;
    for (uint32_t c = lo; c < hi; ++c) local += mass[c];
    sh_a[t] = local;
    __syncthreads();
    if (t == 0) {
        double run = 0.0;
        for (int i = 0; i < 256; ++i) {
            const double v = sh_a[i];
            sh_a[i] = run;
            run += v;
        }
    }
    __syncthreads();
    {
        double cum = sh_a[t], before = 0.0;
        uint32_t found = n_chunks;
        for (uint32_t c = lo; c < hi; ++c) {
            const double next = cum + mass[c];
            if (found == n_chunks && next >= target) {
                found = c;
                before = cum;
            }
            cum = next;
        }
        sh_i[t] = found;
        sh_b[t] = before;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t chunk = n_chunks;
        double before = 0.0;
        for (int i = 0; i < 256 && chunk == n_chunks; ++i)
            if (sh_i[i] < n_chunks) {
                chunk = sh_i[i];
                before = sh_b[i];
            }
        pick_i[0] = chunk;
        pick[0] = before;
    }
    __syncthreads();
    const uint32_t chunk = pick_i[0];
    const double mass_before = pick[0];
    // ---- the accumulation over the chunks in front of it ----
    double part = 0.0;
    for (uint32_t c = lo; c < hi && c < chunk; ++c) part += wsum[c];
    __syncthreads();
    sh_a[t] = part;
    __syncthreads();
    if (t == 0) {
        double run = 0.0;
        for (int i = 0; i < 256; ++i) run += sh_a[i];
        pick[1] = run;
    }
    __syncthreads();
    const double sum_before = pick[1];
    if (chunk == n_chunks) {  // (the whole distribution carries less than alpha - tolerance: everything counts)
        if (t == 0) out[ev.out_index] = sum_before / alpha;
        return;
    }
    // ---- which rank inside the chunk: thread t owns ranks t 16 .. t 16 + 15 of it ----
    const ExactSource<real> src = exact_source<real>(plan_arena, ev, blockIdx.x, probs_all, dim, sides, side_stride);
    constexpr uint32_t OWN = kCvarChunk / 256;
    double p[OWN], v[OWN];
    double lm = 0.0, lw = 0.0;
#pragma unroll
    for (uint32_t k = 0; k < OWN; ++k) {
        const uint64_t pos = uint64_t(chunk) * kCvarChunk + uint64_t(t) * OWN + k;
        p[k] = pos < dim ? src(order[pos]) : 0.0;
        v[k] = pos < dim ? sorted[pos] : 0.0;
        lm += p[k];
        lw = fma(p[k], v[k], lw);
    }
    __syncthreads();
    sh_a[t] = lm;
    sh_b[t] = lw;
    __syncthreads();
    if (t == 0) {
        double run = 0.0;
        for (int i = 0; i < 256; ++i) {
            const double x = sh_a[i];
            sh_a[i] = run;
            run += x;
        }
    }
    __syncthreads();
    {
        double cum = mass_before + sh_a[t];
        uint32_t found = OWN;
        double inside = 0.0, last = 0.0;
#pragma unroll
        for (uint32_t k = 0; k < OWN; ++k) {
            const double next = cum + p[k];
            if (found == OWN) {
                if (next >= target) {
                    found = k;
                    last = fmin(alpha - cum, p[k]) * v[k];  // the last state's mass, clipped to what is missing
                } else {
                    inside = fma(p[k], v[k], inside);
                }
            }
            cum = next;
        }
        sh_i[t] = found;
        sh_a[t] = inside;  // (this thread's ranks in front of the last one)
        __syncthreads();
        if (t == 0) {
            double run = 0.0;
            bool done = false;
            for (int i = 0; i < 256 && !done; ++i) {
                if (sh_i[i] < OWN) {
                    run += sh_a[i];
                    pick_i[1] = uint32_t(i);
                    done = true;
                } else {
                    run += sh_b[i];  // a thread in front of the crossing: all of its ranks
                }
            }
            pick[2] = run;
            if (!done) pick_i[1] = 256;  // (rounding: the chunk's own sum fell short of what the chunk scan saw)
        }
        __syncthreads();
        if (pick_i[1] == 256) {
            if (t == 0) out[ev.out_index] = (sum_before + pick[2]) / alpha;
        } else if (t == pick_i[1]) {
            out[ev.out_index] = (sum_before + pick[2] + last) / alpha;
        }
    }
}

hipError_t launch_cvar_exact(int dtype, const double* probs, uint64_t dim, unsigned n_evals, const uint32_t* order,
                             const double* sorted_values, double alpha, double* chunk_scratch, double* out, hipStream_t stream,
                             const PassArgs& a) {
    if (n_evals == 0) return hipSuccess;
    if (!(alpha > 0.0) || alpha > 1.0 || dim > (uint64_t(1) << 30)) return hipErrorInvalidValue;
    const uint32_t n_chunks = cvar_exact_chunks(dim);
    if (dtype == 0) {
        hipLaunchKernelGGL(cvar_exact_chunks_kernel<double>, dim3(n_chunks, n_evals), dim3(256), 0, stream, a.plan, a.evals, probs, dim,
                           static_cast<const cx<double>*>(a.wtab), a.wtab_stride, order, sorted_values, n_chunks, chunk_scratch);
        hipLaunchKernelGGL(cvar_exact_finish_kernel<double>, dim3(n_evals), dim3(256), 0, stream, a.plan, a.evals, probs, dim,
                           static_cast<const cx<double>*>(a.wtab), a.wtab_stride, order, sorted_values, n_chunks, chunk_scratch, alpha, out);
    } else {
        hipLaunchKernelGGL(cvar_exact_chunks_kernel<float>, dim3(n_chunks, n_evals), dim3(256), 0, stream, a.plan, a.evals, probs, dim,
                           static_cast<const cx<float>*>(a.wtab), a.wtab_stride, order, sorted_values, n_chunks, chunk_scratch);
        hipLaunchKernelGGL(cvar_exact_finish_kernel<float>, dim3(n_evals), dim3(256), 0, stream, a.plan, a.evals, probs, dim,
                           static_cast<const cx<float>*>(a.wtab), a.wtab_stride, order, sorted_values, n_chunks, chunk_scratch, alpha, out);
    }
    return hipGetLastError();
}

static unsigned stream_blocks(uint64_t dim) {
    const uint64_t want = (dim + 255) / 256;
    return unsigned(want < 4096 ? want : 4096);
}

hipError_t launch_probabilities(int dtype, const void* state, uint64_t dim, int n_slots, double* probs,
                                hipStream_t stream) {
    const dim3 grid(stream_blocks(dim), n_slots);
    if (dtype == 0)
        hipLaunchKernelGGL(probabilities_kernel<double>, grid, dim3(256), 0, stream,
                           reinterpret_cast<const cx<double>*>(state), dim, probs);
    else
        hipLaunchKernelGGL(probabilities_kernel<float>, grid, dim3(256), 0, stream,
                           reinterpret_cast<const cx<float>*>(state), dim, probs);
    return hipGetLastError();
}

hipError_t launch_state_to_f64(int dtype, const void* state, uint64_t dim, double* out_re_im, hipStream_t stream) {
    if (dtype == 0)
        hipLaunchKernelGGL(state_to_f64_kernel<double>, dim3(stream_blocks(dim)), dim3(256), 0, stream,
                           reinterpret_cast<const cx<double>*>(state), dim, out_re_im);
    else
        hipLaunchKernelGGL(state_to_f64_kernel<float>, dim3(stream_blocks(dim)), dim3(256), 0, stream,
                           reinterpret_cast<const cx<float>*>(state), dim, out_re_im);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) side_diag_kernel(const double* __restrict__ diag, double* __restrict__ side_diag, const SideDiagJobs jobs) {
    const SideDiagJob j = jobs.job[blockIdx.y];
    const uint32_t count = 1u << j.bits;
    for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < count; x += gridDim.x * blockDim.x)
        side_diag[size_t(j.base) + x] = diag[deposit_bits(x, j.mask)];
}

hipError_t launch_side_diag(const double* diag, double* side_diag, const SideDiagJobs& jobs, int n_jobs, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    uint32_t most = 0;
    for (int i = 0; i < n_jobs; ++i) most = jobs.job[i].bits > most ? jobs.job[i].bits : most;
    const uint32_t blocks = most <= 8 ? 1u : 1u << (most - 8);
    hipLaunchKernelGGL(side_diag_kernel, dim3(blocks < 16u ? blocks : 16u, unsigned(n_jobs)), dim3(256), 0, stream, diag, side_diag, jobs);
    return hipGetLastError();
}

hipError_t read_stamps(unsigned long long* out, int reset) {
#ifdef QSV_STAMPS
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    e = hipMemcpyFromSymbol(out, HIP_SYMBOL(qsv_stamp_table), sizeof(unsigned long long) * kStampPasses * kStampPhases);
    if (e != hipSuccess || !reset) return e;
    static const unsigned long long zeros[kStampPasses * kStampPhases] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(qsv_stamp_table), zeros, sizeof(zeros));
#else
    (void)out;
    (void)reset;
    return hipErrorNotSupported;
#endif
}

hipError_t read_timeline(unsigned long long* out, size_t max_words, unsigned int* n_records, int reset) {
#ifdef QSV_TIMELINE
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    unsigned int count = 0;
    e = hipMemcpyFromSymbol(&count, HIP_SYMBOL(qsv_timeline_count), sizeof(count));
    if (e != hipSuccess) return e;
    if (count > kTimelineWgs) count = kTimelineWgs;
    if (size_t(count) * kTimelineWords > max_words) count = unsigned(max_words / kTimelineWords);
    *n_records = count;
    if (count) {
        e = hipMemcpyFromSymbol(out, HIP_SYMBOL(qsv_timeline), sizeof(unsigned long long) * size_t(count) * kTimelineWords);
        if (e != hipSuccess) return e;
    }
    if (!reset) return hipSuccess;
    const unsigned int zero = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(qsv_timeline_count), &zero, sizeof(zero));
#else
    (void)out;
    (void)max_words;
    (void)n_records;
    (void)reset;
    return hipErrorNotSupported;
#endif
}

}  // namespace qsv
