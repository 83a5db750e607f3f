// HIP kernels for gfx950 (MI355X).  See plan.hpp for the pass/round/exchange model and the plan encoding.
//
// pass_kernel: one workgroup owns one tile of 2^k amplitudes for a whole pass.
//   HBM  -> registers : each thread loads 2^R amplitudes, 16 B (one complex double) per lane per instruction,
//                       lanes on the lowest tile bits (>= 256 B contiguous per 16 lanes)
//   rounds            : 2x2 butterflies between registers of one thread (v_fma_f64), gate matrices are uniform
//                       and arrive through scalar loads
//   exchanges         : tile transposed through LDS (ds_write_b128 / ds_read_b128) under a host-chosen XOR swizzle
//                       that makes both sides bank-conflict free
//   registers -> HBM  : same layout as the load; or, on the last pass of an evaluation with a diagonal operator,
//                       no store at all: sum_i |a_i|^2 D[i] is reduced on chip and one double per workgroup leaves
// No MFMA: a 2x2 gate over 32 B of traffic is 0.4 flop/B, the kernel is HBM bound by construction.
#include "kernels.hpp"
#include "plan.hpp"

namespace qsv {

template <typename real>
struct alignas(2 * sizeof(real)) cx {
    real re, im;
};

__device__ __forceinline__ uint32_t xor_columns(const uint32_t* __restrict__ cols, int t, uint32_t tid) {
    uint32_t x = 0;
    for (int u = 0; u < t; ++u) x ^= (0u - ((tid >> u) & 1u)) & cols[u];
    return x;
}

template <int R>
__device__ __forceinline__ void register_offsets(const uint32_t* __restrict__ rc, uint32_t (&ro)[1 << R]) {
    ro[0] = 0;
#pragma unroll
    for (int e = 1; e < (1 << R); ++e) ro[e] = ro[e & (e - 1)] ^ rc[__builtin_ctz(e)];
}

template <typename real, int R, int J>
__device__ __forceinline__ void butterfly(cx<real> (&amp)[1 << R], const real (&m)[8], uint32_t cr, bool on) {
    if (on) {
#pragma unroll
        for (int e0 = 0; e0 < (1 << R); ++e0) {
            if (e0 & (1 << J)) continue;
            if ((uint32_t(e0) & cr) == cr) {
                constexpr int bit = 1 << J;
                const cx<real> a0 = amp[e0], a1 = amp[e0 | bit];
                amp[e0].re = m[0] * a0.re - m[1] * a0.im + m[2] * a1.re - m[3] * a1.im;
                amp[e0].im = m[0] * a0.im + m[1] * a0.re + m[2] * a1.im + m[3] * a1.re;
                amp[e0 | bit].re = m[4] * a0.re - m[5] * a0.im + m[6] * a1.re - m[7] * a1.im;
                amp[e0 | bit].im = m[4] * a0.im + m[5] * a0.re + m[6] * a1.im + m[7] * a1.re;
            }
        }
    }
}

template <typename real, int R, int J>
struct ButterflyDispatch {
    static __device__ __forceinline__ void run(int j, cx<real> (&amp)[1 << R], const real (&m)[8], uint32_t cr, bool on) {
        if (j == J)
            butterfly<real, R, J>(amp, m, cr, on);
        else
            ButterflyDispatch<real, R, J - 1>::run(j, amp, m, cr, on);
    }
};
template <typename real, int R>
struct ButterflyDispatch<real, R, -1> {
    static __device__ __forceinline__ void run(int, cx<real> (&)[1 << R], const real (&)[8], uint32_t, bool) {}
};

template <typename real, int R>
__global__ void __launch_bounds__(256) pass_kernel(const PassArgs a) {
    using cxr = cx<real>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    cxr* lds = reinterpret_cast<cxr*>(lds_raw);

    const EvalDesc ev = a.evals[blockIdx.y];
    const uint32_t* __restrict__ cp = a.plan + ev.plan_base;
    const uint32_t n_passes = cp[0];
    if (a.pass_index >= n_passes) return;
    const uint32_t* __restrict__ pp = cp + cp[2 + a.pass_index];
    const uint32_t hdr = pp[0];
    const int k = hdr & 0xff, t = (hdr >> 16) & 0xff, n_rounds = hdr >> 24;
    const uint32_t tid = threadIdx.x;
    const bool active = tid < (1u << t);

    // fixed (non-tile) index bits of this workgroup: spread blockIdx.x around the tile positions
    uint64_t base = blockIdx.x;
    for (int j = 0; j < k; ++j) {
        const uint32_t p = pp[kPassHeaderWords + j];
        base = ((base >> p) << (p + 1)) | (base & ((uint64_t(1) << p) - 1));
    }

    const uint32_t* __restrict__ gl = pp + kPassHeaderWords + k;
    const uint32_t* __restrict__ gs = gl + (t + R);
    const uint32_t* __restrict__ rp = gs + (t + R);

    cxr* __restrict__ st = reinterpret_cast<cxr*>(a.states) + uint64_t(ev.state_slot) * a.state_stride + base;
    cxr amp[1 << R];

    {
        const uint32_t tg = xor_columns(gl, t, tid);
        uint32_t ro[1 << R];
        register_offsets<R>(gl + t, ro);
        if (a.pass_index == 0 && (a.mode & kModeSynthFirst)) {
#pragma unroll
            for (int e = 0; e < (1 << R); ++e) {
                amp[e].re = (base == 0 && (tg ^ ro[e]) == 0) ? real(1) : real(0);
                amp[e].im = real(0);
            }
        } else if (active) {
#pragma unroll
            for (int e = 0; e < (1 << R); ++e) amp[e] = st[tg ^ ro[e]];
        }
    }

    bool lds_dirty = false;
    for (int m = 0; m < n_rounds; ++m) {
        const uint32_t rh = rp[0];
        const int n_gates = rh & 0xffff;
        rp += 1;
        if ((rh >> 16) & 1u) {
            const uint32_t* __restrict__ wc = rp;
            const uint32_t* __restrict__ rc = rp + (t + R);
            rp += 2 * (t + R);
            if (lds_dirty) __syncthreads();  // everyone has finished reading the previous exchange
            if (active) {
                const uint32_t wt = xor_columns(wc, t, tid);
                uint32_t wo[1 << R];
                register_offsets<R>(wc + t, wo);
#pragma unroll
                for (int e = 0; e < (1 << R); ++e) lds[wt ^ wo[e]] = amp[e];
            }
            __syncthreads();
            if (active) {
                const uint32_t rt = xor_columns(rc, t, tid);
                uint32_t ro[1 << R];
                register_offsets<R>(rc + t, ro);
#pragma unroll
                for (int e = 0; e < (1 << R); ++e) amp[e] = lds[rt ^ ro[e]];
            }
            lds_dirty = true;
        }
        for (int g = 0; g < n_gates; ++g, rp += kGateWords) {
            const uint32_t w0 = rp[0], cr = rp[1], ct = rp[2], cg = rp[3];
            if ((uint32_t(base) & cg) != cg) continue;  // control is one of this workgroup's fixed bits and is 0
            const double* __restrict__ mp = a.mats + ev.mat_base + size_t(w0 >> 8) * 8;
            real mm[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) mm[i] = real(mp[i]);
            const bool on = active && ((tid & ct) == ct);
            ButterflyDispatch<real, R, R - 1>::run(int(w0 & 0xff), amp, mm, cr, on);
        }
    }

    const bool last = (a.pass_index + 1 == n_passes);
    const uint32_t sg = xor_columns(gs, t, tid);
    uint32_t so[1 << R];
    register_offsets<R>(gs + t, so);
    if ((!last || (a.mode & kModeFinalStore)) && active) {
#pragma unroll
        for (int e = 0; e < (1 << R); ++e) st[sg ^ so[e]] = amp[e];
    }
    if (last && (a.mode & kModeFinalDiag)) {
        double acc = 0.0;
        if (active) {
            const double* __restrict__ d = a.diag + base;
#pragma unroll
            for (int e = 0; e < (1 << R); ++e) {
                const double re = double(amp[e].re), im = double(amp[e].im);
                acc += (re * re + im * im) * d[sg ^ so[e]];
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        double* red = reinterpret_cast<double*>(lds_raw);
        if (lds_dirty) __syncthreads();
        if ((tid & 63u) == 0) red[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
            double total = 0.0;
            const int n_waves = (blockDim.x + 63) >> 6;
            for (int w = 0; w < n_waves; ++w) total += red[w];
            a.partials[size_t(ev.out_index) * a.blocks_per_state + blockIdx.x] = total;
        }
    }
}

template <typename real, int R>
static hipError_t launch_pass_t(dim3 grid, int threads, size_t lds_bytes, hipStream_t stream, const PassArgs& args) {
    // the block reduction at the end needs one double per wave
    const size_t lds = lds_bytes < 64 ? 64 : lds_bytes;
    hipLaunchKernelGGL((pass_kernel<real, R>), grid, dim3(threads), lds, stream, args);
    return hipGetLastError();
}

template <typename real>
static hipError_t launch_pass_r(int r, dim3 grid, int threads, size_t lds_bytes, hipStream_t stream,
                                const PassArgs& args) {
    switch (r) {
        case 1: return launch_pass_t<real, 1>(grid, threads, lds_bytes, stream, args);
        case 2: return launch_pass_t<real, 2>(grid, threads, lds_bytes, stream, args);
        case 3: return launch_pass_t<real, 3>(grid, threads, lds_bytes, stream, args);
        case 4: return launch_pass_t<real, 4>(grid, threads, lds_bytes, stream, args);
        case 5: return launch_pass_t<real, 5>(grid, threads, lds_bytes, stream, args);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pass(int dtype, int r, dim3 grid, int threads, size_t lds_bytes, hipStream_t stream,
                       const PassArgs& args) {
    if (threads > 256) return hipErrorInvalidValue;
    return dtype == 0 ? launch_pass_r<double>(r, grid, threads, lds_bytes, stream, args)
                      : launch_pass_r<float>(r, grid, threads, lds_bytes, stream, args);
}

template <typename real, int R>
static hipError_t configure_t(size_t lds_bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_kernel<real, R>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes < 64 ? 64 : lds_bytes));
}

hipError_t configure_pass_kernels(int dtype, int r, size_t lds_bytes) {
#define QSV_CFG(real)                                   \
    switch (r) {                                        \
        case 1: return configure_t<real, 1>(lds_bytes); \
        case 2: return configure_t<real, 2>(lds_bytes); \
        case 3: return configure_t<real, 3>(lds_bytes); \
        case 4: return configure_t<real, 4>(lds_bytes); \
        case 5: return configure_t<real, 5>(lds_bytes); \
        default: return hipErrorInvalidValue;           \
    }
    if (dtype == 0) {
        QSV_CFG(double)
    } else {
        QSV_CFG(float)
    }
#undef QSV_CFG
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
                                                              double* __restrict__ out) {
    __shared__ double red[4];
    const double* p = partials + size_t(blockIdx.x) * blocks;
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < blocks; i += 256) acc += p[i];
    const double total = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = total;
}

hipError_t launch_reduce_partials(const double* partials, uint32_t blocks, int n_evals, double* out,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(n_evals), dim3(256), 0, stream, partials, blocks, out);
    return hipGetLastError();
}

// ---- general Pauli terms -------------------------------------------------------------------------------
template <typename real>
__global__ void __launch_bounds__(256) pauli_terms_kernel(const cx<real>* __restrict__ states, uint64_t state_stride,
                                                          uint64_t dim, int n_terms,
                                                          const uint64_t* __restrict__ x_mask,
                                                          const uint64_t* __restrict__ z_mask,
                                                          double* __restrict__ term_partials) {
    __shared__ double red[4];
    const int term = blockIdx.y, slot = blockIdx.z;
    const uint64_t x = x_mask[term], z = z_mask[term];
    const cx<real>* __restrict__ st = states + uint64_t(slot) * state_stride;
    double acc_re = 0.0, acc_im = 0.0;
    const uint64_t stride = uint64_t(gridDim.x) * 256;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < dim; i += stride) {
        const uint64_t j = i ^ x;
        const cx<real> a = st[i], b = st[j];
        const double sgn = (__popcll(j & z) & 1) ? -1.0 : 1.0;
        acc_re += sgn * (double(a.re) * double(b.re) + double(a.im) * double(b.im));
        acc_im += sgn * (double(a.re) * double(b.im) - double(a.im) * double(b.re));
    }
    const double tr = block_sum_256(acc_re, red);
    const double ti = block_sum_256(acc_im, red);
    if (threadIdx.x == 0) {
        double* o = term_partials + ((size_t(slot) * n_terms + term) * gridDim.x + blockIdx.x) * 2;
        o[0] = tr;
        o[1] = ti;
    }
}

hipError_t launch_pauli_terms(int dtype, const void* states, uint64_t state_stride, int n_qubits, int n_slots,
                              int n_terms, const uint64_t* x_mask, const uint64_t* z_mask, int nb,
                              double* term_partials, hipStream_t stream) {
    const uint64_t dim = uint64_t(1) << n_qubits;
    dim3 grid(nb, n_terms, n_slots);
    if (dtype == 0)
        hipLaunchKernelGGL(pauli_terms_kernel<double>, grid, dim3(256), 0, stream,
                           reinterpret_cast<const cx<double>*>(states), state_stride, dim, n_terms, x_mask, z_mask,
                           term_partials);
    else
        hipLaunchKernelGGL(pauli_terms_kernel<float>, grid, dim3(256), 0, stream,
                           reinterpret_cast<const cx<float>*>(states), state_stride, dim, n_terms, x_mask, z_mask,
                           term_partials);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) pauli_combine_kernel(const double* __restrict__ term_partials, int n_terms,
                                                            int nb, const uint64_t* __restrict__ x_mask,
                                                            const uint64_t* __restrict__ z_mask,
                                                            const double* __restrict__ coeff_re,
                                                            const double* __restrict__ coeff_im,
                                                            const EvalDesc* __restrict__ evals,
                                                            double* __restrict__ out) {
    __shared__ double red[4];
    const int slot = blockIdx.x;
    double acc = 0.0;
    for (int k = threadIdx.x; k < n_terms; k += 256) {
        const double* p = term_partials + (size_t(slot) * n_terms + k) * nb * 2;
        double tr = 0.0, ti = 0.0;
        for (int b = 0; b < nb; ++b) {
            tr += p[2 * b];
            ti += p[2 * b + 1];
        }
        // multiply by i^{ny}
        const int ny = __popcll(x_mask[k] & z_mask[k]) & 3;
        double pr = tr, pi = ti;
        if (ny == 1) { pr = -ti; pi = tr; }
        else if (ny == 2) { pr = -tr; pi = -ti; }
        else if (ny == 3) { pr = ti; pi = -tr; }
        acc += coeff_re[k] * pr - coeff_im[k] * pi;  // real part of coeff * value
    }
    const double total = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[evals[slot].out_index] = total;
}

hipError_t launch_pauli_combine(const double* term_partials, int n_slots, int n_terms, int nb, const uint64_t* x_mask,
                                const uint64_t* z_mask, const double* coeff_re, const double* coeff_im,
                                const EvalDesc* evals, double* out, hipStream_t stream) {
    hipLaunchKernelGGL(pauli_combine_kernel, dim3(n_slots), dim3(256), 0, stream, term_partials, n_terms, nb, x_mask,
                       z_mask, coeff_re, coeff_im, evals, out);
    return hipGetLastError();
}

// ---- state read-out ------------------------------------------------------------------------------------
template <typename real>
__global__ void __launch_bounds__(256) probabilities_kernel(const cx<real>* __restrict__ st, uint64_t dim,
                                                            double* __restrict__ probs) {
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

static unsigned stream_blocks(uint64_t dim) {
    const uint64_t want = (dim + 255) / 256;
    return unsigned(want < 4096 ? want : 4096);
}

hipError_t launch_probabilities(int dtype, const void* state, uint64_t dim, double* probs, hipStream_t stream) {
    if (dtype == 0)
        hipLaunchKernelGGL(probabilities_kernel<double>, dim3(stream_blocks(dim)), dim3(256), 0, stream,
                           reinterpret_cast<const cx<double>*>(state), dim, probs);
    else
        hipLaunchKernelGGL(probabilities_kernel<float>, dim3(stream_blocks(dim)), dim3(256), 0, stream,
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

}  // namespace qsv
