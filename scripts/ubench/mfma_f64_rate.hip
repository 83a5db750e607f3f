// fp64 through the matrix pipe on gfx950, alone and next to fp64 VALU work of the same wave: does v_mfma_f64_4x4x4_4b_f64
// (four independent 4x4 * 4x4 products per instruction: a 2x2 complex gate as a real 4x4 matrix on 16 amplitude pairs) run
// beside v_fma_f64, and at what rate?  The deep gate pass is bound by fp64 VALU issue: a second pipe is the
// one thing that could lift that bound.
//   MODE 0: 16 independent v_fma_f64 per iteration        MODE 1: 16 independent mfma 4x4x4_4b per iteration
//   MODE 2: 16 + 16 interleaved                           MODE 3: 16 mfma 16x16x4 per iteration     MODE 4: 16 + 16 (16x16x4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, int iters) {
    double a[16], m[16];
    double4_t big[8];
    const double v0 = out[threadIdx.x], v1 = out[threadIdx.x + 256];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = v0 + i;
        m[i] = v1 - i;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) big[i] = double4_t{v0, v1, v0 + i, v1 - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0 || MODE == 2 || MODE == 4) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(v0), "v"(v1));
            if (MODE == 1 || MODE == 2) m[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(v0, v1, m[i], 0, 0, 0);
            if (MODE == 3 || MODE == 4) big[i & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v1, big[i & 7], 0, 0, 0);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + m[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += big[i][0] + big[i][1] + big[i][2] + big[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double* d_out, int waves_per_simd, double fma_per_iter_lane_valu, double fma_per_instr_mfma) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = double(blocks) * 4;
    const double valu = (MODE == 0 || MODE == 2 || MODE == 4) ? waves * iters * 16 : 0;  // wave-instructions
    const double mfma = (MODE >= 1) ? waves * iters * 16 : 0;
    const double flops = 2.0 * (valu * 64 + mfma * fma_per_instr_mfma);
    const double cycles = ms * 1e-3 * 2.4e9;  // per SIMD
    printf("%-34s waves/SIMD=%d  %8.3f ms  cycles per SIMD per (VALU instr %.2f, MFMA instr %.2f)  %.1f TFLOP/s fp64\n", name,
           waves_per_simd, ms, valu ? cycles / (valu / 1024) : 0.0, mfma ? cycles / (mfma / 1024) : 0.0, flops / (ms * 1e-3) / 1e12);
    (void)fma_per_iter_lane_valu;
}

int main() {
    double* d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(double));
    std::vector<double> h(256 * 8 * 256, 1e-3);
    hipMemcpy(d_out, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64 alone", d_out, w, 16, 0);
        run<1>("mfma_f64_4x4x4_4b alone", d_out, w, 0, 256);
        run<2>("v_fma_f64 + mfma 4x4x4 interleaved", d_out, w, 16, 256);
        run<3>("mfma_f64_16x16x4 alone", d_out, w, 0, 1024);
        run<4>("v_fma_f64 + mfma 16x16x4", d_out, w, 16, 1024);
    }
    return 0;
}
