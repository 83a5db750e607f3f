// Measures issue rates of v_fma_f64 (VGPR and SGPR operand forms) and v_mov_b64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, const double* coef, int iters) {
    double a[16];
    const double c0 = coef[0], c1 = coef[1];   // uniform -> SGPR
    const double v0 = out[threadIdx.x];         // per-lane -> VGPR
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = v0 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) a[i] = fma(a[i], c0, c1 /*sgpr,sgpr -> one goes to vgpr*/);
            if (MODE == 1) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(v0), "v"(a[(i + 5) & 15])); }
            if (MODE == 2) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "s"(c0), "v"(a[(i + 5) & 15])); }
            if (MODE == 5) { asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(v0), "v"(a[(i + 5) & 15])); }
            if (MODE == 6) { asm volatile("v_mov_b32 %0, %1" : "=v"(((int*)a)[i]) : "v"(((int*)a)[(i + 5) & 15])); }
            if (MODE == 7) { asm volatile("v_xor_b32 %0, %1, %2" : "=v"(((int*)a)[i]) : "s"(iters), "v"(((int*)a)[(i + 5) & 15])); }
            if (MODE == 3) { asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 5) & 15])); }
            if (MODE == 4) { asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[i]) : "s"(c0), "v"(a[(i + 5) & 15])); }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double* d_out, double* d_coef, int waves_per_simd) {
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_coef, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_coef, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = double(blocks) * 4 * iters * 16;
    const double per_simd = wave_instr / 1024.0;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD  (=%.2f cycles @2.4GHz)  %.1f TFLOP-equiv\n", name,
           waves_per_simd, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, wave_instr * 128 / (ms * 1e-3) / 1e12);
}

int main() {
    double *d_out, *d_coef;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(double));
    hipMalloc(&d_coef, 64);
    std::vector<double> h(256 * 8 * 256, 1.0);
    hipMemcpy(d_out, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    double c[2] = {0.999999, 1e-9};
    hipMemcpy(d_coef, c, 16, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4, 8}) {
        run<1>("v_fma_f64 vgpr operands", d_out, d_coef, w);
        run<2>("v_fma_f64 one sgpr operand", d_out, d_coef, w);
        run<3>("v_mov_b64", d_out, d_coef, w);
        run<4>("v_mul_f64 sgpr operand", d_out, d_coef, w);
        run<5>("v_fmac_f64", d_out, d_coef, w);
        run<6>("v_mov_b32", d_out, d_coef, w);
        run<7>("v_xor_b32 sgpr", d_out, d_coef, w);
    }
    return 0;
}
