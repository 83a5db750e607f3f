// Issue-rate test of the compiler-generated u-type butterfly (14 FMAs per amplitude pair) on register-resident
// amplitudes: no memory traffic inside the timed loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename real> struct alignas(2 * sizeof(real)) cx { real re, im; };

template <typename real, int R, int J>
__device__ __forceinline__ void butterfly(cx<real> (&amp)[1 << R], const real (&m)[7]) {
    constexpr int tbit = 1 << J;
#pragma unroll
    for (int e0 = 0; e0 < (1 << R); ++e0) {
        if (e0 & tbit) continue;
        const real a0r = amp[e0].re, a0i = amp[e0].im, a1r = amp[e0 | tbit].re, a1i = amp[e0 | tbit].im;
        real u = m[1] * a1r, w = m[1] * a1i, p = m[5] * a1r, q = m[5] * a1i;
        u = fma(-m[2], a1i, u); w = fma(m[2], a1r, w); p = fma(-m[6], a1i, p); q = fma(m[6], a1r, q);
        p = fma(m[3], a0r, p); q = fma(m[3], a0i, q);
        amp[e0 | tbit].re = fma(-m[4], a0i, p); amp[e0 | tbit].im = fma(m[4], a0r, q);
        amp[e0].re = fma(m[0], a0r, u); amp[e0].im = fma(m[0], a0i, w);
    }
}

template <int R, int WPS>
__global__ void __launch_bounds__(256, WPS) k(cx<double>* __restrict__ data, const double* __restrict__ mats, int n_gates) {
    cx<double> amp[1 << R];
    const size_t base = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) << R;
#pragma unroll
    for (int e = 0; e < (1 << R); ++e) amp[e] = data[base + e];
    const double* mp = mats;
    for (int g = 0; g < n_gates; ++g, mp += 8) {
        const double mm[7] = {mp[0], mp[2], mp[3], mp[4], mp[5], mp[6], mp[7]};
        const int j = g % R;
        if (j == 0) butterfly<double, R, 0>(amp, mm);
        else if (j == 1) butterfly<double, R, 1>(amp, mm);
        else if (j == 2) butterfly<double, R, 2>(amp, mm);
        else if constexpr (R > 3) butterfly<double, R, 3>(amp, mm);
    }
#pragma unroll
    for (int e = 0; e < (1 << R); ++e) data[base + e] = amp[e];
}

template <int R, int WPS>
void run(cx<double>* d, double* m, int waves_per_simd) {
    const int n_gates = 3000;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<R, WPS>), dim3(blocks), dim3(256), 0, 0, d, m, 8);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, WPS>), dim3(blocks), dim3(256), 0, 0, d, m, n_gates);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fma_instr = double(blocks) * 4 * n_gates * (1 << (R - 1)) * 14;
    printf("R=%d cap=%d waves/SIMD launched=%d: %.3f ms, %.2f ns per FMA wave-instr per SIMD (%.2f cyc @2.4GHz), %.1f TFLOP\n", R, WPS,
           waves_per_simd, ms, ms * 1e6 / (fma_instr / 1024), ms * 1e6 / (fma_instr / 1024) * 2.4, fma_instr * 128 / (ms * 1e-3) / 1e12);
}

int main() {
    cx<double>* d; double* m;
    const size_t n = size_t(256) * 8 * 256 * 16;
    (void)hipMalloc(&d, n * sizeof(cx<double>));
    (void)hipMalloc(&m, 3008 * 8 * sizeof(double));
    std::vector<double> h(n * 2, 0.001), hm(3008 * 8);
    for (size_t i = 0; i < hm.size(); i += 8) { double c = 0.8, s = 0.6; hm[i]=c; hm[i+1]=0; hm[i+2]=-s*0.6; hm[i+3]=-s*0.8; hm[i+4]=s*0.8; hm[i+5]=s*0.6; hm[i+6]=c*0.28; hm[i+7]=c*0.96; }
    (void)hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(m, hm.data(), hm.size() * 8, hipMemcpyHostToDevice);
    run<3, 2>(d, m, 1); run<3, 2>(d, m, 2); run<3, 4>(d, m, 4); run<3, 6>(d, m, 6); run<3, 8>(d, m, 8);
    run<4, 2>(d, m, 1); run<4, 2>(d, m, 2); run<4, 4>(d, m, 4);
    return 0;
}
