// How much earlier than hipStreamSynchronize does the host see a value the kernel's last workgroups write to pinned host
// memory?  (The result buffer of the evaluator is such memory.)
// build: hipcc --offload-arch=gfx950 -O2 flag_vs_sync.hip -o flag_vs_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>

__global__ void work(double* scratch, volatile double* results, int iterations, double value) {
    double v = scratch[blockIdx.x];
    for (int i = 0; i < iterations; ++i) v = v * 1.0000001 + 1e-9;
    scratch[blockIdx.x] = v;
    if (threadIdx.x == 0) results[blockIdx.x] = value;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const int n = 64, steps = 2000;
    double *d = nullptr, *h = nullptr;
    CHECK(hipMalloc(&d, n * sizeof(double)));
    CHECK(hipMemset(d, 0, n * sizeof(double)));
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&h), n * sizeof(double), hipHostMallocDefault));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    using clk = std::chrono::steady_clock;
    for (int iters : {500, 5000}) {
        double t_sync = 0, t_flag = 0, t_after = 0;
        for (int i = 0; i < steps + 50; ++i) {
            for (int k = 0; k < n; ++k) h[k] = NAN;
            auto t0 = clk::now();
            hipLaunchKernelGGL(work, dim3(n), dim3(256), 0, s, d, h, iters, double(i));
            if (i & 1) {
                CHECK(hipStreamSynchronize(s));
                if (i >= 50) t_sync += std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            } else {
                volatile double* v = h;
                for (;;) {
                    bool all = true;
                    for (int k = 0; k < n; ++k) all = all && !std::isnan(v[k]);
                    if (all) break;
                }
                auto t1 = clk::now();
                CHECK(hipStreamSynchronize(s));
                if (i >= 50) {
                    t_flag += std::chrono::duration<double, std::micro>(t1 - t0).count();
                    t_after += std::chrono::duration<double, std::micro>(clk::now() - t1).count();
                }
            }
        }
        printf("iterations %5d: launch -> hipStreamSynchronize returns %.1f us; launch -> all results visible %.1f us (+ %.1f us until the synchronisation that follows returns)\n",
               iters, t_sync / (steps / 2), t_flag / (steps / 2), t_after / (steps / 2));
    }
    return 0;
}
