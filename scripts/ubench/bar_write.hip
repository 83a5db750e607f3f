// Can the host write a device-resident buffer directly (large BAR), and how long after such writes does a kernel see them?
// build: hipcc --offload-arch=gfx950 -O2 bar_write.hip -o bar_write
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <csetjmp>
#include <csignal>
#include <immintrin.h>

__global__ void reader(const double* src, double* out, int n, int reps) {
    // one wave: a dependent chain of reads (latency), then a sum of everything (what prepare_eval does with its parameters)
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += src[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

static sigjmp_buf jump;
static void on_segv(int) { siglongjmp(jump, 1); }

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    int large_bar = -1;
    CHECK(hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0));
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    const int n = 4096;  // 32 KiB of parameters
    double *d_plain = nullptr, *d_fine = nullptr, *h_pinned = nullptr, *d_out = nullptr, *h_out = nullptr;
    CHECK(hipMalloc(&d_plain, n * sizeof(double)));
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&d_fine), n * sizeof(double), hipDeviceMallocFinegrained) != hipSuccess) d_fine = nullptr;
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_pinned), n * sizeof(double), hipHostMallocDefault));
    CHECK(hipMalloc(&d_out, 64 * sizeof(double)));
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_out), 64 * sizeof(double), hipHostMallocDefault));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    struct Target { const char* name; double* p; } targets[] = {{"pinned host", h_pinned}, {"hipMalloc", d_plain}, {"hipExtMallocWithFlags(finegrained)", d_fine}};
    using clk = std::chrono::steady_clock;
    for (const Target& t : targets) {
        if (!t.p) { printf("%s: allocation failed\n", t.name); continue; }
        if (sigsetjmp(jump, 1)) { printf("%s: host access faults\n", t.name); continue; }
        volatile double* v = t.p;
        v[0] = 1.0;  // (faults here if the host cannot reach it)
        int wrong = 0;
        double t_write = 0, t_step = 0;
        const int steps = 2000;
        for (int i = 0; i < steps + 20; ++i) {
            auto t0 = clk::now();
            for (int k = 0; k < n; ++k) t.p[k] = double(i + k);
            _mm_sfence();
            auto t1 = clk::now();
            hipLaunchKernelGGL(reader, dim3(1), dim3(64), 0, s, t.p, h_out, n, 1);
            CHECK(hipStreamSynchronize(s));
            auto t2 = clk::now();
            const double want = double(n) * i + double(n) * (n - 1) / 2;
            if (h_out[0] != want) ++wrong;
            if (i >= 20) {
                t_write += std::chrono::duration<double, std::micro>(t1 - t0).count();
                t_step += std::chrono::duration<double, std::micro>(t2 - t1).count();
            }
        }
        printf("%-36s host writes 32 KiB in %.2f us; launch + read + sync %.2f us; wrong sums %d of %d\n", t.name, t_write / steps, t_step / steps, wrong, steps + 20);
    }
    // how the copy is made matters on a write-combining mapping: memcpy of a parameter block from cached host memory
    {
        static double src[4096];
        for (int k = 0; k < 4096; ++k) src[k] = k;
        for (const Target& t : targets) {
            if (!t.p || t.p == nullptr) continue;
            for (size_t bytes : {size_t(2048), size_t(30720)}) {
                auto t0 = clk::now();
                for (int i = 0; i < 2000; ++i) {
                    std::memcpy(t.p, src, bytes);
                    _mm_sfence();
                }
                const double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / 2000;
                printf("%-36s memcpy of %5zu bytes + sfence: %.2f us\n", t.name, bytes, us);
            }
        }
    }
    return 0;
}
