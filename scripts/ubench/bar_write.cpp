// Can the host store straight into device memory (fine-grained allocation, large BAR), and does a kernel launched
// afterwards see the data?  (Would take the PCIe round trips out of prepare's descriptor and parameter reads.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

__global__ void sum_kernel(const double* in, int n, double* out) {
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) *out = s;
}

int main() {
    double* dev = nullptr;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&dev), 1 << 16, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    hipPointerAttribute_t attr;
    e = hipPointerGetAttributes(&attr, dev);
    printf("attributes: %s type=%d host=%p device=%p\n", hipGetErrorString(e), int(attr.type), attr.hostPointer, attr.devicePointer);
    double* out;
    (void)hipHostMalloc(reinterpret_cast<void**>(&out), 64);
    const int n = 64;
    printf("writing from the host...\n");
    fflush(stdout);
    for (int i = 0; i < n; ++i) dev[i] = i + 1;  // (faults here if the memory is not host accessible)
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(64), 0, 0, dev, n, out);
    (void)hipDeviceSynchronize();
    printf("sum = %.1f (expected %.1f)\n", *out, n * (n + 1) / 2.0);
    // latency: host store -> kernel sees it, against the same with pinned host memory
    double* pinned;
    (void)hipHostMalloc(reinterpret_cast<void**>(&pinned), 1 << 16);
    for (const char* which : {"device (BAR)", "pinned host"}) {
        double* buf = which[0] == 'd' ? dev : pinned;
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < 200; ++rep) {
            for (int i = 0; i < n; ++i) buf[i] = rep + i;
            hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(64), 0, 0, buf, n, out);
            (void)hipDeviceSynchronize();
        }
        printf("%s: %.1f us per store + launch + sync\n", which, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200);
    }
    return 0;
}
