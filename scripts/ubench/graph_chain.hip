// Three short dependent kernels per step (the shape of a split-evaluation step: 128 / 512 / 64 workgroups), launched
// one by one on a stream or as one instantiated hipGraph: wall time per step including the final synchronisation.
// build: hipcc --offload-arch=gfx950 -O2 graph_chain.hip -o graph_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin(double* p, int iterations) {
    double v = p[blockIdx.x];
    for (int i = 0; i < iterations; ++i) v = v * 1.0000001 + 1e-9;
    p[blockIdx.x] = v;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    double* d = nullptr;
    CHECK(hipMalloc(&d, 4096 * sizeof(double)));
    CHECK(hipMemset(d, 0, 4096 * sizeof(double)));
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int steps = 2000;
    for (int spin_iters : {200, 2000, 6000}) {
        auto launch_all = [&]() {
            hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, d, spin_iters * 3);
            hipLaunchKernelGGL(spin, dim3(512), dim3(256), 0, s, d, spin_iters);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, spin_iters);
        };
        for (int i = 0; i < 50; ++i) { launch_all(); CHECK(hipStreamSynchronize(s)); }
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) { launch_all(); CHECK(hipStreamSynchronize(s)); }
        const double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
        hipGraph_t graph;
        hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        launch_all();
        CHECK(hipStreamEndCapture(s, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int i = 0; i < 50; ++i) { CHECK(hipGraphLaunch(exec, s)); CHECK(hipStreamSynchronize(s)); }
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) { CHECK(hipGraphLaunch(exec, s)); CHECK(hipStreamSynchronize(s)); }
        const double graphed = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
        // one kernel alone, for scale
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) { hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, d, spin_iters * 3); CHECK(hipStreamSynchronize(s)); }
        const double one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
        printf("spin %5d: three launches %.1f us per step, as a graph %.1f us, the first kernel alone %.1f us\n", spin_iters, plain, graphed, one);
        (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
    }
    return 0;
}
