// Order of the basis states by the value of a diagonal operator: what the exact-probability CVaR (kernels.hpp:
// launch_cvar_exact) walks.  Once per operator, off the hot path: a library radix sort (hipCUB) of (value, index) pairs --
// stable, so states of equal value stay in index order, the order Python's sort leaves them in
// (reference: queasars/circuit_evaluation/expectation_calculation.py:16-17).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>

#include "kernels.hpp"

namespace qsv {

namespace {
__global__ void __launch_bounds__(256) iota_kernel(uint32_t* __restrict__ out, uint64_t dim) {
    const uint64_t stride = uint64_t(gridDim.x) * 256;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < dim; i += stride) out[i] = uint32_t(i);
}
// (-0.0 sorts below +0.0 in a radix sort and equal to it in a comparison sort: one representation)
__global__ void __launch_bounds__(256) canonical_zero_kernel(const double* __restrict__ in, double* __restrict__ out, uint64_t dim) {
    const uint64_t stride = uint64_t(gridDim.x) * 256;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < dim; i += stride) out[i] = in[i] + 0.0;
}
}  // namespace

hipError_t sort_states_by_value(const double* values, uint64_t dim, uint32_t* order, double* sorted_values, hipStream_t stream) {
    if (dim == 0 || dim > (uint64_t(1) << 30)) return hipErrorInvalidValue;
    const unsigned blocks = unsigned(std::min<uint64_t>((dim + 255) / 256, 8192));
    uint32_t* index = nullptr;
    double* keys = nullptr;
    void* temp = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&index), dim * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&keys), dim * sizeof(double));
    size_t temp_bytes = 0;
    if (e == hipSuccess)
        e = hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys, sorted_values, index, order, int(dim), 0, 64, stream);
    if (e == hipSuccess) e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(iota_kernel, dim3(blocks), dim3(256), 0, stream, index, dim);
        hipLaunchKernelGGL(canonical_zero_kernel, dim3(blocks), dim3(256), 0, stream, values, keys, dim);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, sorted_values, index, order, int(dim), 0, 64, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (temp) (void)hipFree(temp);
    if (keys) (void)hipFree(keys);
    if (index) (void)hipFree(index);
    return e;
}

}  // namespace qsv
