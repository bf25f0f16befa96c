// probe.hip -- a stand-in for a collective's kernel: a FIXED, small number of workgroups that stream `n` floats from src
// to dst (grid-stride), launched on a caller-given stream.  Used by tools/overlap_probe.py to see whether such a kernel
// runs beside the persistent epoch kernel or has to wait for it.  Diagnostic only; not part of libgeglove.so.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(256) void k_probe_copy(float4 *__restrict__ dst, const float4 *__restrict__ src, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

extern "C" int probe_copy(float *dst, const float *src, int64_t n, int blocks, void *stream) {
    hipLaunchKernelGGL(k_probe_copy, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4 *>(dst), reinterpret_cast<const float4 *>(src), n / 4);
    return (int)hipGetLastError();
}
