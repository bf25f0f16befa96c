// copybench.hip -- which plain copy reaches the guide's 6.29 TB/s on this box?  (for ge_copy_bandwidth)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(v4 *__restrict__ dst, const v4 *__restrict__ src, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(r[u], dst + i + u * stride); else dst[i + u * stride] = r[u]; }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
template <int U, bool NT> double run(v4 *b, v4 *a, int64_t n4, int blocks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_copy<U, NT>), dim3(blocks), dim3(256), 0, 0, b, a, n4);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_copy<U, NT>), dim3(blocks), dim3(256), 0, 0, b, a, n4);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * n4 * 16 * 5 / (ms * 1e-3) / 1e12;
}
int main() {
    for (int64_t bytes : {256ll << 20, 1ll << 30, 2ll << 30}) {
        const int64_t n4 = bytes / 16;
        v4 *a, *b; (void)hipMalloc((void **)&a, bytes); (void)hipMalloc((void **)&b, bytes); (void)hipMemset(a, 1, bytes);
        for (int blocks : {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32})
            printf("bytes %lld blocks %d: U1 %.2f  U2 %.2f  U4 %.2f  U8 %.2f  U4nt %.2f  U8nt %.2f TB/s\n", (long long)bytes, blocks,
                   run<1, false>(b, a, n4, blocks), run<2, false>(b, a, n4, blocks), run<4, false>(b, a, n4, blocks), run<8, false>(b, a, n4, blocks),
                   run<4, true>(b, a, n4, blocks), run<8, true>(b, a, n4, blocks));
        (void)hipFree(a); (void)hipFree(b);
    }
    return 0;
}
