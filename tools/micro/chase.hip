// chase.hip -- dependent-load latency over a region of device memory (round 2: what differs between a fast and a slow process?).
// One lane per wavefront walks `steps` rows of `row_bytes` bytes picked by an LCG whose next state depends on the loaded word
// (times zero), so every load waits for the previous one: cycles per load = latency of one random row access (sc1, like the
// trainer's).  `waves` wavefronts run at once (1 = idle-chip latency).  Diagnostic only; not part of libgeglove.so.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(64) void k_chase(const char *base, int64_t rows, int64_t row_bytes, int32_t steps, unsigned long long *cycles) {
    if ((threadIdx.x & 63) != 0) return;
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x + 1);
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int32_t k = 0; k < steps; ++k) {
        s = s * 6364136223846793005ull + 1442695040888963407ull + (acc & 0u);
        const int64_t r = (int64_t)((s >> 24) % (uint64_t)rows);
        const uint32_t v = __hip_atomic_load(reinterpret_cast<const uint32_t *>(base + r * row_bytes), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += v;
        s += (uint64_t)(v & 0u);              // the dependency the compiler cannot remove
        asm volatile("" : "+v"(acc));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    cycles[blockIdx.x] = (t1 - t0) + (acc & 0u);
}

extern "C" double chase(const void *base, int64_t rows, int64_t row_bytes, int steps, int waves) {
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(unsigned long long) * waves) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_chase, dim3(waves), dim3(64), 0, 0, (const char *)base, rows, row_bytes, steps, d);
    unsigned long long *h = new unsigned long long[waves];
    double out = -1;
    if (hipMemcpy(h, d, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost) == hipSuccess) {
        double sum = 0; for (int i = 0; i < waves; ++i) sum += (double)h[i];
        out = sum / waves / steps;
    }
    delete[] h; (void)hipFree(d);
    return out;
}
