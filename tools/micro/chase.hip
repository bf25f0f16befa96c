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

// Per-XCD streaming rate: one 256-thread workgroup per CU reads its own 16 MiB slice of `buf` with 16-byte loads, `reps` times;
// out[3*b .. ] = {XCC id, shader cycles, bytes} of workgroup b.  All workgroups run at once (launch with <= one per CU).
__global__ __launch_bounds__(256) void k_xcd_stream(const float4 *buf, int64_t slice4, int reps, unsigned long long *out) {
    const float4 *p = buf + (int64_t)blockIdx.x * slice4;
    float4 acc = make_float4(0, 0, 0, 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
        for (int64_t k = threadIdx.x; k < slice4; k += 256) {
            const float4 v = p[k];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[3 * blockIdx.x] = xcc & 0xf;
        out[3 * blockIdx.x + 1] = t1 - t0;
        out[3 * blockIdx.x + 2] = (unsigned long long)reps * slice4 * 16 + (acc.x + acc.y + acc.z + acc.w == 12345.678f ? 1 : 0);
    }
}

// fills out[8] with GB/s per XCD (bytes of its workgroups / the slowest workgroup's time at 100 MHz memtime ticks... the caller
// passes the tick rate), returns the number of workgroups; buf must hold blocks * 16 MiB
extern "C" int xcd_stream(const void *buf, int blocks, int reps, double *gbps_per_xcd, double *cycles_min_max) {
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(unsigned long long) * 3 * blocks) != hipSuccess) return -1;
    const int64_t slice4 = (16ll << 20) / 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_xcd_stream, dim3(blocks), dim3(256), 0, 0, (const float4 *)buf, slice4, reps, d);
    hipEventRecord(e1, 0);
    unsigned long long *h = new unsigned long long[3 * blocks];
    int rc = -1;
    if (hipMemcpy(h, d, sizeof(unsigned long long) * 3 * blocks, hipMemcpyDeviceToHost) == hipSuccess) {
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        double bytes[8] = {0}, tmax[8] = {0}; unsigned long long cmin = ~0ull, cmax = 0, call = 0;
        for (int b = 0; b < blocks; ++b) {
            const int x = (int)h[3 * b] & 7;
            bytes[x] += (double)h[3 * b + 2];
            if ((double)h[3 * b + 1] > tmax[x]) tmax[x] = (double)h[3 * b + 1];
            cmin = h[3 * b + 1] < cmin ? h[3 * b + 1] : cmin; cmax = h[3 * b + 1] > cmax ? h[3 * b + 1] : cmax; call = cmax;
        }
        // ticks -> seconds: the whole kernel took `ms`; the slowest workgroup's ticks span (almost) all of it
        const double tick_s = call ? (ms * 1e-3) / (double)call : 0;
        for (int x = 0; x < 8; ++x) gbps_per_xcd[x] = tmax[x] > 0 ? bytes[x] / (tmax[x] * tick_s) / 1e9 : 0;
        cycles_min_max[0] = (double)cmin; cycles_min_max[1] = (double)cmax; cycles_min_max[2] = ms;
        rc = blocks;
    }
    delete[] h; (void)hipFree(d); hipEventDestroy(e0); hipEventDestroy(e1);
    return rc;
}
