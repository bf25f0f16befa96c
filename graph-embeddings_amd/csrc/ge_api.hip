// ge_api.hip -- error buffer, device selection, version: the parts of the C ABI that no kernel owns.
#include "ge_common.h"
#include <algorithm>
#include <cstring>

namespace ge {

char *last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

ge_status select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(GE_ERR_HIP, "no HIP device available (%s); libgeglove has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= n) return fail(GE_ERR_ARG, "device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(GE_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GE_ERR_HIP, "device %d is %s; libgeglove is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(GE_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    return GE_OK;
}

}  // namespace ge

namespace {
template <int U>
__global__ __launch_bounds__(256) void k_copy16(float4 *__restrict__ dst, const float4 *__restrict__ src, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * stride] = r[u];
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
}  // namespace

extern "C" {

// What this device's memory system gives a plain 16-byte-per-lane copy (bytes read + bytes written per second): the practical
// ceiling a streaming kernel is measured against, taken on the box the bench runs on.  The best of a few launch shapes
// (tools/micro/copybench.hip: 2 or 4 workgroups per CU, one or two loads in flight per lane; 4.4 - 5.8 TB/s on the boxes of this
// pool, against the 6.29 TB/s MI355X_MICROARCH.md quotes).
ge_status ge_copy_bandwidth(int32_t device, int64_t bytes, int32_t reps, double *gbps) {
    if (!gbps || bytes < (1 << 20) || reps < 1) return ge::fail(GE_ERR_ARG, "ge_copy_bandwidth: need >= 1 MiB, >= 1 repetition and an output");
    ge_status st = ge::select_device(device);
    if (st != GE_OK) return st;
    float4 *a = nullptr, *b = nullptr;
    const int64_t n4 = bytes / 16;
    GE_HIP(hipMalloc((void **)&a, (size_t)n4 * 16));
    hipError_t e = hipMalloc((void **)&b, (size_t)n4 * 16);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e == hipSuccess) e = hipMemset(a, 1, (size_t)n4 * 16);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double best = 0;
    for (int shape = 0; shape < 4 && e == hipSuccess; ++shape) {
        const int blocks = (shape & 1) ? 1024 : 512;
        auto launch = [&]() {
            if (shape < 2) hipLaunchKernelGGL(k_copy16<1>, dim3(blocks), dim3(256), 0, 0, b, a, n4);
            else hipLaunchKernelGGL(k_copy16<2>, dim3(blocks), dim3(256), 0, 0, b, a, n4);
        };
        launch();                                                       // warm-up
        e = hipEventRecord(e0, 0);
        for (int r = 0; r < reps && e == hipSuccess; ++r) launch();
        if (e == hipSuccess) e = hipEventRecord(e1, 0);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && ms > 0) best = std::max(best, 2.0 * (double)n4 * 16.0 * reps / ((double)ms * 1e-3) / 1e9);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(a); if (b) (void)hipFree(b);
    if (e != hipSuccess) return ge::fail(GE_ERR_HIP, "ge_copy_bandwidth: %s", hipGetErrorString(e));
    *gbps = best;
    return GE_OK;
}

const char *ge_last_error(void) { return ge::last_error_buf(); }

const char *ge_version(void) { return "geglove 0.1.0 (gfx950)"; }

int32_t ge_glove_cfg_size(void) { return (int32_t)sizeof(ge_glove_cfg); }
int32_t ge_bca_cfg_size(void) { return (int32_t)sizeof(ge_bca_cfg); }

int32_t ge_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { ge::fail(GE_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); return -1; }
    int good = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++good;
    }
    return good;
}

}  // extern "C"
