// ge_api.hip -- error buffer, device selection, version: the parts of the C ABI that no kernel owns.
#include "ge_common.h"
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
__global__ __launch_bounds__(256) void k_copy16(float4 *__restrict__ dst, const float4 *__restrict__ src, int64_t n4) {
    // four 16-byte loads in flight per lane before the first store
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
}  // namespace

extern "C" {

// What this device's memory system gives a plain 16-byte-per-lane copy (bytes read + bytes written per second): the practical
// ceiling a streaming kernel is measured against (MI355X_MICROARCH.md: 6.29 TB/s), taken on the box the bench runs on.
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
    float ms = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_copy16, dim3(256 * 8), dim3(256), 0, 0, b, a, n4);                       // warm-up
        e = hipEventRecord(e0, 0);
        for (int r = 0; r < reps && e == hipSuccess; ++r) hipLaunchKernelGGL(k_copy16, dim3(256 * 8), dim3(256), 0, 0, b, a, n4);
        if (e == hipSuccess) e = hipEventRecord(e1, 0);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(a); if (b) (void)hipFree(b);
    if (e != hipSuccess) return ge::fail(GE_ERR_HIP, "ge_copy_bandwidth: %s", hipGetErrorString(e));
    *gbps = 2.0 * (double)n4 * 16.0 * reps / ((double)ms * 1e-3) / 1e9;
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
