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

extern "C" {

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
