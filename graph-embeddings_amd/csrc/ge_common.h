// ge_common.h -- status/error plumbing shared by the HIP translation units of libgeglove.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <exception>
#include <new>
#include "../../include/geglove.h"

namespace ge {

char *last_error_buf();               // thread-local, 512 bytes (ge_api.hip)
inline ge_status fail(ge_status code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define GE_HIP(expr)                                                                     \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess)                                                            \
            return ge::fail(_e == hipErrorOutOfMemory ? GE_ERR_OOM : GE_ERR_HIP,         \
                            "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                            __FILE__, __LINE__);                                         \
    } while (0)

// No C++ exception crosses the C ABI: entry points whose bodies allocate on the host run as  GE_GUARD(name_impl(args)).
#define GE_GUARD(call)                                                                                     \
    try { return (call); }                                                                                  \
    catch (const std::bad_alloc &) { return ge::fail(GE_ERR_OOM, "host allocation failed"); }              \
    catch (const std::exception &e_) { return ge::fail(GE_ERR_STATE, "internal error: %s", e_.what()); }   \
    catch (...) { return ge::fail(GE_ERR_STATE, "internal error"); }

// Selects the device and verifies it is gfx950 (there is no fallback path).
ge_status select_device(int device);

}  // namespace ge
