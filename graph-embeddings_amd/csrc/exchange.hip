// exchange.hip -- the elementwise half of the multi-GPU context exchange (the reference is a single
// JVM and has no counterpart).  One pass over a replicated fp32 table per step instead of six library passes:
//
//   land:  table += wire - own        what the OTHER ranks contributed to the all-reduced delta sum in `wire`
//   take:  d = bf16(table - base);  wire = own = d;  base = table (after landing)
//
// 24 bytes per element (12 read, 12 written) for land+take.  HBM-bound streaming: 8 elements per lane and trip,
// 128-bit accesses, grid-stride over a grid sized to the device.
#include "ge_common.h"

namespace {

__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;       // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <bool LAND, bool TAKE>
__device__ __forceinline__ void turn_one(float &t, float &b, uint32_t &w, uint32_t &o) {
    float d = 0.0f;
    if (TAKE) d = t - b;                                   // this rank's moves since the last take
    if (LAND) {
        const float r = bf16_to_f32(w) - bf16_to_f32(o);
        t += r;
        if (!TAKE) b += r;                                 // not this rank's move: keep it out of the next delta
    }
    if (TAKE) { b = t; w = o = f32_to_bf16_rne(d); }
}

template <bool LAND, bool TAKE>
__global__ __launch_bounds__(256) void k_exchange_turn(float *__restrict__ table, float *__restrict__ base,
                                                       uint16_t *__restrict__ wire, uint16_t *__restrict__ own, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n8 = n >> 3;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n8; v += stride) {
        float4 t0 = reinterpret_cast<const float4 *>(table)[2 * v], t1 = reinterpret_cast<const float4 *>(table)[2 * v + 1];
        float4 b0 = reinterpret_cast<const float4 *>(base)[2 * v],  b1 = reinterpret_cast<const float4 *>(base)[2 * v + 1];
        uint4 w = LAND ? reinterpret_cast<const uint4 *>(wire)[v] : make_uint4(0, 0, 0, 0);
        uint4 o = LAND ? reinterpret_cast<const uint4 *>(own)[v]  : make_uint4(0, 0, 0, 0);
        float tt[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        uint32_t ww[4] = {w.x, w.y, w.z, w.w}, oo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t wl = ww[k] & 0xffffu, wh = ww[k] >> 16, ol = oo[k] & 0xffffu, oh = oo[k] >> 16;
            turn_one<LAND, TAKE>(tt[2 * k], bb[2 * k], wl, ol);
            turn_one<LAND, TAKE>(tt[2 * k + 1], bb[2 * k + 1], wh, oh);
            ww[k] = wl | (wh << 16); oo[k] = ol | (oh << 16);
        }
        if (LAND) {
            reinterpret_cast<float4 *>(table)[2 * v]     = make_float4(tt[0], tt[1], tt[2], tt[3]);
            reinterpret_cast<float4 *>(table)[2 * v + 1] = make_float4(tt[4], tt[5], tt[6], tt[7]);
        }
        reinterpret_cast<float4 *>(base)[2 * v]     = make_float4(bb[0], bb[1], bb[2], bb[3]);
        reinterpret_cast<float4 *>(base)[2 * v + 1] = make_float4(bb[4], bb[5], bb[6], bb[7]);
        if (TAKE) {
            reinterpret_cast<uint4 *>(wire)[v] = make_uint4(ww[0], ww[1], ww[2], ww[3]);
            reinterpret_cast<uint4 *>(own)[v]  = make_uint4(oo[0], oo[1], oo[2], oo[3]);
        }
    }
    // ragged tail (n % 8 elements), one lane each
    const int64_t i = (n8 << 3) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float t = table[i], b = base[i];
        uint32_t w = LAND ? wire[i] : 0u, o = LAND ? own[i] : 0u;
        turn_one<LAND, TAKE>(t, b, w, o);
        if (LAND) table[i] = t;
        base[i] = b;
        if (TAKE) { wire[i] = (uint16_t)w; own[i] = (uint16_t)o; }
    }
}

}  // namespace

extern "C" ge_status ge_exchange_turn(float *table, float *base, uint16_t *wire, uint16_t *own, int64_t count,
                                      int32_t land, int32_t take, void *stream) {
    if (!table || !base || !wire || !own) return ge::fail(GE_ERR_ARG, "ge_exchange_turn: null pointer");
    if (count < 0) return ge::fail(GE_ERR_ARG, "ge_exchange_turn: negative count");
    if (((uintptr_t)table | (uintptr_t)base) % 16 || ((uintptr_t)wire | (uintptr_t)own) % 16)
        return ge::fail(GE_ERR_ARG, "ge_exchange_turn: buffers must be 16-byte aligned");
    if (!land && !take) return GE_OK;
    if (count == 0) return GE_OK;
    int dev = 0, cus = 256;
    GE_HIP(hipGetDevice(&dev));
    GE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t want = ((count >> 3) + 255) / 256;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)cus * 8));
    hipStream_t s = (hipStream_t)stream;
    if (land && take)  hipLaunchKernelGGL((k_exchange_turn<true, true>),  dim3(blocks), dim3(256), 0, s, table, base, wire, own, count);
    else if (land)     hipLaunchKernelGGL((k_exchange_turn<true, false>), dim3(blocks), dim3(256), 0, s, table, base, wire, own, count);
    else               hipLaunchKernelGGL((k_exchange_turn<false, true>), dim3(blocks), dim3(256), 0, s, table, base, wire, own, count);
    GE_HIP(hipGetLastError());
    return GE_OK;
}
