// exchange.hip -- the elementwise half of the multi-GPU context exchange for bf16 rows (the reference is a
// single JVM and has no counterpart; the fp32 tables' pass and the all-reduce are in sync.hip).
//
//   land:  value += wire - own;  base += wire - own     what the OTHER ranks contributed to the all-reduced sum in `wire`
//   take:  d = bf16(value - base) (before landing);  own = d;  base += d      (`wire` is the all-reduce's receive buffer:
//          the caller sums own -> wire out of place, or copies own into wire and sums in place)
//
// `base` is always  consensus + this rank's deltas in flight  (consensus = start + every landed sum, the same on all
// ranks): what bf16 drops from a delta stays in value - base and goes out with the next one, so the replicas differ only
// by what is in flight plus one rounding, however long the run (error feedback).  HBM-bound streaming.
#include "ge_common.h"

namespace {

__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;       // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// The same turn for a context table stored as bf16 with fp32 master rows for the hub columns (GE_DTYPE_BF16,
// BASELINE config C5).  A row's value lives in hub_rows[hub_index[v]] (fp32) when the column is a hub ON THIS RANK
// (hub sets differ per rank: each sees its own shard), else in the bf16 table; the wire carries every row either way.
// `base` is fp32 for every row (consensus + own deltas in flight, see above), so the stochastic rounding with which a
// landed ordinary row is re-narrowed -- as the update kernel does -- shows up in table - base and is fed back with
// the next delta instead of accumulating.  4 elements per lane (dim % 4 == 0).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; x *= 0xC2B2AE3Du; x ^= x >> 16;
    return x;
}

template <bool LAND, bool TAKE>
__global__ __launch_bounds__(256) void k_exchange_turn_bf16(uint16_t *__restrict__ table, int64_t stride4, float *__restrict__ hub_rows,
                                                            const int32_t *__restrict__ hub_index, int32_t D4, int64_t n4,
                                                            float *__restrict__ base, uint16_t *__restrict__ wire, uint16_t *__restrict__ own,
                                                            uint32_t seed) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
        const int32_t v = (int32_t)(q / D4);
        const int32_t c4 = (int32_t)(q - (int64_t)v * D4);
        const int32_t hub = hub_index[v];
        float t[4];
        if (hub >= 0) {
            const float4 tv = reinterpret_cast<const float4 *>(hub_rows)[(int64_t)hub * D4 + c4];
            t[0] = tv.x; t[1] = tv.y; t[2] = tv.z; t[3] = tv.w;
        } else {
            const uint2 tv = reinterpret_cast<const uint2 *>(table)[(int64_t)v * stride4 + c4];
            t[0] = bf16_to_f32(tv.x & 0xffffu); t[1] = bf16_to_f32(tv.x >> 16); t[2] = bf16_to_f32(tv.y & 0xffffu); t[3] = bf16_to_f32(tv.y >> 16);
        }
        const float4 bv = reinterpret_cast<const float4 *>(base)[q];
        float b[4] = {bv.x, bv.y, bv.z, bv.w};
        uint32_t w[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};
        if (LAND) {
            const uint2 wv = reinterpret_cast<const uint2 *>(wire)[q], ov = reinterpret_cast<const uint2 *>(own)[q];
            w[0] = wv.x & 0xffffu; w[1] = wv.x >> 16; w[2] = wv.y & 0xffffu; w[3] = wv.y >> 16;
            o[0] = ov.x & 0xffffu; o[1] = ov.x >> 16; o[2] = ov.y & 0xffffu; o[3] = ov.y >> 16;
        }
        uint32_t t16[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = TAKE ? t[k] - b[k] : 0.0f;
            if (LAND) {
                const float r = bf16_to_f32(w[k]) - bf16_to_f32(o[k]);
                const float tn = t[k] + r;
                b[k] += r;
                t[k] = tn;
                if (hub < 0) {                                       // stored as bf16: 16 random low bits, truncate
                    const uint32_t rnd = mix32((uint32_t)(q * 4 + k) * 0x9E3779B1u + seed) >> 16;
                    const uint32_t bits = __float_as_uint(tn);
                    t16[k] = ((bits & 0x7f800000u) == 0x7f800000u) ? bits >> 16 : (bits + rnd) >> 16;   // inf / NaN pass through
                }
            }
            if (TAKE) { w[k] = o[k] = f32_to_bf16_rne(d); b[k] += bf16_to_f32(w[k]); }
        }
        if (LAND) {
            if (hub >= 0) reinterpret_cast<float4 *>(hub_rows)[(int64_t)hub * D4 + c4] = make_float4(t[0], t[1], t[2], t[3]);
            else reinterpret_cast<uint2 *>(table)[(int64_t)v * stride4 + c4] = make_uint2(t16[0] | (t16[1] << 16), t16[2] | (t16[3] << 16));
        }
        reinterpret_cast<float4 *>(base)[q] = make_float4(b[0], b[1], b[2], b[3]);
        if (TAKE) reinterpret_cast<uint2 *>(own)[q] = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
    }
}

}  // namespace

extern "C" ge_status ge_exchange_turn_bf16(uint16_t *table, int32_t row_stride, float *hub_rows, const int32_t *hub_index, int32_t vocab_size, int32_t dim,
                                           float *base, uint16_t *wire, uint16_t *own, int32_t land, int32_t take, uint32_t seed, void *stream) {
    if (!table || !base || !hub_index || !wire || !own) return ge::fail(GE_ERR_ARG, "ge_exchange_turn_bf16: null pointer");
    if (vocab_size < 0 || dim <= 0 || dim % 4 != 0) return ge::fail(GE_ERR_ARG, "ge_exchange_turn_bf16: dim must be a positive multiple of 4");
    if (row_stride < dim || row_stride % 4 != 0) return ge::fail(GE_ERR_ARG, "ge_exchange_turn_bf16: row_stride must be a multiple of 4 and >= dim");
    if (((uintptr_t)table | (uintptr_t)wire | (uintptr_t)own) % 8 || ((uintptr_t)hub_rows | (uintptr_t)base) % 16)
        return ge::fail(GE_ERR_ARG, "ge_exchange_turn_bf16: misaligned buffer");
    if ((!land && !take) || vocab_size == 0) return GE_OK;
    int dev = 0, cus = 256;
    GE_HIP(hipGetDevice(&dev));
    GE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t n4 = (int64_t)vocab_size * (dim / 4);
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n4 + 255) / 256, (int64_t)cus * 8));
    hipStream_t s = (hipStream_t)stream;
    if (land && take)  hipLaunchKernelGGL((k_exchange_turn_bf16<true, true>), dim3(blocks), dim3(256), 0, s, table, (int64_t)(row_stride / 4), hub_rows, hub_index, dim / 4, n4, base, wire, own, seed);
    else if (land)     hipLaunchKernelGGL((k_exchange_turn_bf16<true, false>), dim3(blocks), dim3(256), 0, s, table, (int64_t)(row_stride / 4), hub_rows, hub_index, dim / 4, n4, base, wire, own, seed);
    else               hipLaunchKernelGGL((k_exchange_turn_bf16<false, true>), dim3(blocks), dim3(256), 0, s, table, (int64_t)(row_stride / 4), hub_rows, hub_index, dim / 4, n4, base, wire, own, seed);
    GE_HIP(hipGetLastError());
    return GE_OK;
}

