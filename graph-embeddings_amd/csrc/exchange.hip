// exchange.hip -- the elementwise half of the multi-GPU context exchange (the reference is a single
// JVM and has no counterpart).  One pass over a replicated fp32 table per step instead of six library passes:
//
//   land:  table += wire - own;  base += wire - own     what the OTHER ranks contributed to the all-reduced sum in `wire`
//   take:  d = bf16(table - base) (before landing);  wire = own = d;  base += d
//
// `base` is therefore always  consensus + this rank's deltas in flight  (consensus = start + every landed sum, the same
// on all ranks): what bf16 drops from a delta stays in table - base and goes out with the next one, so the replicas
// differ only by what is in flight plus one rounding, however long the run (error feedback).
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
    if (TAKE) d = t - b;                                   // this rank's moves the others have not been sent yet
    if (LAND) {
        const float r = bf16_to_f32(w) - bf16_to_f32(o);
        t += r;
        b += r;                                            // not this rank's move: keep it out of the next delta
    }
    if (TAKE) { w = o = f32_to_bf16_rne(d); b += bf16_to_f32(w); }   // the base advances by what is SENT: the rounding residual stays in t - b
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

// The same turn over the ROW part of a table of fat rows (fp32 Hogwild layout: row_stride = dim + 4 floats, the row's
// bias at [dim], padding behind it): columns >= cols are left alone (the bias follows another merge rule, the padding
// stays zero) and their wire / own slots are written as zero on a take so the all-reduce can run over the whole buffer.
template <bool LAND, bool TAKE, int G>      // G = elements per lane: 4 (row_stride and cols multiples of 4) or 1
__global__ __launch_bounds__(256) void k_exchange_turn_rows(float *__restrict__ table, float *__restrict__ base,
                                                            uint16_t *__restrict__ wire, uint16_t *__restrict__ own,
                                                            int64_t n_groups, int32_t stride_g, int32_t cols_g) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += step) {
        const int32_t c = (int32_t)(g % stride_g);
        if (c >= cols_g) {
            if (TAKE) {
                if (G == 4) { reinterpret_cast<uint2 *>(wire)[g] = make_uint2(0, 0); reinterpret_cast<uint2 *>(own)[g] = make_uint2(0, 0); }
                else { wire[g] = 0; own[g] = 0; }
            }
            continue;
        }
        if (G == 4) {
            float4 tv = reinterpret_cast<const float4 *>(table)[g], bv = reinterpret_cast<const float4 *>(base)[g];
            uint2 wv = LAND ? reinterpret_cast<const uint2 *>(wire)[g] : make_uint2(0, 0), ov = LAND ? reinterpret_cast<const uint2 *>(own)[g] : make_uint2(0, 0);
            float t[4] = {tv.x, tv.y, tv.z, tv.w}, b[4] = {bv.x, bv.y, bv.z, bv.w};
            uint32_t w[4] = {wv.x & 0xffffu, wv.x >> 16, wv.y & 0xffffu, wv.y >> 16}, o[4] = {ov.x & 0xffffu, ov.x >> 16, ov.y & 0xffffu, ov.y >> 16};
#pragma unroll
            for (int k = 0; k < 4; ++k) turn_one<LAND, TAKE>(t[k], b[k], w[k], o[k]);
            if (LAND) reinterpret_cast<float4 *>(table)[g] = make_float4(t[0], t[1], t[2], t[3]);
            reinterpret_cast<float4 *>(base)[g] = make_float4(b[0], b[1], b[2], b[3]);
            if (TAKE) {
                reinterpret_cast<uint2 *>(wire)[g] = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
                reinterpret_cast<uint2 *>(own)[g]  = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
            }
        } else {
            float t = table[g], b = base[g];
            uint32_t w = LAND ? wire[g] : 0u, o = LAND ? own[g] : 0u;
            turn_one<LAND, TAKE>(t, b, w, o);
            if (LAND) table[g] = t;
            base[g] = b;
            if (TAKE) { wire[g] = (uint16_t)w; own[g] = (uint16_t)o; }
        }
    }
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
__global__ __launch_bounds__(256) void k_exchange_turn_bf16(uint16_t *__restrict__ table, float *__restrict__ hub_rows,
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
            const uint2 tv = reinterpret_cast<const uint2 *>(table)[q];
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
            else reinterpret_cast<uint2 *>(table)[q] = make_uint2(t16[0] | (t16[1] << 16), t16[2] | (t16[3] << 16));
        }
        reinterpret_cast<float4 *>(base)[q] = make_float4(b[0], b[1], b[2], b[3]);
        if (TAKE) {
            reinterpret_cast<uint2 *>(wire)[q] = make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
            reinterpret_cast<uint2 *>(own)[q]  = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
        }
    }
}

}  // namespace

extern "C" ge_status ge_exchange_turn_bf16(uint16_t *table, float *hub_rows, const int32_t *hub_index, int32_t vocab_size, int32_t dim,
                                           float *base, uint16_t *wire, uint16_t *own, int32_t land, int32_t take, uint32_t seed, void *stream) {
    if (!table || !base || !hub_index || !wire || !own) return ge::fail(GE_ERR_ARG, "ge_exchange_turn_bf16: null pointer");
    if (vocab_size < 0 || dim <= 0 || dim % 4 != 0) return ge::fail(GE_ERR_ARG, "ge_exchange_turn_bf16: dim must be a positive multiple of 4");
    if (((uintptr_t)table | (uintptr_t)wire | (uintptr_t)own) % 8 || ((uintptr_t)hub_rows | (uintptr_t)base) % 16)
        return ge::fail(GE_ERR_ARG, "ge_exchange_turn_bf16: misaligned buffer");
    if ((!land && !take) || vocab_size == 0) return GE_OK;
    int dev = 0, cus = 256;
    GE_HIP(hipGetDevice(&dev));
    GE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t n4 = (int64_t)vocab_size * (dim / 4);
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n4 + 255) / 256, (int64_t)cus * 8));
    hipStream_t s = (hipStream_t)stream;
    if (land && take)  hipLaunchKernelGGL((k_exchange_turn_bf16<true, true>),  dim3(blocks), dim3(256), 0, s, table, hub_rows, hub_index, dim / 4, n4, base, wire, own, seed);
    else if (land)     hipLaunchKernelGGL((k_exchange_turn_bf16<true, false>), dim3(blocks), dim3(256), 0, s, table, hub_rows, hub_index, dim / 4, n4, base, wire, own, seed);
    else               hipLaunchKernelGGL((k_exchange_turn_bf16<false, true>), dim3(blocks), dim3(256), 0, s, table, hub_rows, hub_index, dim / 4, n4, base, wire, own, seed);
    GE_HIP(hipGetLastError());
    return GE_OK;
}

extern "C" ge_status ge_exchange_turn_rows(float *table, float *base, uint16_t *wire, uint16_t *own, int64_t rows, int32_t row_stride,
                                           int32_t cols, int32_t land, int32_t take, void *stream) {
    if (!table || !base || !wire || !own) return ge::fail(GE_ERR_ARG, "ge_exchange_turn_rows: null pointer");
    if (rows < 0 || row_stride <= 0 || cols < 0 || cols > row_stride) return ge::fail(GE_ERR_ARG, "ge_exchange_turn_rows: need 0 <= cols <= row_stride, rows >= 0");
    if (((uintptr_t)table | (uintptr_t)base) % 16 || ((uintptr_t)wire | (uintptr_t)own) % 8)
        return ge::fail(GE_ERR_ARG, "ge_exchange_turn_rows: misaligned buffer");
    if ((!land && !take) || rows == 0) return GE_OK;
    int dev = 0, cus = 256;
    GE_HIP(hipGetDevice(&dev));
    GE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const bool by4 = row_stride % 4 == 0 && cols % 4 == 0;
    const int64_t n_groups = rows * (int64_t)(by4 ? row_stride / 4 : row_stride);
    const int32_t sg = by4 ? row_stride / 4 : row_stride, cg = by4 ? cols / 4 : cols;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_groups + 255) / 256, (int64_t)cus * 8));
    hipStream_t s = (hipStream_t)stream;
#define GE_LAUNCH_ROWS(L, T)                                                                                                              \
    do {                                                                                                                                  \
        if (by4) hipLaunchKernelGGL((k_exchange_turn_rows<L, T, 4>), dim3(blocks), dim3(256), 0, s, table, base, wire, own, n_groups, sg, cg); \
        else     hipLaunchKernelGGL((k_exchange_turn_rows<L, T, 1>), dim3(blocks), dim3(256), 0, s, table, base, wire, own, n_groups, sg, cg); \
    } while (0)
    if (land && take) GE_LAUNCH_ROWS(true, true);
    else if (land)    GE_LAUNCH_ROWS(true, false);
    else              GE_LAUNCH_ROWS(false, true);
#undef GE_LAUNCH_ROWS
    GE_HIP(hipGetLastError());
    return GE_OK;
}

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
