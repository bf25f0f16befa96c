// sync.hip -- multi-GPU context exchange behind the C ABI (ge_sync_*, include/geglove.h).
//
// Rows shard: rank g owns a block of focus rows and their nonzeros; the CONTEXT side (context rows, cBias and their
// AdaGrad accumulators) is replicated and reconciled by an all-reduce of the per-rank DELTAS.  The merge rule lives here
// and nowhere else (chosen by simulating the ranks with the oracle):
//   context rows, gradSqContext, gradSqCBias   new = old + sum_g delta_g            (steps carry the learning rate / squares add up)
//   cBias                                      new = old + sum_g delta_g / #{g : delta_g != 0}
//                                              (the reference updates biases WITHOUT a learning rate, Adagrad.java:88-89:
//                                               one rank alone moves a hub bias most of the way, adding eight moves diverges)
//   accumulators                               only every cfg.accum_every-th exchange
// Every fp32 table keeps  base = the consensus c (start + every landed sum, the same bits on every rank)  and  own = this
// rank's delta in flight; an exchange is
//   take:  d = narrow(table - c - own in flight);  own = d              (what bf16 drops stays in the table: error feedback)
//   ---    all-reduce(SUM) own -> wire (out of place: a take writes ONE narrow buffer) on the transport's stream, under the
//          next epoch if the caller wishes (ge_sync_turn)
//   land:  c += wire (mean rule: wire / count);  table = c + (table - c_old - own)
// (bf16 rows go through ge_exchange_turn_bf16, whose base is consensus + own in flight.)
// The table itself is never on the wire, so the epoch kernel may keep updating it while the all-reduce runs.
// Transport: RCCL over xGMI (loaded at run time, so a single-GPU user needs no RCCL), or three callbacks of the host
// (tests run two ranks on one GPU over gloo; the C++ CLI rehearses N ranks inside one process).
// The wire / base buffers are DENSE [rows x cols] whatever the table's row stride (fat rows, interleaved records).
#include "ge_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <chrono>
#include <vector>

// glove.hip
namespace ge {
ge_status glove_sync_view(ge_glove *h, int32_t *opt, int32_t *mode, void **stream, int32_t *device);
ge_status glove_epoch_segment(ge_glove *h, int32_t iteration, int32_t seg, int32_t nseg, int32_t leave_blocks, hipEvent_t after_reset);
ge_status glove_epoch_finish(ge_glove *h, double *cost_sum);
ge_status glove_epoch_progress(ge_glove *h, const unsigned long long **counter, int64_t *tickets, hipEvent_t *done);
const std::vector<int32_t> *glove_hub_columns(const ge_glove *h);
const std::vector<int32_t> *glove_hub_counts(const ge_glove *h);
const std::vector<int32_t> *glove_kernel_hubs(const ge_glove *h);
}

namespace {

__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;       // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// One exchange step over a logical [rows x cols] matrix stored in `table` with `t_stride` floats between rows; base, wire,
// own (and cnt) are dense.  base = the CONSENSUS c (start + every landed sum: the same bits on every rank), own = this
// rank's delta in flight:
//   land:  c += m (m = wire, or wire / cnt under the mean rule);  resid = (table - c_old) - own;  table = c + resid
//   take:  d = narrow(resid)  (resid = table - c: the moves not sent yet + what earlier narrowing dropped);  own = d
//          (`wire` is written by the all-reduce alone: send buffer own, receive buffer wire)
// After a synchronous exchange with an fp32 wire resid is exactly 0, so every rank's table IS c: bit-identical replicas.
// W16: the wire is bf16.  MEAN: a take also writes cnt = (d != 0), all-reduced beside the wire.
template <bool LAND, bool TAKE, bool W16, bool MEAN>
__global__ __launch_bounds__(256) void k_sync_turn(float *__restrict__ table, int64_t t_stride, int32_t cols, int64_t n,
                                                   float *__restrict__ base, void *__restrict__ wire_, void *__restrict__ own_,
                                                   float *__restrict__ cnt) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) {
        const int64_t r = k / cols;
        float *tp = table + r * t_stride + (k - r * cols);
        float c = base[k];
        float resid = *tp - c;
        if (LAND) {
            float w, o;
            if (W16) { w = bf16_to_f32(reinterpret_cast<const uint16_t *>(wire_)[k]); o = bf16_to_f32(reinterpret_cast<const uint16_t *>(own_)[k]); }
            else { w = reinterpret_cast<const float *>(wire_)[k]; o = reinterpret_cast<const float *>(own_)[k]; }
            if (MEAN) w = w / fmaxf(cnt[k], 1.0f);
            c += w;
            resid -= o;                                                   // this rank's sent delta is inside w now
            *tp = c + resid;
            base[k] = c;
        }
        if (TAKE) {
            if (W16) {
                const uint32_t h = f32_to_bf16_rne(resid);
                reinterpret_cast<uint16_t *>(own_)[k] = (uint16_t)h;
                if (MEAN) cnt[k] = bf16_to_f32(h) != 0.0f ? 1.0f : 0.0f;
            } else {
                reinterpret_cast<float *>(own_)[k] = resid;
                if (MEAN) cnt[k] = resid != 0.0f ? 1.0f : 0.0f;
            }
        }
    }
}

// The same step for the large tables (cols and the table's row stride multiples of 4, sum rule): 22 bytes per element for land +
// take on a bf16 wire (read: table 4, base 4, wire 2, own 2; written: table 4, base 4, own 2), HBM streaming.  One thread = one
// group of four consecutive elements; consecutive threads take consecutive groups of the DENSE buffers, so every wave instruction
// on base / wire / own covers one contiguous kilobyte (512 bytes on a bf16 buffer) whatever the row length, and all 64 lanes work
// (a wave per row left 14 of 64 lanes idle at dim 200 and read the dense buffers in 800-byte pieces n_waves rows apart).  The
// table side follows the records: group q lives in row q / cols4.  U groups per thread are loaded before any is stored.
// NT: the dense buffers are touched once per exchange and never fit a cache -- nontemporal accesses keep them out of the way of
// the table's lines.
template <bool LAND, bool TAKE, bool W16, bool NT>
__global__ __launch_bounds__(256) void k_sync_turn_flat4(float *__restrict__ table, int64_t t_stride, uint32_t cols4, uint32_t n4,
                                                         float *__restrict__ base, void *__restrict__ wire_, void *__restrict__ own_) {
    constexpr int U = 4;
    const uint32_t step = gridDim.x * blockDim.x;
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    auto ld4 = [](const float4 *p) -> float4 {
        if constexpr (NT) { const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
        else return *p;
    };
    auto ld2 = [](const uint2 *p) -> uint2 {
        if constexpr (NT) { const v2u v = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(p)); return make_uint2(v.x, v.y); }
        else return *p;
    };
    auto st4 = [](float4 *p, float4 v) {
        if constexpr (NT) { v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; __builtin_nontemporal_store(w, reinterpret_cast<v4f *>(p)); }
        else *p = v;
    };
    auto st2 = [](uint2 *p, uint2 v) {
        if constexpr (NT) { v2u w; w.x = v.x; w.y = v.y; __builtin_nontemporal_store(w, reinterpret_cast<v2u *>(p)); }
        else *p = v;
    };
    for (uint32_t q0 = blockIdx.x * blockDim.x + threadIdx.x; q0 < n4; q0 += U * step) {      // n4 < 2^29: vocab * dim < 2^31 (geglove.h)
        float4 tv[U], cv[U], wf[U], of[U]; uint2 wh[U], oh[U]; float4 *tp[U]; bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = q0 + (uint32_t)u * step;
            live[u] = q < n4 && q >= q0;                                   // (q >= q0: no wrap-around of the 32-bit index)
            if (!live[u]) continue;
            const uint32_t r = q / cols4, g = q - r * cols4;
            tp[u] = reinterpret_cast<float4 *>(table + (int64_t)r * t_stride) + g;
            tv[u] = *tp[u];
            cv[u] = ld4(reinterpret_cast<const float4 *>(base) + q);
            if (LAND) {
                if (W16) { wh[u] = ld2(reinterpret_cast<const uint2 *>(wire_) + q); oh[u] = ld2(reinterpret_cast<const uint2 *>(own_) + q); }
                else { wf[u] = ld4(reinterpret_cast<const float4 *>(wire_) + q); of[u] = ld4(reinterpret_cast<const float4 *>(own_) + q); }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            const uint32_t q = q0 + (uint32_t)u * step;
            float res[4] = {tv[u].x - cv[u].x, tv[u].y - cv[u].y, tv[u].z - cv[u].z, tv[u].w - cv[u].w};
            if (LAND) {
                float w[4], o[4];
                if (W16) {
                    w[0] = bf16_to_f32(wh[u].x & 0xffffu); w[1] = bf16_to_f32(wh[u].x >> 16); w[2] = bf16_to_f32(wh[u].y & 0xffffu); w[3] = bf16_to_f32(wh[u].y >> 16);
                    o[0] = bf16_to_f32(oh[u].x & 0xffffu); o[1] = bf16_to_f32(oh[u].x >> 16); o[2] = bf16_to_f32(oh[u].y & 0xffffu); o[3] = bf16_to_f32(oh[u].y >> 16);
                } else {
                    w[0] = wf[u].x; w[1] = wf[u].y; w[2] = wf[u].z; w[3] = wf[u].w; o[0] = of[u].x; o[1] = of[u].y; o[2] = of[u].z; o[3] = of[u].w;
                }
                const float c[4] = {cv[u].x + w[0], cv[u].y + w[1], cv[u].z + w[2], cv[u].w + w[3]};
#pragma unroll
                for (int k = 0; k < 4; ++k) res[k] -= o[k];
                *tp[u] = make_float4(c[0] + res[0], c[1] + res[1], c[2] + res[2], c[3] + res[3]);
                st4(reinterpret_cast<float4 *>(base) + q, make_float4(c[0], c[1], c[2], c[3]));
            }
            if (TAKE) {
                if (W16) {
                    const uint32_t h0 = f32_to_bf16_rne(res[0]), h1 = f32_to_bf16_rne(res[1]), h2 = f32_to_bf16_rne(res[2]), h3 = f32_to_bf16_rne(res[3]);
                    st2(reinterpret_cast<uint2 *>(own_) + q, make_uint2(h0 | (h1 << 16), h2 | (h3 << 16)));
                } else st4(reinterpret_cast<float4 *>(own_) + q, make_float4(res[0], res[1], res[2], res[3]));
            }
        }
    }
}

// ---- hub rows, reconciled between the segments of an epoch (ge_sync_epoch) ----------------------------------------------------
// The busiest context rows receive thousands of updates per rank and epoch.  Early in training they grow multiplicatively, and
// eight ranks that each multiply a row by 2.6 from the same start sum to a factor 14: with one exchange per epoch the sharded run
// overshoots where the single-GPU run settles, and its accumulators -- fed by gradients that were small while it lagged -- never
// catch up (measured with eight ranks on one GPU: DESIGN.md 7).  Inside one GPU the same rows are kept together by publishing
// deltas every m_j updates; across GPUs they are kept together by exchanging them S times per epoch: they are few (the union of
// the ranks' hub columns: a few thousand rows, megabytes), so each such exchange is one small fp32 all-reduce.
// One wavefront per hub row.  buf = [rows H x D | accumulator rows H x D | gradSqCBias H | cBias H | count H].
// take: buf = table - base (fp32, exact: nothing of these rows is ever in flight in the large exchange, whose take sees 0 for them)
// ROW16: the parameter rows are bf16 (GE_DTYPE_BF16 handles): a row's value is its fp32 master row where the column is a hub ON THIS
// RANK, else the bf16 table entry; a landed value goes back the same way (the bf16 entry with stochastic rounding, whose error the large
// exchange's take then finds in value - base and feeds back, as for every bf16 row).
__device__ __forceinline__ uint32_t hub_mix32(uint32_t x) { x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; x *= 0xC2B2AE3Du; x ^= x >> 16; return x; }
// The merge of a hub row's summed deltas.  Between two exchanges every rank stepped with lr / sqrt(G0 + its OWN squared gradients), where
// one sequential pass over the same updates would have had EVERYBODY's in the accumulator: summed as they are, W ranks' pushes overshoot by up
// to sqrt(W), and a row that all ranks push hard -- a hub row early in training -- leaves the single-GPU trajectory (four ranks: bumps; six
// ranks, eight exchanges per epoch: NaN in the third epoch).  The summed delta is therefore scaled, per element, by
// sqrt((G0 + E / W) / (G0 + E)) -- G0 = the accumulator all ranks agreed on at the last exchange, E = the accumulator deltas of this one,
// summed -- which is 1 while the accumulated G0 dominates and 1 / sqrt(W) when the exchange's own gradients do.  The accumulators
// themselves add.  inv_world = 0: the plain sum (GE_SYNC_MERGE=sum).  IEEE operations only (the models in tests/ repeat them bit for bit).
__device__ __forceinline__ float merge_scale(float g0, float e_sum, float inv_world) {
    if (!(inv_world > 0.0f)) return 1.0f;
    const float e = fmaxf(e_sum, 0.0f);
    return __builtin_sqrtf((g0 + e * inv_world) / (g0 + e));
}
template <bool ROW16>
__global__ __launch_bounds__(256) void k_hub_take(const int32_t *__restrict__ list, int32_t H, int32_t D,
                                                  const void *__restrict__ rows_, int64_t rows_stride, const float *__restrict__ rows_base,
                                                  const float *__restrict__ masters, const int32_t *__restrict__ master_index,
                                                  const float *__restrict__ acc, int64_t acc_stride, const float *__restrict__ acc_base,
                                                  const float *__restrict__ accb, int64_t accb_stride, const float *__restrict__ accb_base,
                                                  const float *__restrict__ bias, int64_t bias_stride, const float *__restrict__ bias_base,
                                                  float *__restrict__ buf) {
    const int lane = threadIdx.x & 63;
    const int32_t h = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= H) return;
    const int64_t v = list[h];
    const int32_t mi = ROW16 ? master_index[v] : -1;
    float *const b_rows = buf, *const b_acc = buf + (int64_t)H * D, *const b_accb = b_acc + (int64_t)H * D, *const b_bias = b_accb + H, *const b_cnt = b_bias + H;
    for (int32_t d = lane; d < D; d += 64) {
        float val;
        if (ROW16) val = mi >= 0 ? masters[(int64_t)mi * D + d] : bf16_to_f32(reinterpret_cast<const uint16_t *>(rows_)[v * rows_stride + d]);
        else val = reinterpret_cast<const float *>(rows_)[v * rows_stride + d];
        b_rows[(int64_t)h * D + d] = val - rows_base[v * D + d];
        b_acc[(int64_t)h * D + d] = acc[v * acc_stride + d] - acc_base[v * D + d];
    }
    if (lane == 0) {
        b_accb[h] = accb[v * accb_stride] - accb_base[v];
        const float db = bias[v * bias_stride] - bias_base[v];
        b_bias[h] = db; b_cnt[h] = db != 0.0f ? 1.0f : 0.0f;
    }
}
// land: base += the summed deltas (cBias: the mean over the ranks that moved it); table = base -- every rank's hub rows ARE the consensus
template <bool ROW16>
__global__ __launch_bounds__(256) void k_hub_land(const int32_t *__restrict__ list, int32_t H, int32_t D,
                                                  void *__restrict__ rows_, int64_t rows_stride, float *__restrict__ rows_base,
                                                  float *__restrict__ masters, const int32_t *__restrict__ master_index, uint32_t seed,
                                                  float *__restrict__ acc, int64_t acc_stride, float *__restrict__ acc_base,
                                                  float *__restrict__ accb, int64_t accb_stride, float *__restrict__ accb_base,
                                                  float *__restrict__ bias, int64_t bias_stride, float *__restrict__ bias_base,
                                                  const float *__restrict__ buf, float inv_world) {
    const int lane = threadIdx.x & 63;
    const int32_t h = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= H) return;
    const int64_t v = list[h];
    const int32_t mi = ROW16 ? master_index[v] : -1;
    const float *const b_rows = buf, *const b_acc = buf + (int64_t)H * D, *const b_accb = b_acc + (int64_t)H * D, *const b_bias = b_accb + H, *const b_cnt = b_bias + H;
    for (int32_t d = lane; d < D; d += 64) {
        const float c = rows_base[v * D + d] + merge_scale(acc_base[v * D + d], b_acc[(int64_t)h * D + d], inv_world) * b_rows[(int64_t)h * D + d];
        rows_base[v * D + d] = c;
        if (ROW16) {
            if (mi >= 0) masters[(int64_t)mi * D + d] = c;
            else {
                const uint32_t bits = __float_as_uint(c), rnd = hub_mix32((uint32_t)(h * D + d) * 0x9E3779B1u + seed) >> 16;
                reinterpret_cast<uint16_t *>(rows_)[v * rows_stride + d] = (uint16_t)(((bits & 0x7f800000u) == 0x7f800000u) ? bits >> 16 : (bits + rnd) >> 16);
            }
        } else reinterpret_cast<float *>(rows_)[v * rows_stride + d] = c;
        const float a = acc_base[v * D + d] + b_acc[(int64_t)h * D + d];
        acc_base[v * D + d] = a; acc[v * acc_stride + d] = a;
    }
    if (lane == 0) {
        const float a = accb_base[v] + b_accb[h];
        accb_base[v] = a; accb[v * accb_stride] = a;
        const float c = bias_base[v] + b_bias[h] / fmaxf(b_cnt[h], 1.0f);
        bias_base[v] = c; bias[v * bias_stride] = c;
    }
}
__global__ void k_mark(const int32_t *list, const int32_t *value, int32_t n, float *flags) {       // value == nullptr: 1
    const int32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) flags[list[k]] = value ? (float)value[k] : 1.0f;
}

// ---- the LIVE exchange of hub rows: while the epoch kernel runs ----------------------------------------------------------------
// The rows of `list` are hub columns of the epoch kernel on EVERY rank: the kernel moves them (and their accumulator rows) by float
// atomic adds of deltas only, never by stores -- so another adder is safe beside it.  take: own = row - base (the rank's moves since the
// last take, read with agent-coherent loads: the atomics execute at the memory side) -> buf and own; [all-reduce of buf]; land: the OTHER
// ranks' moves, buf - own, are ADDED to the row atomically and base += buf (base is touched by these two kernels only).  After a land the
// next take finds exactly the rank's moves since this take: (row + others + later) - (base + sum) = later.  Nothing here waits for the
// epoch kernel and nothing the epoch kernel does is lost; the scalars of the rows (bias, its accumulator: last-writer-wins stores in the
// kernel) are left to the exact exchange at the end of the epoch.
// (bf16 handles: `masters` / `mindex` = the fp32 master rows of the hub columns, where the epoch kernel keeps and moves such a row;
// every row of the list has one on every rank)
__global__ __launch_bounds__(256) void k_live_take(const int32_t *__restrict__ list, int32_t H, int32_t D,
                                                   const float *rows, int64_t rows_stride, const float *masters, const int32_t *__restrict__ mindex, const float *__restrict__ rows_base,
                                                   const float *acc, int64_t acc_stride, const float *__restrict__ acc_base,
                                                   float *__restrict__ buf, float *__restrict__ own) {
    const int lane = threadIdx.x & 63;
    const int32_t h = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= H) return;
    const int64_t v = list[h];
    for (int32_t d = lane; d < D; d += 64) {
        const float r = __hip_atomic_load(mindex ? masters + (int64_t)mindex[v] * D + d : rows + v * rows_stride + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float a = __hip_atomic_load(acc + v * acc_stride + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float dr = r - rows_base[v * D + d], da = a - acc_base[v * D + d];
        const int64_t k = (int64_t)h * D + d;
        buf[k] = dr; own[k] = dr;
        buf[(int64_t)H * D + k] = da; own[(int64_t)H * D + k] = da;
    }
}
__global__ __launch_bounds__(256) void k_live_land(const int32_t *__restrict__ list, int32_t H, int32_t D,
                                                   float *rows, int64_t rows_stride, float *masters, const int32_t *__restrict__ mindex, float *__restrict__ rows_base,
                                                   float *acc, int64_t acc_stride, float *__restrict__ acc_base,
                                                   const float *__restrict__ buf, const float *__restrict__ own, float inv_world) {
    const int lane = threadIdx.x & 63;
    const int32_t h = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= H) return;
    const int64_t v = list[h];
    for (int32_t d = lane; d < D; d += 64) {
        const int64_t k = (int64_t)h * D + d, ka = (int64_t)H * D + k;
        const float merged = merge_scale(acc_base[v * D + d], buf[ka], inv_world) * buf[k];      // (the same on every rank: base and buf are)
        unsafeAtomicAdd(mindex ? masters + (int64_t)mindex[v] * D + d : rows + v * rows_stride + d, merged - own[k]);
        unsafeAtomicAdd(acc + v * acc_stride + d, buf[ka] - own[ka]);
        rows_base[v * D + d] += merged;
        acc_base[v * D + d] += buf[ka];
    }
}

// dense <-> strided copies (base initialisation, replicate)
__global__ void k_gather(const float *table, int64_t t_stride, int32_t cols, int64_t n, float *dense) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) { const int64_t r = k / cols; dense[k] = table[r * t_stride + (k - r * cols)]; }
}
__global__ void k_scatter2(float *table, int64_t t_stride, int32_t cols, int64_t n, const float *dense, float *base) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) { const int64_t r = k / cols; const float v = dense[k]; table[r * t_stride + (k - r * cols)] = v; base[k] = v; }
}
// bf16 context rows: current values widened (hub rows from their fp32 masters)
__global__ void k_bf16_values(const uint16_t *table, int64_t stride, const float *hub_rows, const int32_t *hub_index, int32_t D, int64_t n, float *out) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) {
        const int64_t v = k / D; const int32_t h = hub_index[v];
        out[k] = h >= 0 ? hub_rows[(int64_t)h * D + (k - v * D)] : bf16_to_f32(table[v * stride + (k - v * D)]);
    }
}

// ---- RCCL, resolved at run time --------------------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, ncclConfig_t *) = nullptr;      // optional (a second communicator for the hub rows)
    bool ok = false;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;                 // rank threads of one process may arrive together
    std::call_once(once, [] {
        // a copy that is already in the process (PyTorch-ROCm bundles one) wins: one RCCL per process
        static const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        static const char *paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : paths) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
        r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        r.CommSplit = (decltype(r.CommSplit))dlsym(r.lib, "ncclCommSplit");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Broadcast && r.GetErrorString;
    });
    return r;
}
#define GE_NCCL(expr)                                                                                              \
    do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return ge::fail(GE_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(_r)); } while (0)

struct Entry {
    const char *name = "";
    float *table = nullptr; int64_t t_stride = 0; int32_t cols = 0; int64_t rows = 0, n = 0;
    bool mean = false, lazy = false, w16 = false, bf16_rows = false;
    float *base = nullptr, *cnt = nullptr; void *wire = nullptr, *own = nullptr;
    bool in_flight = false;
    void *ticket = nullptr, *ticket_cnt = nullptr;             // callback transport
};

unsigned grid_for(int64_t n, int cus) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)cus * 8)); }

}  // namespace

// N ranks inside ONE process (one host thread each), on any number of devices: buffers meet in host memory.  What the C++
// CLI uses when it has fewer GPUs than ranks, and what lets one GPU rehearse an N-rank run without RCCL.
struct ge_local_group {
    int world = 0;
    std::mutex m; std::condition_variable cv; int arrived = 0; unsigned long generation = 0;
    bool aborted = false;                                  // a rank failed: nobody may wait for it any more
    int timeout_s = [] { const char *e = std::getenv("GE_LOCAL_GROUP_TIMEOUT_S"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 600; }();
    std::vector<std::vector<unsigned char>> stage;        // one host buffer per rank
    // false: the group was aborted (before or while waiting); the caller returns GE_ERR_STATE
    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return false;
        const unsigned long g = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else if (!cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return generation != g || aborted; })) {
            aborted = true; cv.notify_all();              // a rank that never comes (stuck, or gone without ge_local_group_abort) must not hold the others for ever
            std::fprintf(stderr, "geglove: a rank of the local group did not reach an exchange within %d s; the group is aborted\n", timeout_s);
        }
        return !aborted;
    }
    void abort() { std::lock_guard<std::mutex> lk(m); aborted = true; cv.notify_all(); }
};

struct ge_sync {
    ge_glove *h = nullptr;
    ge_local_group *loop = nullptr;
    ge_sync_cfg cfg{};
    ge_transport tr{};              // callbacks (copied); tr.start == nullptr: RCCL
    ncclComm_t comm = nullptr;
    hipStream_t main = nullptr;     // the handle's stream: take / land kernels are ordered with its epochs
    hipStream_t side = nullptr;     // RCCL's stream: the all-reduce runs beside the next epoch
    hipEvent_t ev_taken = nullptr, ev_reduced = nullptr;
    ncclComm_t hub_comm = nullptr;  // the hub rows' small all-reduces: a communicator and a stream of their own, so that they need not
    hipStream_t hub_side = nullptr; // queue behind the large all-reduce that runs under the epoch (ncclCommSplit; else the same pair)
    hipEvent_t ev_hub_a = nullptr, ev_hub_b = nullptr;
    int device = 0, cus = 256;
    std::vector<Entry> ent;
    std::vector<void *> owned;
    int64_t calls = 0;
    uint32_t seed = 0x5EED;
    int32_t *hub_list = nullptr; int32_t n_hub = 0;       // the union of the ranks' hub columns (ascending), on the device
    float *hub_buf = nullptr;                              // [n_hub x (2 D + 3)] fp32: the hub rows' deltas of one small exchange
    float hub_top_count = 0.0f;                            // nonzeros of the busiest hub column in the whole job: sets how often the hub rows are exchanged per epoch
    int32_t *live_list = nullptr; int32_t n_live = 0;     // the columns that are hubs of the epoch kernel on EVERY rank (fp32 rows): exchanged while the kernel runs
    float *live_buf = nullptr, *live_own = nullptr;        // [n_live x 2 D] each
    hipEvent_t ev_reset = nullptr;                         // the epoch's ticket counter holds its first ticket
    unsigned long long *progress = nullptr;                // pinned host word the ticket counter is copied into
    bool live = false;                                     // ge_sync_epoch exchanges the live rows beside the epoch kernel (else: the epoch in segments)
    float merge_inv_world = 0.0f;                          // 1 / world: the hub rows' summed deltas are scaled by merge_scale(); 0: summed as they are
    int32_t live_cap = 128;                                // most live exchanges per epoch (halved once before the run gives the live form up)
    int32_t late_streak = 0;                               // consecutive live epochs in which some rank's exchanges fell behind
    int64_t live_epochs = 0, live_late = 0;                // epochs run live, and exchanges in them that were issued a whole interval late
    ge_context_layout lay{};
    template <typename T> hipError_t alloc(T **out, size_t n) {
        hipError_t e = hipMalloc((void **)out, sizeof(T) * std::max<size_t>(n, 1));
        if (e == hipSuccess) owned.push_back((void *)*out);
        return e;
    }
};

namespace {

// sum of every rank's device buffer, in rank order, back into each rank's buffer (blocking; the caller's stream is idle)
// st: the stream the device copies are ordered on (nullptr: plain blocking copies -- they wait for the null stream, i.e. for an epoch kernel
// running there; the live exchange passes its own stream)
ge_status local_allreduce(ge_local_group *g, int rank, void *buf, int64_t count, int32_t dtype, bool bcast, int src, hipStream_t st = nullptr) {
    const size_t bytes = (size_t)count * (dtype == GE_DTYPE_BF16 ? 2 : dtype == 2 ? 8 : 4);       // dtype 2: host doubles (scalars)
    // a rank whose copy fails still reaches the barriers (its peers would wait for ever otherwise) and aborts the group
    static const char *gone = "ge_local_group: another rank of the group failed";
    std::vector<unsigned char> &mine = g->stage[(size_t)rank];
    mine.resize(bytes);
    hipError_t he = hipSuccess;
    if (dtype == 2) std::memcpy(mine.data(), buf, bytes);
    else if (st) { he = hipMemcpyAsync(mine.data(), buf, bytes, hipMemcpyDeviceToHost, st); if (he == hipSuccess) he = hipStreamSynchronize(st); }
    else he = hipMemcpy(mine.data(), buf, bytes, hipMemcpyDeviceToHost);
    if (he != hipSuccess) { g->abort(); return ge::fail(GE_ERR_HIP, "local all-reduce: copy to the host failed: %s", hipGetErrorString(he)); }
    if (!g->barrier()) return ge::fail(GE_ERR_STATE, "%s", gone);
    std::vector<unsigned char> out(bytes);
    if (bcast) std::memcpy(out.data(), g->stage[(size_t)src].data(), bytes);
    else if (dtype == GE_DTYPE_F32) {
        float *o = (float *)out.data();
        for (int64_t k = 0; k < count; ++k) { float a = 0.0f; for (int r = 0; r < g->world; ++r) a += ((const float *)g->stage[(size_t)r].data())[k]; o[k] = a; }
    } else if (dtype == GE_DTYPE_BF16) {
        uint16_t *o = (uint16_t *)out.data();
        for (int64_t k = 0; k < count; ++k) {
            float a = 0.0f;
            for (int r = 0; r < g->world; ++r) { const uint32_t hbits = ((const uint16_t *)g->stage[(size_t)r].data())[k]; float f; const uint32_t u = hbits << 16; std::memcpy(&f, &u, 4); a += f; }
            uint32_t u; std::memcpy(&u, &a, 4);
            o[k] = (uint16_t)(((u & 0x7fffffffu) > 0x7f800000u) ? ((u >> 16) | 0x40u) : ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16));
        }
    } else {
        double *o = (double *)out.data();
        for (int64_t k = 0; k < count; ++k) {
            double a = ((const double *)g->stage[0].data())[k];
            for (int r = 1; r < g->world; ++r) { const double v = ((const double *)g->stage[(size_t)r].data())[k]; a = src == 1 ? std::max(a, v) : a + v; }   // src doubles as the op for scalars
            o[k] = a;
        }
    }
    if (!g->barrier()) return ge::fail(GE_ERR_STATE, "%s", gone);      // every rank has read the stage
    if (dtype == 2) std::memcpy(buf, out.data(), bytes);
    else if ((he = st ? hipMemcpyAsync(buf, out.data(), bytes, hipMemcpyHostToDevice, st) : hipMemcpy(buf, out.data(), bytes, hipMemcpyHostToDevice)) != hipSuccess
             || (st && (he = hipStreamSynchronize(st)) != hipSuccess)) {
        g->abort();
        return ge::fail(GE_ERR_HIP, "local all-reduce: copy to the device failed: %s", hipGetErrorString(he));
    }
    return GE_OK;
}

ge_status launch_turn(ge_sync *s, Entry &e, bool land, bool take) {
    if (!land && !take) return GE_OK;
    if (e.bf16_rows) {
        s->seed = s->seed * 1664525u + 1013904223u;                    // same sequence on every rank, a new draw per turn
        return ge_exchange_turn_bf16((uint16_t *)s->lay.table, s->lay.row_stride, s->lay.hub_rows, s->lay.hub_index, s->lay.vocab_size, s->lay.dim, e.base,
                                     (uint16_t *)e.wire, (uint16_t *)e.own, land, take, s->seed ^ ((uint32_t)s->cfg.rank * 0x9E3779B1u), s->main);
    }
    if (!e.mean && e.cols % 4 == 0 && e.t_stride % 4 == 0 && ((uintptr_t)e.table % 16) == 0 && e.n / 4 < ((int64_t)1 << 31)) {       // the large tables
        const int64_t n4 = e.n / 4;
        static const int per_cu = [] { const char *v = std::getenv("GE_SYNC_BLOCKS_PER_CU"); const int k = v ? std::atoi(v) : 0; return k > 0 ? k : 8; }();
        const dim3 g4((unsigned)std::max<int64_t>(1, std::min<int64_t>((n4 + 1023) / 1024, (int64_t)s->cus * per_cu))), b4(256);
        static const bool nt = [] { const char *v = std::getenv("GE_SYNC_NT"); return v ? std::atoi(v) != 0 : true; }();
#define GE_TURN4(L, T)                                                                                                              \
        do {                                                                                                                        \
            if (e.w16 && nt) hipLaunchKernelGGL((k_sync_turn_flat4<L, T, true, true>), g4, b4, 0, s->main, e.table, e.t_stride, (uint32_t)(e.cols / 4), (uint32_t)n4, e.base, e.wire, e.own); \
            else if (e.w16) hipLaunchKernelGGL((k_sync_turn_flat4<L, T, true, false>), g4, b4, 0, s->main, e.table, e.t_stride, (uint32_t)(e.cols / 4), (uint32_t)n4, e.base, e.wire, e.own); \
            else if (nt) hipLaunchKernelGGL((k_sync_turn_flat4<L, T, false, true>), g4, b4, 0, s->main, e.table, e.t_stride, (uint32_t)(e.cols / 4), (uint32_t)n4, e.base, e.wire, e.own); \
            else hipLaunchKernelGGL((k_sync_turn_flat4<L, T, false, false>), g4, b4, 0, s->main, e.table, e.t_stride, (uint32_t)(e.cols / 4), (uint32_t)n4, e.base, e.wire, e.own); \
        } while (0)
        if (land && take) GE_TURN4(true, true); else if (land) GE_TURN4(true, false); else GE_TURN4(false, true);
#undef GE_TURN4
        GE_HIP(hipGetLastError());
        return GE_OK;
    }
    const dim3 g(grid_for(e.n, s->cus)), b(256);
#define GE_TURN(L, T)                                                                                                                        \
    do {                                                                                                                                     \
        if (e.mean) hipLaunchKernelGGL((k_sync_turn<L, T, false, true>), g, b, 0, s->main, e.table, e.t_stride, e.cols, e.n, e.base, e.wire, e.own, e.cnt); \
        else if (e.w16) hipLaunchKernelGGL((k_sync_turn<L, T, true, false>), g, b, 0, s->main, e.table, e.t_stride, e.cols, e.n, e.base, e.wire, e.own, e.cnt); \
        else hipLaunchKernelGGL((k_sync_turn<L, T, false, false>), g, b, 0, s->main, e.table, e.t_stride, e.cols, e.n, e.base, e.wire, e.own, e.cnt); \
    } while (0)
    if (land && take) GE_TURN(true, true); else if (land) GE_TURN(true, false); else GE_TURN(false, true);
#undef GE_TURN
    GE_HIP(hipGetLastError());
    return GE_OK;
}

// starts the all-reduce of every entry taken in this call
ge_status start_reduce(ge_sync *s, const std::vector<Entry *> &taken) {
    if (taken.empty()) return GE_OK;
    // the transports that sum in place (local group, the host's callbacks) get the send buffer copied into the receive buffer
    // first; RCCL reduces out of place, own -> wire
    if (s->loop || s->tr.start)
        for (Entry *e : taken) GE_HIP(hipMemcpyAsync(e->wire, e->own, (size_t)e->n * (e->w16 ? 2 : 4), hipMemcpyDeviceToDevice, s->main));
    if (s->loop) {
        GE_HIP(hipStreamSynchronize(s->main));
        for (Entry *e : taken) {
            ge_status st = local_allreduce(s->loop, s->cfg.rank, e->wire, e->n, e->w16 ? GE_DTYPE_BF16 : GE_DTYPE_F32, false, 0);
            if (st == GE_OK && e->mean) st = local_allreduce(s->loop, s->cfg.rank, e->cnt, e->n, GE_DTYPE_F32, false, 0);
            if (st != GE_OK) return st;
        }
    } else if (!s->tr.start) {
        GE_HIP(hipEventRecord(s->ev_taken, s->main));
        GE_HIP(hipStreamWaitEvent(s->side, s->ev_taken, 0));
        for (Entry *e : taken) {
            GE_NCCL(rccl().AllReduce(e->own, e->wire, (size_t)e->n, e->w16 ? ncclBfloat16 : ncclFloat32, ncclSum, s->comm, s->side));
            if (e->mean) GE_NCCL(rccl().AllReduce(e->cnt, e->cnt, (size_t)e->n, ncclFloat32, ncclSum, s->comm, s->side));
        }
        GE_HIP(hipEventRecord(s->ev_reduced, s->side));
    } else {
        GE_HIP(hipStreamSynchronize(s->main));                          // the host's collective reads the buffers
        for (Entry *e : taken) {
            ge_status st = s->tr.start(s->tr.user, e->wire, e->n, e->w16 ? GE_DTYPE_BF16 : GE_DTYPE_F32, &e->ticket);
            if (st == GE_OK && e->mean) st = s->tr.start(s->tr.user, e->cnt, e->n, GE_DTYPE_F32, &e->ticket_cnt);
            if (st != GE_OK) return ge::fail(st, "transport.start failed for %s", e->name);
        }
    }
    for (Entry *e : taken) e->in_flight = true;
    return GE_OK;
}

ge_status wait_reduce(ge_sync *s) {
    bool any = false;
    for (Entry &e : s->ent) any = any || e.in_flight;
    if (!any || s->loop) return GE_OK;
    if (!s->tr.start) { GE_HIP(hipStreamWaitEvent(s->main, s->ev_reduced, 0)); return GE_OK; }
    for (Entry &e : s->ent) {
        if (!e.in_flight) continue;
        ge_status st = s->tr.wait(s->tr.user, e.ticket);
        if (st == GE_OK && e.mean) st = s->tr.wait(s->tr.user, e.ticket_cnt);
        if (st != GE_OK) return ge::fail(st, "transport.wait failed for %s", e.name);
    }
    return GE_OK;
}

// sum of a small fp32 device buffer over the ranks, ordered on the handle's stream (RCCL: asynchronous, on the hub stream)
ge_status allreduce_f32_small(ge_sync *s, float *dbuf, int64_t n) {
    if (s->loop) {
        GE_HIP(hipStreamSynchronize(s->main));
        return local_allreduce(s->loop, s->cfg.rank, dbuf, n, GE_DTYPE_F32, false, 0);
    }
    if (s->tr.start) {
        GE_HIP(hipStreamSynchronize(s->main));
        void *t = nullptr;
        ge_status st = s->tr.start(s->tr.user, dbuf, n, GE_DTYPE_F32, &t);
        if (st == GE_OK) st = s->tr.wait(s->tr.user, t);
        return st == GE_OK ? GE_OK : ge::fail(st, "transport failed for the hub rows");
    }
    GE_HIP(hipEventRecord(s->ev_hub_a, s->main));
    GE_HIP(hipStreamWaitEvent(s->hub_side, s->ev_hub_a, 0));
    GE_NCCL(rccl().AllReduce(dbuf, dbuf, (size_t)n, ncclFloat32, ncclSum, s->hub_comm, s->hub_side));
    GE_HIP(hipEventRecord(s->ev_hub_b, s->hub_side));
    GE_HIP(hipStreamWaitEvent(s->main, s->ev_hub_b, 0));
    return GE_OK;
}

// one small exchange of the hub rows (rows, accumulator rows, both scalars): take, all-reduce, land; exact replicas of these rows after it
ge_status hub_exchange(ge_sync *s) {
    if (s->n_hub == 0) return GE_OK;
    const Entry &er = s->ent[0], &eb = s->ent[1], &ea = s->ent[2], &eab = s->ent[3];
    const int32_t H = s->n_hub, D = s->lay.dim;
    const dim3 g((unsigned)((H + 3) / 4)), b(256);
    const bool r16 = er.bf16_rows;
    void *rows = r16 ? (void *)s->lay.table : (void *)er.table;
    const int64_t rstride = r16 ? (int64_t)s->lay.row_stride : er.t_stride;
    if (r16) hipLaunchKernelGGL(k_hub_take<true>, g, b, 0, s->main, s->hub_list, H, D, rows, rstride, er.base, s->lay.hub_rows, s->lay.hub_index, ea.table, ea.t_stride, ea.base,
                                eab.table, eab.t_stride, eab.base, eb.table, eb.t_stride, eb.base, s->hub_buf);
    else hipLaunchKernelGGL(k_hub_take<false>, g, b, 0, s->main, s->hub_list, H, D, rows, rstride, er.base, (const float *)nullptr, (const int32_t *)nullptr, ea.table, ea.t_stride, ea.base,
                            eab.table, eab.t_stride, eab.base, eb.table, eb.t_stride, eb.base, s->hub_buf);
    GE_HIP(hipGetLastError());
    ge_status st = allreduce_f32_small(s, s->hub_buf, (int64_t)H * (2 * D + 3));
    if (st != GE_OK) return st;
    s->seed = s->seed * 1664525u + 1013904223u;
    const uint32_t seed = s->seed ^ ((uint32_t)s->cfg.rank * 0x9E3779B1u);
    if (r16) hipLaunchKernelGGL(k_hub_land<true>, g, b, 0, s->main, s->hub_list, H, D, rows, rstride, er.base, s->lay.hub_rows, s->lay.hub_index, seed, ea.table, ea.t_stride, ea.base,
                                eab.table, eab.t_stride, eab.base, eb.table, eb.t_stride, eb.base, s->hub_buf, s->merge_inv_world);
    else hipLaunchKernelGGL(k_hub_land<false>, g, b, 0, s->main, s->hub_list, H, D, rows, rstride, er.base, (float *)nullptr, (const int32_t *)nullptr, seed, ea.table, ea.t_stride, ea.base,
                            eab.table, eab.t_stride, eab.base, eb.table, eb.t_stride, eb.base, s->hub_buf, s->merge_inv_world);
    GE_HIP(hipGetLastError());
    return GE_OK;
}

// one live exchange (k_live_take / all-reduce / k_live_land), everything on the hub stream: safe while the epoch kernel runs on the handle's
ge_status live_exchange(ge_sync *s) {
    if (s->n_live == 0) return GE_OK;
    const Entry &er = s->ent[0], &ea = s->ent[2];
    const int32_t H = s->n_live, D = s->lay.dim;
    const dim3 g((unsigned)((H + 3) / 4)), b(256);
    const bool r16 = er.bf16_rows;
    hipLaunchKernelGGL(k_live_take, g, b, 0, s->hub_side, (const int32_t *)s->live_list, H, D, (const float *)er.table, er.t_stride,
                       r16 ? (const float *)s->lay.hub_rows : (const float *)nullptr, r16 ? s->lay.hub_index : (const int32_t *)nullptr, (const float *)er.base,
                       (const float *)ea.table, ea.t_stride, (const float *)ea.base, s->live_buf, s->live_own);
    GE_HIP(hipGetLastError());
    const int64_t n = (int64_t)H * 2 * D;
    if (s->loop) {
        GE_HIP(hipStreamSynchronize(s->hub_side));
        ge_status st = local_allreduce(s->loop, s->cfg.rank, s->live_buf, n, GE_DTYPE_F32, false, 0, s->hub_side);
        if (st != GE_OK) return st;
    } else GE_NCCL(rccl().AllReduce(s->live_buf, s->live_buf, (size_t)n, ncclFloat32, ncclSum, s->hub_comm, s->hub_side));
    hipLaunchKernelGGL(k_live_land, g, b, 0, s->hub_side, (const int32_t *)s->live_list, H, D, er.table, er.t_stride,
                       r16 ? s->lay.hub_rows : (float *)nullptr, r16 ? s->lay.hub_index : (const int32_t *)nullptr, er.base, ea.table, ea.t_stride, ea.base,
                       (const float *)s->live_buf, (const float *)s->live_own, s->merge_inv_world);
    GE_HIP(hipGetLastError());
    return GE_OK;
}

// how ge_sync_epoch reconciles the hub rows: how often per epoch, and whether beside the running kernel (live) or between segments of it.
// The busiest column decides.  Measured on the bench's matrix split over four ranks (625 k vertices, 103 M nonzeros, the busiest column
// in every row: 625 k nonzeros): 8 exchanges per epoch -- 78 k updates of that row between two exchanges, all ranks together --
// leave the single-GPU cost trajectory in epochs 3 to 5 and may not come back, 16 follow it within 2 %, 24 and more exactly -- with the hub
// rows' deltas summed as they are.  With merge_scale (above) 8 exchanges already stay within 2 - 4 % at four and at six ranks and 16 within
// 1 %.  So: one exchange per 65 536 updates of the busiest column (10 there; 76 at the bench's eight-GPU size, where a rank puts four
// times as many updates on that column per epoch), at least max(8, ranks); a live exchange costs the epoch nothing and may come 128 times
// per epoch, a segment costs a kernel boundary (0.18 ms at the bench size) plus the exchange and is capped at 64.
void hub_plan(const ge_sync *s, int32_t segments, bool *live, int32_t *exchanges) {
    const bool lv = s->live && s->n_live > 0;
    const int32_t cap = lv ? s->live_cap : 64;
    int32_t n = segments;
    if (n <= 0) {
        n = (int32_t)std::min<double>(cap, std::ceil((double)s->hub_top_count / 65536.0));
        n = std::max(n, std::max(8, s->cfg.world));
    }
    *live = lv; *exchanges = std::min(n, cap);
}

ge_status turn(ge_sync *s, bool land, bool take, bool everything) {
    if (!s) return ge::fail(GE_ERR_ARG, "null ge_sync handle");
    if (s->cfg.world == 1) return GE_OK;
    GE_HIP(hipSetDevice(s->device));
    bool due = false;
    if (take) { ++s->calls; due = everything || s->calls % std::max(1, s->cfg.accum_every) == 0; }
    if (take && !land) for (Entry &e : s->ent) if (e.in_flight) return ge::fail(GE_ERR_STATE, "ge_sync: finish the exchange in flight first");
    if (land) { ge_status st = wait_reduce(s); if (st != GE_OK) return st; }
    std::vector<Entry *> taken;
    for (Entry &e : s->ent) {
        const bool do_land = land && e.in_flight, do_take = take && (due || !e.lazy);
        ge_status st = launch_turn(s, e, do_land, do_take);
        if (st != GE_OK) return st;
        if (do_land) e.in_flight = false;
        if (do_take) taken.push_back(&e);
    }
    return start_reduce(s, taken);
}

}  // namespace

extern "C" {

int32_t ge_sync_cfg_size(void) { return (int32_t)sizeof(ge_sync_cfg); }

ge_status ge_local_group_create(int32_t world, ge_local_group **out) {
    if (!out || world < 1) return ge::fail(GE_ERR_ARG, "ge_local_group_create: world must be >= 1");
    ge_local_group *g = new (std::nothrow) ge_local_group();
    if (!g) return ge::fail(GE_ERR_OOM, "host allocation failed");
    g->world = world; g->stage.resize((size_t)world);
    *out = g;
    return GE_OK;
}
void ge_local_group_destroy(ge_local_group *g) { delete g; }
void ge_local_group_abort(ge_local_group *g) { if (g) g->abort(); }

ge_status ge_rccl_unique_id(void *id128) {
    if (!id128) return ge::fail(GE_ERR_ARG, "null id buffer");
    if (!rccl().ok) return ge::fail(GE_ERR_HIP, "RCCL is not available (librccl.so.1 could not be loaded)");
    ncclUniqueId id;
    GE_NCCL(rccl().GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return GE_OK;
}

// One rank, one GPU: RCCL is opened, a communicator of size 1 made, a sum and a broadcast run through it and compared
// with what went in.  What a single-GPU box can check of the RCCL path (symbols, calling convention, stream order).
ge_status ge_rccl_selftest(int32_t device) {
    ge_status st = ge::select_device(device);
    if (st != GE_OK) return st;
    if (!rccl().ok) return ge::fail(GE_ERR_HIP, "RCCL is not available (librccl.so.1 could not be loaded)");
    ncclUniqueId id;
    GE_NCCL(rccl().GetUniqueId(&id));
    ncclComm_t comm = nullptr;
    GE_NCCL(rccl().CommInitRank(&comm, 1, id, 0));
    const int n = 4096;
    float *d = nullptr; hipStream_t side = nullptr;
    std::vector<float> h((size_t)n), back((size_t)n);
    for (int k = 0; k < n; ++k) h[(size_t)k] = 0.25f * (float)k - 7.0f;
    hipError_t he = hipMalloc((void **)&d, sizeof(float) * n);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipMemcpyAsync(d, h.data(), sizeof(float) * n, hipMemcpyHostToDevice, side);
    ncclResult_t nr = ncclSuccess;
    if (he == hipSuccess) nr = rccl().AllReduce(d, d, (size_t)n, ncclFloat32, ncclSum, comm, side);
    if (he == hipSuccess && nr == ncclSuccess) nr = rccl().Broadcast(d, d, (size_t)n, ncclFloat32, 0, comm, side);
    ncclComm_t comm2 = nullptr;                      // the second communicator ge_sync makes for the hub rows (ncclCommSplit), used the same way
    if (he == hipSuccess && nr == ncclSuccess && rccl().CommSplit) {
        nr = rccl().CommSplit(comm, 0, 0, &comm2, nullptr);
        if (nr == ncclSuccess && comm2) nr = rccl().AllReduce(d, d, (size_t)n, ncclFloat32, ncclSum, comm2, side);
    }
    if (he == hipSuccess && nr == ncclSuccess) he = hipMemcpyAsync(back.data(), d, sizeof(float) * n, hipMemcpyDeviceToHost, side);
    if (he == hipSuccess && nr == ncclSuccess) he = hipStreamSynchronize(side);
    if (d) (void)hipFree(d);
    if (side) (void)hipStreamDestroy(side);
    if (comm2) (void)rccl().CommDestroy(comm2);
    (void)rccl().CommDestroy(comm);
    if (nr != ncclSuccess) return ge::fail(GE_ERR_HIP, "RCCL self-test: %s", rccl().GetErrorString(nr));
    if (he != hipSuccess) return ge::fail(GE_ERR_HIP, "RCCL self-test: %s", hipGetErrorString(he));
    if (std::memcmp(h.data(), back.data(), sizeof(float) * n) != 0) return ge::fail(GE_ERR_STATE, "RCCL self-test: a one-rank sum changed the data");
    return GE_OK;
}

void ge_sync_destroy(ge_sync *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->main) (void)hipStreamSynchronize(s->main);
    if (s->side) (void)hipStreamSynchronize(s->side);
    if (s->hub_side) (void)hipStreamSynchronize(s->hub_side);
    if (s->hub_comm && s->hub_comm != s->comm && rccl().ok) (void)rccl().CommDestroy(s->hub_comm);
    if (s->comm && rccl().ok) (void)rccl().CommDestroy(s->comm);
    if (s->ev_reset) (void)hipEventDestroy(s->ev_reset);
    if (s->progress) (void)hipHostFree(s->progress);
    if (s->ev_hub_a) (void)hipEventDestroy(s->ev_hub_a);
    if (s->ev_hub_b) (void)hipEventDestroy(s->ev_hub_b);
    if (s->hub_side) (void)hipStreamDestroy(s->hub_side);
    for (void *q : s->owned) (void)hipFree(q);
    if (s->ev_taken) (void)hipEventDestroy(s->ev_taken);
    if (s->ev_reduced) (void)hipEventDestroy(s->ev_reduced);
    if (s->side) (void)hipStreamDestroy(s->side);
    delete s;
}

static ge_status ge_sync_create_impl(ge_glove *h, const ge_sync_cfg *cfg, ge_sync **out) {
    if (!out) return ge::fail(GE_ERR_ARG, "out is null");
    *out = nullptr;
    if (!h || !cfg) return ge::fail(GE_ERR_ARG, "null argument");
    if (cfg->world < 1 || cfg->rank < 0 || cfg->rank >= cfg->world) return ge::fail(GE_ERR_ARG, "invalid world / rank %d / %d", cfg->world, cfg->rank);
    if (cfg->wire != GE_DTYPE_F32 && cfg->wire != GE_DTYPE_BF16) return ge::fail(GE_ERR_ARG, "wire must be GE_DTYPE_F32 or GE_DTYPE_BF16");
    if (cfg->accum_every < 0) return ge::fail(GE_ERR_ARG, "accum_every must be >= 0");
    if (cfg->transport && (!cfg->transport->start || !cfg->transport->wait || !cfg->transport->broadcast)) return ge::fail(GE_ERR_ARG, "transport needs start, wait and broadcast");
    int32_t opt = 0, mode = 0, device = 0; void *stream = nullptr;
    ge_status st = ge::glove_sync_view(h, &opt, &mode, &stream, &device);
    if (st != GE_OK) return st;
    if (mode != GE_MODE_HOGWILD) return ge::fail(GE_ERR_STATE, "the context exchange is defined for GE_MODE_HOGWILD handles");
    if (opt != GE_OPT_ADAGRAD) return ge::fail(GE_ERR_STATE, "the context exchange (which deltas add, which average) is defined for adagrad only");
    ge_sync *s = new (std::nothrow) ge_sync();
    if (!s) return ge::fail(GE_ERR_OOM, "host allocation failed");
    s->h = h; s->cfg = *cfg; s->cfg.transport = nullptr; s->cfg.rccl_id = nullptr; s->cfg.local_group = nullptr;
    s->loop = cfg->local_group;
    if (s->loop && s->loop->world != cfg->world) { delete s; return ge::fail(GE_ERR_ARG, "local group has %d ranks, cfg.world is %d", s->loop->world, cfg->world); }
    if (s->cfg.accum_every == 0) s->cfg.accum_every = 4;
    s->merge_inv_world = 1.0f / (float)std::max(1, cfg->world);
    if (const char *e = std::getenv("GE_SYNC_MERGE")) if (std::strcmp(e, "sum") == 0) s->merge_inv_world = 0.0f;        // experiments: the hub rows' deltas summed as they are
    if (cfg->transport) s->tr = *cfg->transport;
    s->main = (hipStream_t)stream; s->device = device;
#define GE_TRYS(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { ge_status _s = ge::fail(_e == hipErrorOutOfMemory ? GE_ERR_OOM : GE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); ge_sync_destroy(s); return _s; } } while (0)
    GE_TRYS(hipSetDevice(device));
    GE_TRYS(hipDeviceGetAttribute(&s->cus, hipDeviceAttributeMultiprocessorCount, device));
    st = ge_glove_context_layout(h, &s->lay);
    if (st != GE_OK) { ge_sync_destroy(s); return st; }
    const int64_t V = s->lay.vocab_size; const int32_t D = s->lay.dim;
    const bool w16 = cfg->wire == GE_DTYPE_BF16;
    auto add = [&](const char *name, float *table, int64_t stride, int32_t cols, bool mean, bool lazy, bool narrow) {
        Entry e; e.name = name; e.table = table; e.t_stride = stride; e.cols = cols; e.rows = V; e.n = V * cols;
        e.mean = mean; e.lazy = lazy; e.w16 = narrow && !mean;
        s->ent.push_back(e);
    };
    if (s->lay.dtype == GE_DTYPE_BF16) {
        add("context rows (bf16)", nullptr, D, D, false, false, true);
        s->ent.back().bf16_rows = true;
    } else add("context rows", (float *)s->lay.table, s->lay.row_stride, D, false, false, w16);
    // the two scalars of a row, wherever the handle keeps them (a vector, a column of the fat row, columns behind the accumulator row)
    add("cBias", s->lay.bias, s->lay.bias_stride, 1, true, false, false);
    add("gradSqContext", s->lay.accum, s->lay.accum_stride, D, false, true, w16);
    add("gradSqCBias", s->lay.accum_bias, s->lay.accum_bias_stride, 1, false, true, false);
    if (cfg->world > 1) {
        for (Entry &e : s->ent) {
            GE_TRYS(s->alloc(&e.base, (size_t)e.n));
            const size_t wb = e.w16 ? 2 : 4;
            GE_TRYS(s->alloc((char **)&e.wire, (size_t)e.n * wb)); GE_TRYS(s->alloc((char **)&e.own, (size_t)e.n * wb));
            if (e.mean) GE_TRYS(s->alloc(&e.cnt, (size_t)e.n));
            // the base is the table NOW, before any local pass
            if (e.bf16_rows) hipLaunchKernelGGL(k_bf16_values, dim3(grid_for(e.n, s->cus)), dim3(256), 0, s->main, (const uint16_t *)s->lay.table, (int64_t)s->lay.row_stride, s->lay.hub_rows,
                                                s->lay.hub_index, D, e.n, e.base);
            else hipLaunchKernelGGL(k_gather, dim3(grid_for(e.n, s->cus)), dim3(256), 0, s->main, e.table, e.t_stride, e.cols, e.n, e.base);
        }
        GE_TRYS(hipGetLastError());
        GE_TRYS(hipStreamSynchronize(s->main));
        if (!s->tr.start && !s->loop) {
            if (!cfg->rccl_id) { ge_sync_destroy(s); return ge::fail(GE_ERR_ARG, "world > 1 needs a transport, a local group or an RCCL unique id (ge_rccl_unique_id on rank 0, handed to every rank)"); }
            if (!rccl().ok) { ge_sync_destroy(s); return ge::fail(GE_ERR_HIP, "RCCL is not available (librccl.so.1 could not be loaded)"); }
            GE_TRYS(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
            GE_TRYS(hipEventCreateWithFlags(&s->ev_taken, hipEventDisableTiming));
            GE_TRYS(hipEventCreateWithFlags(&s->ev_reduced, hipEventDisableTiming));
            ncclUniqueId id; std::memcpy(&id, cfg->rccl_id, sizeof(id));
            ncclResult_t r = rccl().CommInitRank(&s->comm, cfg->world, id, cfg->rank);
            if (r != ncclSuccess) { ge_status e2 = ge::fail(GE_ERR_HIP, "ncclCommInitRank failed: %s", rccl().GetErrorString(r)); s->comm = nullptr; ge_sync_destroy(s); return e2; }
            GE_TRYS(hipStreamCreateWithFlags(&s->hub_side, hipStreamNonBlocking));
            GE_TRYS(hipEventCreateWithFlags(&s->ev_hub_a, hipEventDisableTiming));
            GE_TRYS(hipEventCreateWithFlags(&s->ev_hub_b, hipEventDisableTiming));
            // The hub rows' communicator: a split of the first one (every rank makes the same choice: the symbol is there or it is not,
            // GE_SYNC_HUB_COMM=shared turns it off everywhere; a split that fails is an error, not a silent fallback that would leave
            // the ranks on different communicators)
            s->hub_comm = s->comm;
            const char *hc = std::getenv("GE_SYNC_HUB_COMM");
            if (rccl().CommSplit && !(hc && std::strcmp(hc, "shared") == 0)) {
                ncclComm_t c2 = nullptr;
                ncclResult_t r2 = rccl().CommSplit(s->comm, 0, cfg->rank, &c2, nullptr);
                if (r2 != ncclSuccess || !c2) { ge_status e2 = ge::fail(GE_ERR_HIP, "ncclCommSplit failed: %s (GE_SYNC_HUB_COMM=shared uses one communicator)", rccl().GetErrorString(r2)); ge_sync_destroy(s); return e2; }
                s->hub_comm = c2;
            }
        }
        // The hub rows of the small exchanges (ge_sync_epoch): the union of the ranks' busy columns -- count on the rank >= max(256,
        // N_rank / 20 480), whatever the handle's worker count or hot-column setting (every rank flags its own in a [V] vector, the
        // vector is summed).  A bf16 handle keeps the fp32 masters of ITS hubs, and a column that is a
        // hub on one rank and an ordinary bf16 row on another is written back as each rank stores it (k_hub_land).
        {
            const std::vector<int32_t> *mine = ge::glove_hub_columns(h), *mine_n = ge::glove_hub_counts(h);
            float *flags = nullptr; int32_t *tmp = nullptr, *tmp_n = nullptr;
            GE_TRYS(hipMalloc((void **)&flags, sizeof(float) * (size_t)std::max<int64_t>(V, 1)));
            s->owned.push_back(flags);
            // pass 1: every rank writes the nonzero counts of ITS busy columns, the vector is summed: > 0 = in the union, the value = the
            // column's nonzeros in the whole job (over the ranks that flagged it)
            GE_TRYS(hipMemsetAsync(flags, 0, sizeof(float) * (size_t)V, s->main));
            const int32_t nm = mine ? (int32_t)mine->size() : 0;
            if (nm > 0) {
                GE_TRYS(hipMalloc((void **)&tmp, sizeof(int32_t) * (size_t)nm));
                s->owned.push_back(tmp);
                GE_TRYS(hipMalloc((void **)&tmp_n, sizeof(int32_t) * (size_t)nm));
                s->owned.push_back(tmp_n);
                GE_TRYS(hipMemcpyAsync(tmp, mine->data(), sizeof(int32_t) * (size_t)nm, hipMemcpyHostToDevice, s->main));
                GE_TRYS(hipMemcpyAsync(tmp_n, mine_n->data(), sizeof(int32_t) * (size_t)nm, hipMemcpyHostToDevice, s->main));
                hipLaunchKernelGGL(k_mark, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, s->main, (const int32_t *)tmp, (const int32_t *)tmp_n, nm, flags);
            }
            st = allreduce_f32_small(s, flags, V);
            if (st != GE_OK) { ge_sync_destroy(s); return st; }
            std::vector<float> hf((size_t)V);
            GE_TRYS(hipMemcpyAsync(hf.data(), flags, sizeof(float) * (size_t)V, hipMemcpyDeviceToHost, s->main));
            GE_TRYS(hipStreamSynchronize(s->main));
            std::vector<int32_t> all;
            for (int64_t v = 0; v < V; ++v) if (hf[(size_t)v] > 0.0f) { all.push_back((int32_t)v); s->hub_top_count = std::max(s->hub_top_count, hf[(size_t)v]); }
            s->n_hub = (int32_t)all.size();
            if (s->n_hub > 0) {
                GE_TRYS(s->alloc(&s->hub_list, (size_t)s->n_hub));
                GE_TRYS(hipMemcpy(s->hub_list, all.data(), sizeof(int32_t) * all.size(), hipMemcpyHostToDevice));
                GE_TRYS(s->alloc(&s->hub_buf, (size_t)s->n_hub * (size_t)(2 * D + 3)));
            }
            // The live form (fp32 rows; RCCL or a local group -- a host callback cannot run beside the epoch kernel): the hub rows are exchanged
            // while the epoch kernel runs, which is safe for rows the kernel moves by atomic adds only: its hub columns.  A sharded handle's
            // hub rule includes every column that is busy on the rank itself (glove_layout.hip), so what is not a hub here is a column at
            // the threshold that another rank counted in: a delta landing on such a row can fall into a worker's load-update-store and be
            // overwritten -- that rank then reports the loss with its next take and the replicas stay consistent (one exchange's worth
            // of the others' moves on that row is dropped, what Hogwild does to ordinary rows all the time).  Every rank must find its own busy
            // columns among its kernel's hubs (not so with hot columns switched off): they agree by a one-word all-reduce.
            const char *mode_env = std::getenv("GE_SYNC_EPOCH");                     // "segments": the epoch in segments, as before the live exchange
            const std::vector<int32_t> *kh = ge::glove_kernel_hubs(h);
            bool mine_ok = kh != nullptr;
            if (mine_ok && mine) for (int32_t v : *mine) if (!std::binary_search(kh->begin(), kh->end(), v)) { mine_ok = false; break; }
            const bool can = !s->tr.start && !(mode_env && std::strcmp(mode_env, "segments") == 0) && s->n_hub > 0;
            const float vote = (can && mine_ok) ? 0.0f : 1.0f;
            GE_TRYS(hipMemcpyAsync(flags, &vote, sizeof(float), hipMemcpyHostToDevice, s->main));
            GE_TRYS(hipStreamSynchronize(s->main));
            st = allreduce_f32_small(s, flags, 1);
            if (st != GE_OK) { ge_sync_destroy(s); return st; }
            float against = 1.0f;
            GE_TRYS(hipMemcpyAsync(&against, flags, sizeof(float), hipMemcpyDeviceToHost, s->main));
            GE_TRYS(hipStreamSynchronize(s->main));
            if (against == 0.0f && s->lay.dtype == GE_DTYPE_BF16) {
                // bf16 rows: a hub row lives in an fp32 master row that the kernel moves by atomics -- where the column is a hub of the rank.
                // The live set is therefore the columns that are hubs on EVERY rank; the few columns at the threshold that are not get
                // the exact exchange at the end of the epoch only.
                GE_TRYS(hipMemsetAsync(flags, 0, sizeof(float) * (size_t)V, s->main));
                const int32_t nk = (int32_t)kh->size();
                if (nk > 0) {
                    int32_t *tk = nullptr;
                    GE_TRYS(hipMalloc((void **)&tk, sizeof(int32_t) * (size_t)nk));
                    s->owned.push_back(tk);
                    GE_TRYS(hipMemcpyAsync(tk, kh->data(), sizeof(int32_t) * (size_t)nk, hipMemcpyHostToDevice, s->main));
                    hipLaunchKernelGGL(k_mark, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s->main, (const int32_t *)tk, (const int32_t *)nullptr, nk, flags);
                }
                st = allreduce_f32_small(s, flags, V);
                if (st != GE_OK) { ge_sync_destroy(s); return st; }
                GE_TRYS(hipMemcpyAsync(hf.data(), flags, sizeof(float) * (size_t)V, hipMemcpyDeviceToHost, s->main));
                GE_TRYS(hipStreamSynchronize(s->main));
                std::vector<int32_t> lv;
                for (int32_t v : all) if (hf[(size_t)v] == (float)cfg->world) lv.push_back(v);
                s->n_live = (int32_t)lv.size();
                if (s->n_live > 0) {
                    GE_TRYS(s->alloc(&s->live_list, (size_t)s->n_live));
                    GE_TRYS(hipMemcpy(s->live_list, lv.data(), sizeof(int32_t) * lv.size(), hipMemcpyHostToDevice));
                }
            } else if (against == 0.0f) { s->live_list = s->hub_list; s->n_live = s->n_hub; }
            if (against == 0.0f && s->n_live > 0) {
                GE_TRYS(s->alloc(&s->live_buf, (size_t)s->n_live * (size_t)(2 * D)));
                GE_TRYS(s->alloc(&s->live_own, (size_t)s->n_live * (size_t)(2 * D)));
                if (!s->hub_side) GE_TRYS(hipStreamCreateWithFlags(&s->hub_side, hipStreamNonBlocking));      // (a local group has no RCCL streams)
                GE_TRYS(hipEventCreateWithFlags(&s->ev_reset, hipEventDisableTiming));
                GE_TRYS(hipHostMalloc((void **)&s->progress, sizeof(unsigned long long)));
                s->live = true;
            }
        }
    }
#undef GE_TRYS
    *out = s;
    return GE_OK;
}
ge_status ge_sync_create(ge_glove *h, const ge_sync_cfg *cfg, ge_sync **out) { GE_GUARD(ge_sync_create_impl(h, cfg, out)); }

// a rank of a local group that fails inside an exchange call takes the group down with it: its peers' barriers return
// GE_ERR_STATE instead of waiting for a rank that will not come
static ge_status with_abort(ge_sync *s, ge_status st) { if (st != GE_OK && s && s->loop) s->loop->abort(); return st; }
static ge_status turn_guarded(ge_sync *s, bool land, bool take, bool everything) { GE_GUARD(turn(s, land, take, everything)); }
ge_status ge_sync_begin(ge_sync *s, int32_t everything) { return with_abort(s, turn_guarded(s, false, true, everything != 0)); }
ge_status ge_sync_finish(ge_sync *s) { return with_abort(s, turn_guarded(s, true, false, false)); }
ge_status ge_sync_turn(ge_sync *s) { return with_abort(s, turn_guarded(s, true, true, false)); }
ge_status ge_sync_sync(ge_sync *s) {          // lands what an earlier turn left in flight, takes, lands: nothing is in flight afterwards
    ge_status st = ge_sync_turn(s);
    return st == GE_OK ? ge_sync_finish(s) : st;
}

// One epoch of a sharded run: the handle's epoch in `segments` launches with a small exchange of the hub rows behind each (see
// k_hub_take).  segments <= 0: the number of ranks, at least 8.  The large exchange (ge_sync_turn / ge_sync_sync) follows as
// before and finds nothing to do for the hub rows.  world == 1, or no hub rows: ge_glove_epoch.
static ge_status ge_sync_epoch_impl(ge_sync *s, int32_t iteration, int32_t segments, double *cost_sum) {
    if (!s) return ge::fail(GE_ERR_ARG, "null ge_sync handle");
    if (s->cfg.world == 1 || s->n_hub == 0) return ge_glove_epoch(s->h, iteration, cost_sum);
    GE_HIP(hipSetDevice(s->device));
    bool live = false; int32_t S = 8;
    hub_plan(s, segments, &live, &S);
    if (!live) {
        for (int32_t seg = 0; seg < S; ++seg) {
            ge_status st = ge::glove_epoch_segment(s->h, iteration, seg, S, 0, nullptr);
            if (st == GE_OK) st = hub_exchange(s);
            if (st != GE_OK) return st;
        }
        return ge::glove_epoch_finish(s->h, cost_sum);
    }
    // Live: ONE launch of the whole epoch (32 workgroups fewer: their 128 wavefront slots are where the small kernels and RCCL's run), and
    // while it runs this thread watches the epoch's ticket counter and puts the r-th of S - 1 live exchanges on the hub stream when r / S
    // of the tickets are out.  Every rank does exactly S - 1 of them (they are collective) whatever its own pace; an exchange that finds
    // its turn late -- the ranks wait for each other inside the all-reduce, never the epoch kernels -- just runs later.  The epoch ends
    // with the exact exchange of all hub rows (scalars included), as every segment of the segmented form does.
    ge_status st = ge::glove_epoch_segment(s->h, iteration, 0, 1, 32, s->ev_reset);
    if (st != GE_OK) return st;
    const unsigned long long *counter = nullptr; int64_t tickets = 0; hipEvent_t done = nullptr;
    if ((st = ge::glove_epoch_progress(s->h, &counter, &tickets, &done)) != GE_OK) return st;
    GE_HIP(hipEventSynchronize(s->ev_reset));                 // from here on the counter is this epoch's
    const bool dbg = std::getenv("GE_SYNC_DEBUG") != nullptr;
    const auto t_launch = std::chrono::steady_clock::now();
    bool finished = false;
    int32_t late = 0;                                         // exchanges whose turn had passed by a whole interval when they were issued
    for (int32_t r = 1; r < S; ++r) {
        const unsigned long long target = (unsigned long long)(tickets * (int64_t)r / S), next = (unsigned long long)(tickets * (int64_t)(r + 1) / S);
        while (!finished) {
            GE_HIP(hipMemcpyAsync(s->progress, counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->hub_side));
            GE_HIP(hipStreamSynchronize(s->hub_side));
            if (*s->progress >= target) break;
            const hipError_t q = hipEventQuery(done);         // (a kernel that ended early for any reason must not leave this loop spinning)
            if (q == hipSuccess) { finished = true; break; }
            if (q != hipErrorNotReady) return ge::fail(GE_ERR_HIP, "ge_sync_epoch: %s", hipGetErrorString(q));
            std::this_thread::sleep_for(std::chrono::microseconds(60));
        }
        if (finished || *s->progress >= next) ++late;
        const auto t_issue = std::chrono::steady_clock::now();
        if ((st = live_exchange(s)) != GE_OK) return st;
        if (dbg) {
            (void)hipStreamSynchronize(s->hub_side);
            std::fprintf(stderr, "[ge_sync_epoch live] rank %d exchange %d of %d: issued %.2f ms after the launch at ticket %llu of %lld (turn at %llu)%s, took %.2f ms\n", s->cfg.rank, r, S - 1,
                         std::chrono::duration<double, std::milli>(t_issue - t_launch).count(), (unsigned long long)*s->progress, (long long)tickets, target, finished ? ", kernel finished" : "",
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_issue).count());
        }
    }
    GE_HIP(hipStreamSynchronize(s->hub_side));                // the last land is in the tables before the exact exchange reads them
    if ((st = hub_exchange(s)) != GE_OK) return st;           // (on the handle's stream: behind the epoch kernel)
    if ((st = ge::glove_epoch_finish(s->h, cost_sum)) != GE_OK) return st;
    // The live form never holds the epoch kernel back, so exchanges that cannot keep the pace (a slow transport, an epoch of a few
    // milliseconds) would quietly reconcile the hub rows less often than planned -- which is what makes a sharded run unstable.  The
    // ranks therefore vote after every live epoch: a quarter of the exchanges late on half of the ranks in two epochs running, and the run continues
    // in segments (the epoch then waits for every exchange).  One word, summed, on the exact exchange's path.
    const float mine_late = (late * 4 > S) ? 1.0f : 0.0f;
    GE_HIP(hipMemcpyAsync(s->hub_buf, &mine_late, sizeof(float), hipMemcpyHostToDevice, s->main));
    GE_HIP(hipStreamSynchronize(s->main));
    if ((st = allreduce_f32_small(s, s->hub_buf, 1)) != GE_OK) return st;
    float any_late = 0.0f;
    GE_HIP(hipMemcpyAsync(&any_late, s->hub_buf, sizeof(float), hipMemcpyDeviceToHost, s->main));
    GE_HIP(hipStreamSynchronize(s->main));
    s->live_epochs += 1; s->live_late += late;
    // (half of the ranks or more: a single rank whose shard is small ends its kernel early and issues its exchanges "late" without any harm --
    // a transport that is too slow makes every rank late)
    s->late_streak = any_late * 2.0f >= (float)s->cfg.world ? s->late_streak + 1 : 0;
    if (s->late_streak >= 2) {                                // (one such epoch does no harm -- a run needs several epochs of too few exchanges to leave its track)
        s->late_streak = 0;
        if (s->live_cap > 64 && S > 64) {                     // first: no more live exchanges per epoch than the segmented form would make
            s->live_cap = 64;
            if (s->cfg.rank == 0) std::fprintf(stderr, "geglove: the live exchange of the hub rows fell behind the epoch (%d of %d late on this rank); at most 64 per epoch from here on\n", late, S - 1);
        } else {
            s->live = false;
            if (s->cfg.rank == 0) std::fprintf(stderr, "geglove: the live exchange of the hub rows fell behind the epoch (%d of %d late on this rank); continuing with the epoch in segments\n", late, S - 1);
        }
    }
    return GE_OK;
}
// the hub rows of this run (ascending) and one small exchange of them on demand: for hosts that cut their epochs themselves, and for the parity test
static ge_status ge_sync_hub_rows_impl(ge_sync *s, int32_t *out, int32_t capacity, int32_t *count) {
    if (!s || !count) return ge::fail(GE_ERR_ARG, "null argument");
    *count = s->n_hub;
    if (out && capacity > 0 && s->n_hub > 0) {
        GE_HIP(hipSetDevice(s->device));
        GE_HIP(hipMemcpy(out, s->hub_list, sizeof(int32_t) * (size_t)std::min(capacity, s->n_hub), hipMemcpyDeviceToHost));
    }
    return GE_OK;
}
ge_status ge_sync_hub_rows(ge_sync *s, int32_t *out, int32_t capacity, int32_t *count) { GE_GUARD(ge_sync_hub_rows_impl(s, out, capacity, count)); }
static ge_status hub_exchange_guarded(ge_sync *s) {
    try {
        if (!s) return ge::fail(GE_ERR_ARG, "null ge_sync handle");
        if (s->cfg.world == 1) return GE_OK;
        GE_HIP(hipSetDevice(s->device));
        return hub_exchange(s);
    } catch (const std::exception &e) { return ge::fail(GE_ERR_STATE, "internal error: %s", e.what()); }
}
ge_status ge_sync_hub_exchange(ge_sync *s) { return with_abort(s, hub_exchange_guarded(s)); }
static ge_status live_exchange_guarded(ge_sync *s) {
    try {
        if (!s) return ge::fail(GE_ERR_ARG, "null ge_sync handle");
        if (s->cfg.world == 1) return GE_OK;
        if (!s->live) return ge::fail(GE_ERR_STATE, "this run has no live hub rows (bf16 rows, a host transport, GE_SYNC_EPOCH=segments, or no column that is a hub on every rank)");
        GE_HIP(hipSetDevice(s->device));
        ge_status st = live_exchange(s);
        if (st == GE_OK) GE_HIP(hipStreamSynchronize(s->hub_side));
        return st;
    } catch (const std::exception &e) { return ge::fail(GE_ERR_STATE, "internal error: %s", e.what()); }
}
ge_status ge_sync_hub_exchange_live(ge_sync *s) { return with_abort(s, live_exchange_guarded(s)); }
ge_status ge_sync_live_rows(ge_sync *s, int32_t *out, int32_t capacity, int32_t *count) {
    if (!s || !count) return ge::fail(GE_ERR_ARG, "null argument");
    *count = s->live ? s->n_live : 0;
    if (out && capacity > 0 && *count > 0) {
        if (hipSetDevice(s->device) != hipSuccess || hipMemcpy(out, s->live_list, sizeof(int32_t) * (size_t)std::min(capacity, *count), hipMemcpyDeviceToHost) != hipSuccess)
            return ge::fail(GE_ERR_HIP, "ge_sync_live_rows: copy failed");
    }
    return GE_OK;
}
ge_status ge_sync_hub_plan(ge_sync *s, int32_t segments, int32_t *live, int32_t *exchanges, int32_t *live_rows) {
    if (!s) return ge::fail(GE_ERR_ARG, "null ge_sync handle");
    bool lv = false; int32_t n = 0;
    if (s->cfg.world > 1 && s->n_hub > 0) hub_plan(s, segments, &lv, &n);
    if (live) *live = lv ? 1 : 0;
    if (exchanges) *exchanges = n;
    if (live_rows) *live_rows = lv ? s->n_live : 0;
    return GE_OK;
}

static ge_status epoch_guarded(ge_sync *s, int32_t iteration, int32_t segments, double *cost_sum) { GE_GUARD(ge_sync_epoch_impl(s, iteration, segments, cost_sum)); }
ge_status ge_sync_epoch(ge_sync *s, int32_t iteration, int32_t segments, double *cost_sum) { return with_abort(s, epoch_guarded(s, iteration, segments, cost_sum)); }

static ge_status ge_sync_replicate_impl(ge_sync *s, int32_t src) {
    if (!s) return ge::fail(GE_ERR_ARG, "null ge_sync handle");
    if (s->cfg.world == 1) return GE_OK;
    if (src < 0 || src >= s->cfg.world) return ge::fail(GE_ERR_ARG, "replicate: src %d outside [0,%d)", src, s->cfg.world);
    ge_status st = turn(s, true, true, true);                            // land what is in flight, send everything not sent yet
    if (st == GE_OK) st = turn(s, true, false, false);
    if (st != GE_OK) return st;
    // every rank takes rank src's replica: what still differs is the rounding of own deltas.  (bf16 rows live partly in
    // per-rank fp32 master rows -- hub sets differ per rank -- and are left as landed, equal up to bf16 rounding.)
    for (Entry &e : s->ent) {
        if (e.bf16_rows) continue;
        float *stage = (float *)e.wire;                                   // dense fp32 staging; a bf16 wire buffer is too small
        float *tmp = nullptr;
        if (e.w16) { GE_HIP(hipMalloc((void **)&tmp, sizeof(float) * (size_t)e.n)); stage = tmp; }
        hipLaunchKernelGGL(k_gather, dim3(grid_for(e.n, s->cus)), dim3(256), 0, s->main, e.table, e.t_stride, e.cols, e.n, stage);
        ge_status r = GE_OK;
        if (s->loop) {
            if (hipStreamSynchronize(s->main) != hipSuccess) r = ge::fail(GE_ERR_HIP, "replicate: stream synchronize failed");
            else r = local_allreduce(s->loop, s->cfg.rank, stage, e.n, GE_DTYPE_F32, true, src);
        } else if (!s->tr.start) {
            hipError_t he = hipEventRecord(s->ev_taken, s->main);
            if (he == hipSuccess) he = hipStreamWaitEvent(s->side, s->ev_taken, 0);
            ncclResult_t nr = he == hipSuccess ? rccl().Broadcast(stage, stage, (size_t)e.n, ncclFloat32, src, s->comm, s->side) : ncclSuccess;
            if (he == hipSuccess && nr == ncclSuccess) { he = hipEventRecord(s->ev_reduced, s->side); if (he == hipSuccess) he = hipStreamWaitEvent(s->main, s->ev_reduced, 0); }
            if (he != hipSuccess || nr != ncclSuccess) r = ge::fail(GE_ERR_HIP, "replicate: broadcast of %s failed", e.name);
        } else {
            if (hipStreamSynchronize(s->main) != hipSuccess) r = ge::fail(GE_ERR_HIP, "replicate: stream synchronize failed");
            else if ((r = s->tr.broadcast(s->tr.user, stage, e.n, GE_DTYPE_F32, src)) != GE_OK) r = ge::fail(r, "transport.broadcast failed for %s", e.name);
        }
        if (r == GE_OK) hipLaunchKernelGGL(k_scatter2, dim3(grid_for(e.n, s->cus)), dim3(256), 0, s->main, e.table, e.t_stride, e.cols, e.n, stage, e.base);
        hipError_t he = hipStreamSynchronize(s->main);
        if (tmp) (void)hipFree(tmp);
        if (r != GE_OK) return r;
        if (he != hipSuccess) return ge::fail(GE_ERR_HIP, "replicate: %s", hipGetErrorString(he));
    }
    return GE_OK;
}
static ge_status replicate_guarded(ge_sync *s, int32_t src) { GE_GUARD(ge_sync_replicate_impl(s, src)); }
ge_status ge_sync_replicate(ge_sync *s, int32_t src) { return with_abort(s, replicate_guarded(s, src)); }

// Host scalars (the epoch's cost sums, a max over shards) over the same transport, so that a host without a
// collective library of its own (the Java module) needs nothing else.  op: 0 = sum, 1 = max.  Blocking.
static ge_status ge_sync_allreduce_f64_impl(ge_sync *s, double *values, int32_t n, int32_t op) {
    if (!s || !values || n < 0) return ge::fail(GE_ERR_ARG, "invalid argument");
    if (s->cfg.world == 1 || n == 0) return GE_OK;
    if (op != 0 && op != 1) return ge::fail(GE_ERR_ARG, "op must be 0 (sum) or 1 (max)");
    if (s->loop) return local_allreduce(s->loop, s->cfg.rank, values, n, 2, false, op);
    if (s->tr.start) return ge::fail(GE_ERR_STATE, "ge_sync_allreduce_f64 runs over RCCL; a host that brought its own transport reduces its scalars there");
    if (op != 0 && op != 1) return ge::fail(GE_ERR_ARG, "op must be 0 (sum) or 1 (max)");
    GE_HIP(hipSetDevice(s->device));
    double *d = nullptr;
    GE_HIP(hipMalloc((void **)&d, sizeof(double) * (size_t)n));
    hipError_t he = hipMemcpyAsync(d, values, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s->side);
    ncclResult_t nr = he == hipSuccess ? rccl().AllReduce(d, d, (size_t)n, ncclFloat64, op == 0 ? ncclSum : ncclMax, s->comm, s->side) : ncclSuccess;
    if (he == hipSuccess && nr == ncclSuccess) he = hipMemcpyAsync(values, d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s->side);
    if (he == hipSuccess && nr == ncclSuccess) he = hipStreamSynchronize(s->side);
    (void)hipFree(d);
    if (nr != ncclSuccess) return ge::fail(GE_ERR_HIP, "ncclAllReduce failed: %s", rccl().GetErrorString(nr));
    if (he != hipSuccess) return ge::fail(GE_ERR_HIP, "scalar all-reduce: %s", hipGetErrorString(he));
    return GE_OK;
}
static ge_status allreduce_guarded(ge_sync *s, double *values, int32_t n, int32_t op) { GE_GUARD(ge_sync_allreduce_f64_impl(s, values, n, op)); }
ge_status ge_sync_allreduce_f64(ge_sync *s, double *values, int32_t n, int32_t op) { return with_abort(s, allreduce_guarded(s, values, n, op)); }

}  // extern "C"
