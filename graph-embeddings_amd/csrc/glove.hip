// glove.hip -- GloVe / pGloVe AdaGrad trainer for MI355X (gfx950): kernels + C ABI.
//
// Reference semantics (J/ = src/main/java/org/uu/nl/embedding/ of Phaken/graph-embeddings):
//   Optimizer ctor           J/opt/Optimizer.java:34-64     -> k_init_java / k_fill_rows
//   Adagrad.createJob        J/opt/grad/Adagrad.java:42-98  -> k_adagrad_exact (bit-exact) / k_adagrad_runs (Hogwild)
//   GloveCost / PGloveCost   J/opt/GloveCost.java:7-20, J/opt/PGloveCost.java:7-20 -> cost_terms()
//   Optimizer.extractResult  J/opt/Optimizer.java:129-140   -> k_extract
//
// This file is compiled with -ffp-contract=off: the exact kernel must not fuse a*b+c
// (Java never does); the Hogwild kernel asks for FMAs explicitly where it wants them.
//
// Roofline: HBM.  SURVEY.md 8(d) counts read 16*D+28 / write 16*D+16 bytes per pair-update (both row pairs through HBM);
// k_adagrad_runs keeps one pair in registers per run and has to move 20 + 4 * row bytes per nonzero + 4 * row bytes per
// run (ge_glove_info.schedule_bytes).  No MFMA: sparse gather + length-D dot.

#include "ge_common.h"
#include "ge_javarand.h"
#include "ge_cost.h"
#include "ge_layout.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <new>
#include <type_traits>
#include <vector>

namespace {

enum { ORDER_IDENTITY = 0, ORDER_PERM = 1, ORDER_BIJECTION = 2 };

struct GloveParams {
    float *focus, *context, *fbias, *cbias;
    float *gsf, *gsc, *gsfb, *gscb;          // Adagrad.gradSq* | Adam/AMSGrad.M1*
    float *m2f, *m2c, *m2fb, *m2cb;          // Adam/AMSGrad.M2*
    double correction;                       // Adam: lr*sqrt(1-beta2^(t+1))/(1-beta1^(t+1))  (Adam.java:84), per epoch
    int32_t opt;                             // GE_OPT_*
    float *hub32;                            // bf16 embeddings: fp32 master rows of the hub columns [n_hub x D]
    const int32_t *hub_index;                // bf16 embeddings: column -> row of hub32, -1 for ordinary columns
    const int32_t *I, *J;
    const float *X;
    const int32_t *perm;
    const double *L;       // Hogwild: per-nonzero log term  (k_cost_terms)
    const float *W;        // Hogwild: per-nonzero weight
    const int32_t *bA, *bB;   // Hogwild, blocked order: resident / streamed row id per re-ordered position
    int64_t n_chunks;         // chunks of <= RUN_CHUNK nonzeros per epoch
    int64_t ticket_end;       // this launch hands out the tickets below this one (n_chunks: the whole epoch; less: one segment of it)
    int64_t n_hchunks;        // blocked order: the first n_hchunks chunks are hub columns (column-major)
    int32_t blocked;          // 1: chunk c = positions [cstart[c], cstart[c+1]) of the re-ordered arrays (ge_layout.h)
    int32_t hot_enabled;      // blocked order: hub chunks publish deltas with atomics
    int32_t flush_every;      // a hub run publishes its delta at least every this many nonzeros (general order)
    const int32_t *cstart;    // blocked order: first position of every chunk (n_chunks + 1 entries)
    const int32_t *cmeta;     // blocked order: hub chunk: flush limit of its runs; row chunk: the row that publishes by delta, or -1
    double *cost_out;      // Hogwild: one double accumulator
    unsigned long long *queue;   // Hogwild: next chunk to hand out (zeroed before every launch)
    double xmax;
    int64_t N;
    int32_t D;
    int32_t DS;            // row stride of the fp32 row tables in floats (>= RW; a multiple of RW when a row and its accumulator rows interleave)
    int32_t RW;            // row width in floats: D, or D + 4 when a row carries its bias at [D] (fat rows, Hogwild)
    int32_t ES;            // bf16 rows: bf16 elements between consecutive embedding rows (D, or 2 * DS when a row shares a record with its accumulator row)
    int32_t cost_kind;
    float lr;
    int32_t order_mode;
    uint32_t bij_mask, bij_shift;
    uint32_t bij_key[4];
};

// The Hogwild kernel reads (l, w) instead of X: X never changes, so the fp64 log / sqrt run once
// per nonzero at create time (one lane per nonzero) instead of once per update, and the update
// kernel keeps its registers for rows in flight.  +8 bytes read per update (0.1 % at D=200).
__global__ void k_cost_terms(const float *X, int64_t n, int kind, double xmax, double *L, float *W) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < n; k += stride) {
        double l; float w;
        cost_terms<false>(kind, X[k], xmax, l, w);
        L[k] = l; W[k] = w;
    }
}

// ---- epoch order ----------------------------------------------------------------------
// Keyed bijection of [0, 2^b) (odd multiply, xor-shift, add key: each step invertible),
// cycle-walked into [0, N).  2^b < 2N so the expected number of rounds is < 2.
__host__ __device__ __forceinline__ uint32_t bij_round(uint32_t x, const GloveParams &p) {
    const uint32_t m = p.bij_mask, s = p.bij_shift;
    x = (x + p.bij_key[0]) & m;  x = (x * 0x9E3779B1u) & m;  x ^= x >> s;
    x = (x + p.bij_key[1]) & m;  x = (x * 0x85EBCA6Bu) & m;  x ^= x >> s;
    x = (x + p.bij_key[2]) & m;  x = (x * 0xC2B2AE35u) & m;  x ^= x >> s;
    x = (x + p.bij_key[3]) & m;  x = (x * 0x27D4EB2Fu) & m;  x ^= x >> s;
    return x;
}
__device__ __forceinline__ int64_t map_index(const GloveParams &p, int64_t k) {
    if (p.order_mode == ORDER_PERM) return p.perm[k];
    if (p.order_mode == ORDER_BIJECTION) {
        uint32_t x = (uint32_t)k;
        do { x = bij_round(x, p); } while ((int64_t)x >= p.N);
        return x;
    }
    return k;
}

// ---- init -----------------------------------------------------------------------------
// Optimizer ctor, J/opt/Optimizer.java:50-57: per row i the draws are fBias, cBias, then
// focus[i,d], context[i,d] interleaved; value = (float)(nextFloat() - 0.5) / dimension.
// Row i starts (2+2D)*i draws into the stream: jump the LCG there, then run sequentially.
// The tables are written in their final layout (no dense staging copy): a side whose pointer is null is skipped (a
// sharded handle initialises all V context rows and only its own focus rows), `stride` = floats (bf16: elements)
// between rows, bias_col >= 0 = fat rows (bias at [bias_col], the padding behind it zeroed), else the bias vectors;
// ROW16 = bf16 storage (round to nearest even), whose hub columns also get their fp32 master row.
__device__ __forceinline__ uint16_t bf16_rne(float f) {
    const uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
template <bool ROW16>
__global__ void k_init_java(void *focus_, void *context_, float *fbias, float *cbias, int64_t bias_stride, int32_t focus_row0,
                            int32_t row0, int32_t rows, int32_t D, int64_t stride, int32_t bias_col, int32_t row_width,
                            uint64_t seed_state, const int32_t *hub_index, float *hub32) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int64_t i = (int64_t)row0 + r;
    ge::JavaRandom rng{ge::JavaRandom::jump(seed_state, (uint64_t)i * (uint64_t)(2 + 2 * D))};
    const float fd = (float)D;
    const float fb = (float)((double)rng.next_float() - 0.5) / fd;
    const float cb = (float)((double)rng.next_float() - 0.5) / fd;
    using RT = typename std::conditional<ROW16, uint16_t, float>::type;
    RT *f = focus_ ? reinterpret_cast<RT *>(focus_) + (i - focus_row0) * stride : nullptr;
    RT *c = context_ ? reinterpret_cast<RT *>(context_) + i * stride : nullptr;
    float *hub = (ROW16 && c && hub32 && hub_index[i] >= 0) ? hub32 + (int64_t)hub_index[i] * D : nullptr;
    for (int32_t d = 0; d < D; ++d) {
        const float fv = (float)((double)rng.next_float() - 0.5) / fd;
        const float cv = (float)((double)rng.next_float() - 0.5) / fd;
        if constexpr (ROW16) { if (f) f[d] = bf16_rne(fv); if (c) c[d] = bf16_rne(cv); if (hub) hub[d] = cv; }
        else { if (f) f[d] = fv; if (c) c[d] = cv; }
    }
    if constexpr (!ROW16) {
        if (bias_col >= 0) {
            if (f) { f[bias_col] = fb; for (int32_t d = bias_col + 1; d < row_width; ++d) f[d] = 0.0f; }
            if (c) { c[bias_col] = cb; for (int32_t d = bias_col + 1; d < row_width; ++d) c[d] = 0.0f; }
        }
    }
    if (bias_col < 0) {
        if (f) fbias[(i - focus_row0) * bias_stride] = fb;
        if (c) cbias[i * bias_stride] = cb;
    }
}

__global__ void k_fill(float *p, int64_t n, float v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
// rows of an accumulator table in its final layout: columns [0, valid) = v, [valid, width) = 0 (fat-row padding)
__global__ void k_fill_rows(float *p, int64_t rows, int64_t stride, int32_t valid, int32_t width, float v) {
    const int64_t n = rows * width, step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        const int64_t r = i / width; const int32_t c = (int32_t)(i - r * width);
        p[r * stride + c] = c < valid ? v : 0.0f;
    }
}

// Optimizer.extractResult: (focus + context) / 2 in fp32, widened on store for the f64 form.
template <typename OUT>
__global__ void k_extract(const float *focus, const float *context, OUT *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (OUT)((focus[i] + context[i]) / 2.0f);
}

// Fat rows (fp32 Hogwild tables): [row(D) | bias | 3 x 0].  Columns [col0, col0+ncols) of `rows` fat rows <-> a dense
// [rows x ncols] array: ncols = D, col0 = 0 is the row table as the API shows it, ncols = 1, col0 = D its bias vector.
__global__ void k_fat_gather(const float *fat, int64_t rows, int32_t DS, int32_t col0, int32_t ncols, float *dense) {
    const int64_t n = rows * ncols, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / ncols; const int32_t c = (int32_t)(i - r * ncols);
        dense[i] = fat[r * DS + col0 + c];
    }
}
__global__ void k_fat_scatter(float *fat, int64_t rows, int32_t DS, int32_t col0, int32_t ncols, const float *dense) {
    const int64_t n = rows * ncols, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / ncols; const int32_t c = (int32_t)(i - r * ncols);
        fat[r * DS + col0 + c] = dense[i];
    }
}

// bf16 embeddings (BASELINE config C5): storage conversions.  Round-to-nearest-even on the way in (init, set_state).
// dense fp32 [rows x D] <-> bf16 rows `es` elements apart
__global__ void k_f32_to_bf16(const float *src, uint16_t *dst, int64_t n, int32_t D, int64_t es) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t u = __builtin_bit_cast(uint32_t, src[i]);
        const int64_t r = i / D;
        dst[r * es + (i - r * D)] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
}
__global__ void k_bf16_to_f32(const uint16_t *src, float *dst, int64_t n, int32_t D, int64_t es) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { const int64_t r = i / D; dst[i] = __builtin_bit_cast(float, (uint32_t)src[r * es + (i - r * D)] << 16); }
}
// hub rows live in the fp32 master table: dir 0 = full[col] -> hub32[idx], dir 1 = hub32[idx] -> full[col]
__global__ void k_hub_rows(float *full, float *hub32, const int32_t *hub_index, int32_t V, int32_t D, int dir) {
    const int32_t v = blockIdx.x;
    const int32_t h = hub_index[v];
    if (h < 0) return;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        if (dir == 0) hub32[(int64_t)h * D + d] = full[(int64_t)v * D + d];
        else full[(int64_t)v * D + d] = hub32[(int64_t)h * D + d];
    }
}

// ---- exact (deterministic) kernel -------------------------------------------------------
// One wavefront walks nonzeros [k_begin, k_end) strictly in order and reproduces the Java
// arithmetic bit for bit: products in fp32, summed sequentially in d order (GloveCost.java:10-11),
// fp64 sqrt/div in the update (Adagrad.java:76-77,88-89), per-job fp32 cost accumulator (:60).
__global__ __launch_bounds__(64) void k_adagrad_exact(GloveParams p, int64_t k_begin, int64_t k_end,
                                                      float *job_cost) {
    extern __shared__ float s_prod[];
    const int lane = threadIdx.x;
    const int32_t D = p.D;
    const double lr = (double)p.lr;
    float cost = 0.0f;
    for (int64_t k = k_begin; k < k_end; ++k) {
        const int64_t idx = map_index(p, k);
        const int32_t bu = p.I[idx], bv = p.J[idx];
        const float x = p.X[idx];
        float *foc = p.focus + (int64_t)bu * D, *ctx = p.context + (int64_t)bv * D;
        float *g1s = p.gsf + (int64_t)bu * D,   *g2s = p.gsc + (int64_t)bv * D;
        for (int32_t d = lane; d < D; d += 64) s_prod[d] = foc[d] * ctx[d];
        __syncthreads();
        float ic = 0.0f;
        for (int32_t d = 0; d < D; ++d) ic = ic + s_prod[d];
        double l; float w;
        cost_terms<true>(p.cost_kind, x, p.xmax, l, w);
        ic = (float)((double)ic + ((double)(p.fbias[bu] + p.cbias[bv]) - l));
        float wc = w * ic;
        cost = (float)((double)cost + (0.5 * (double)wc) * (double)ic);
        __syncthreads();
        if (p.opt == GE_OPT_ADAGRAD) {
            for (int32_t d = lane; d < D; d += 64) {
                const float f = foc[d], c = ctx[d];
                const float grad1 = wc * c;
                const float grad2 = wc * f;
                foc[d] = (float)((double)f - ((double)grad1 / sqrt((double)g1s[d])) * lr);
                ctx[d] = (float)((double)c - ((double)grad2 / sqrt((double)g2s[d])) * lr);
                g1s[d] = g1s[d] + grad1 * grad1;
                g2s[d] = g2s[d] + grad2 * grad2;
            }
            if (lane == 0) {
                p.fbias[bu] = (float)((double)p.fbias[bu] - (double)wc / sqrt((double)p.gsfb[bu]));
                p.cbias[bv] = (float)((double)p.cbias[bv] - (double)wc / sqrt((double)p.gscb[bv]));
                wc = wc * wc;
                p.gsfb[bu] = p.gsfb[bu] + wc;
                p.gscb[bv] = p.gscb[bv] + wc;
            }
        } else {
            // Adam.java:103-145 / AMSGrad.java:117-160.  All moment arithmetic is fp32 (beta1, 1-beta1, ... are floats),
            // the parameter step goes through fp64 exactly as `focus[d1] -= correction * m1 / (sqrt(v1) + epsilon)`.
            const bool ams = p.opt == GE_OPT_AMSGRAD;
            const float beta1 = 0.9f, beta2 = 0.999f, epsilon = 1e-7f;
            const float omb1 = 1 - beta1, omb2 = 1 - beta2;
            float *m2f = p.m2f + (int64_t)bu * D, *m2c = p.m2c + (int64_t)bv * D;
            auto fmaxj = [](float a, float b) { return (a <= b) ? b : ((a + b) != (a + b) ? __builtin_nanf("") : a); };   // FastMath.max
            auto step = [&](float par, float m, float v) -> float {
                return ams ? (float)((double)par - lr / (sqrt((double)v) + (double)epsilon) * (double)m)
                           : (float)((double)par - p.correction * (double)m / (sqrt((double)v) + (double)epsilon));
            };
            for (int32_t d = lane; d < D; d += 64) {
                const float f = foc[d], c = ctx[d];
                const float grad_u = wc * c, grad_v = wc * f;
                const float m1 = beta1 * g1s[d] + omb1 * grad_u;
                const float m2 = beta1 * g2s[d] + omb1 * grad_v;
                float v1 = beta2 * m2f[d] + omb2 * (grad_u * grad_u);
                float v2 = beta2 * m2c[d] + omb2 * (grad_v * grad_v);
                if (ams) { v1 = fmaxj(m2f[d], v1); v2 = fmaxj(m2c[d], v2); }
                foc[d] = step(f, m1, v1);
                ctx[d] = step(c, m2, v2);
                g1s[d] = m1; g2s[d] = m2; m2f[d] = v1; m2c[d] = v2;
            }
            if (lane == 0) {
                const float m1 = beta1 * p.gsfb[bu] + omb1 * wc;
                const float m2 = beta1 * p.gscb[bv] + omb1 * wc;
                float v1 = beta2 * p.m2fb[bu] + omb2 * (wc * wc);
                float v2 = beta2 * p.m2cb[bv] + omb2 * (wc * wc);
                if (ams) { v1 = fmaxj(p.m2fb[bu], v1); v2 = fmaxj(p.m2cb[bv], v2); }
                p.fbias[bu] = step(p.fbias[bu], m1, v1);
                p.cbias[bv] = step(p.cbias[bv], m2, v2);
                p.gsfb[bu] = m1; p.gscb[bv] = m2; p.m2fb[bu] = v1; p.m2cb[bv] = v2;
            }
        }
        __syncthreads();   // the next nonzero may read what this one wrote (same wave, program order)
    }
    if (lane == 0) *job_cost = cost;
}

// ---- Hogwild kernel -----------------------------------------------------------------------
// One wavefront = one sequential worker, like one Java pool thread (Adagrad.createJob walks its
// slice in order).  A worker takes a chunk of 128 nonzeros of the epoch order, sorts it by column
// in registers (bitonic network, no LDS, no barrier) and walks it sequentially:
//   * a RUN of equal j keeps context[j], gradSqContext[j], cBias[j], gradSqCBias[j] in registers --
//     loaded once, updated in place with exact sequential semantics inside the run, written once;
//   * the other side of every nonzero is streamed: 16-byte-per-lane buffer loads (row base in SGPRs,
//     hardware bounds check masks the lanes past the row), the fp32 dot is wave-reduced, the fused
//     AdaGrad update is written straight back.  The NEXT nonzero's streamed rows are requested into LDS
//     (buffer_load ... lds) before the current one waits for its own, so a request has a whole step to
//     complete (see s_setA / s_setB below for why LDS and not registers).
// (Blocked layout, the default: the resident side is the hub COLUMN in hub chunks and the focus ROW elsewhere.)
// Workers never lock (Hogwild).  ~10^4 workers are in flight where the JVM has <= #cores, so for
// HOT columns (hub nodes that sit in a large share of the nonzeros) a plain read-modify-write
// would lose most concurrent updates and training stalls (measured): their runs read
// the row with agent-coherent (sc1) loads and publish the run's DELTA with float atomic adds.
// Run-combining is what makes that affordable: same-address atomics serialise at ~25 ns per
// cache line at the memory side.
template <int VW> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<1> { using T = float; };

template <int VW> __device__ __forceinline__ float &comp(typename Vec<VW>::T &v, int c);
template <> __device__ __forceinline__ float &comp<4>(float4 &v, int c) { return (&v.x)[c]; }
template <> __device__ __forceinline__ float &comp<2>(float2 &v, int c) { return (&v.x)[c]; }
template <> __device__ __forceinline__ float &comp<1>(float &v, int) { return v; }

typedef int   ge_i4 __attribute__((ext_vector_type(4)));
typedef int   ge_i2 __attribute__((ext_vector_type(2)));

constexpr int AUX_PLAIN = 0;
constexpr int AUX_SC1 = 16;     // agent-coherent: bypasses the CU's L1, sees memory-side atomics

template <int VW, int AUX>
__device__ __forceinline__ typename Vec<VW>::T buf_load(__amdgpu_buffer_rsrc_t rs, int off) {
    typename Vec<VW>::T r;
    if constexpr (VW == 4) { ge_i4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX); __builtin_memcpy(&r, &v, 16); }
    else if constexpr (VW == 2) { ge_i2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, AUX); __builtin_memcpy(&r, &v, 8); }
    else { int v = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, AUX); __builtin_memcpy(&r, &v, 4); }
    return r;
}
template <int VW>
__device__ __forceinline__ void buf_store(typename Vec<VW>::T val, __amdgpu_buffer_rsrc_t rs, int off) {
    if constexpr (VW == 4) { ge_i4 v; __builtin_memcpy(&v, &val, 16); __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, AUX_SC1); }
    else if constexpr (VW == 2) { ge_i2 v; __builtin_memcpy(&v, &val, 8); __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, AUX_SC1); }
    else { int v; __builtin_memcpy(&v, &val, 4); __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, AUX_SC1); }
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t rs, int off, bool coherent) {
    int v = coherent ? __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, AUX_SC1)
                     : __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, AUX_PLAIN);
    return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Sum over the 64 lanes, result in every lane.  Four DPP adds reduce each row of 16 lanes
// (quad swaps, half-row mirror, row mirror), then the four row sums are read through SGPRs.
__device__ __forceinline__ float dpp_add(float v, int ctrl_sel) {
    const int iv = __builtin_bit_cast(int, v);
    int r;
    switch (ctrl_sel) {
        case 0:  r = __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xF, 0xF, false); break;    // quad_perm [1,0,3,2]
        case 1:  r = __builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
        case 2:  r = __builtin_amdgcn_update_dpp(0, iv, 0x141, 0xF, 0xF, false); break;   // row_half_mirror
        default: r = __builtin_amdgcn_update_dpp(0, iv, 0x140, 0xF, 0xF, false); break;   // row_mirror
    }
    return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add(v, 0); v = dpp_add(v, 1); v = dpp_add(v, 2); v = dpp_add(v, 3);
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// four bf16 values (8 bytes) widened to fp32
__device__ __forceinline__ float4 widen_bf16x4(float2 raw) {
    const uint32_t lo = __builtin_bit_cast(uint32_t, raw.x), hi = __builtin_bit_cast(uint32_t, raw.y);
    float4 r;
    r.x = __builtin_bit_cast(float, lo << 16); r.y = __builtin_bit_cast(float, lo & 0xFFFF0000u);
    r.z = __builtin_bit_cast(float, hi << 16); r.w = __builtin_bit_cast(float, hi & 0xFFFF0000u);
    return r;
}
#define GE_LDS(ptr) ((__attribute__((address_space(3))) void *)(ptr))
// one load-to-LDS wave instruction: lane L brings W bytes from offset `voff` of the resource to lds + L * W
template <int W> __device__ __forceinline__ void load_to_lds(__amdgpu_buffer_rsrc_t rs, float *lds, int voff) {
    static_assert(W == 16 || W == 4, "16-byte or 4-byte lanes");
    if constexpr (W == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, GE_LDS(lds), 16, voff, 0, 0, 16 /* AUX_SC1 */);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, GE_LDS(lds), 4, voff, 0, 0, 16 /* AUX_SC1 */);
}
// s_waitcnt immediate (gfx9 layout) that waits until at most n vector-memory instructions are outstanding and for nothing else
constexpr int wait_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

constexpr int RUN_CHUNK = 128;            // nonzeros per worker chunk (2 per lane)
constexpr int KEY_PAD = 0x7FFFFFFF;       // sorts last; marks the unused tail of the last chunk

// OPT: GE_OPT_ADAGRAD keeps one auxiliary row per side (gradSq), Adam / AMSGrad two (M1, M2).
// EMB16: the embedding rows (focus, context) are stored as bf16, everything else stays fp32 (BASELINE config C5).
//   Rows are widened to fp32 when loaded; a resident row stays fp32 in registers for its whole run and is
//   narrowed once, with STOCHASTIC rounding (an update of 1e-5 on a value of 0.1 is far below half a bf16 ulp and
//   would always round away).  Hub context rows keep an fp32 master copy (hub32) that their runs read and
//   publish into with the same atomics as in the fp32 build; they never touch the bf16 table during training.
template <int VW, int NCH, int OPT, bool EMB16, bool FAT>
__global__ __launch_bounds__(256, (NCH == 1 && VW == 4 && OPT == GE_OPT_ADAGRAD) ? 5 : 1) void k_adagrad_runs(GloveParams p, int32_t n_workers) {
    using VT = typename Vec<VW>::T;
    static_assert(!EMB16 || VW == 4, "bf16 embeddings need dim % 4 == 0");
    constexpr bool MOM = OPT != GE_OPT_ADAGRAD;
    // FAT rows (every fp32 Hogwild table): a row is D + 4 floats, element [D] is the row's bias (in the accumulator /
    // moment tables: the bias accumulator / moment), the rest padding that stays zero.  The bias rides in the sector
    // the row's tail already occupies, so a streamed update costs four row accesses and no 4-byte ones.
    // bf16 rows (EMB16) have no fp32 row to carry the bias: both scalars ride behind the ACCUMULATOR row,
    // [gradSq (D) | the bias accumulator | the bias | 2 x 0]; the bf16 row itself stays D elements wide.
    // (FAT is chosen on the host: fp32 rows whose bias lane fits the last register chunk -- a dimension that fills its 64-lane
    // chunks exactly would need one more chunk for that one lane and keeps the separate bias tables instead.)
    uint32_t sr_state = 0x9E3779B9u * (uint32_t)(threadIdx.x + 1) + (uint32_t)blockIdx.x * 0x85EBCA6Bu + p.bij_key[0];
    // embedding-row access: fp32 build = plain 16-byte vectors; bf16 build = 8 bytes widened / narrowed here
    auto emb_rsrc = [&](float *base, int64_t id) {
        if constexpr (EMB16) return make_rsrc(reinterpret_cast<uint16_t *>(base) + id * p.ES, (uint32_t)p.D * 2u);
        else return make_rsrc(base + id * p.DS, (uint32_t)p.RW * 4u);
    };
    auto emb_load = [&](__amdgpu_buffer_rsrc_t rs, int q) -> VT {
        if constexpr (EMB16) return widen_bf16x4(buf_load<2, AUX_SC1>(rs, ((threadIdx.x & 63) + q * 64) * 8));
        else return buf_load<VW, AUX_SC1>(rs, ((threadIdx.x & 63) + q * 64) * VW * 4);
    };
    auto emb_store = [&](VT v, __amdgpu_buffer_rsrc_t rs, int q) {
        if constexpr (EMB16) {
            auto narrow = [&](float f) -> uint32_t {      // stochastic rounding: add 16 random low bits, truncate
                sr_state = sr_state * 1664525u + 1013904223u;
                return (__builtin_bit_cast(uint32_t, f) + (sr_state >> 16)) >> 16;
            };
            float2 raw;
            raw.x = __builtin_bit_cast(float, narrow(v.x) | (narrow(v.y) << 16));
            raw.y = __builtin_bit_cast(float, narrow(v.z) | (narrow(v.w) << 16));
            buf_store<2>(raw, rs, ((threadIdx.x & 63) + q * 64) * 8);
        } else buf_store<VW>(v, rs, ((threadIdx.x & 63) + q * 64) * VW * 4);
    };
    constexpr float BETA1 = 0.9f, BETA2 = 0.999f, EPS = 1e-7f, OMB1 = 1 - BETA1, OMB2 = 1 - BETA2;   // Adam.java:45-53
    const float corr = OPT == GE_OPT_ADAM ? (float)p.correction : p.lr;      // Adam.java:84 | AMSGrad.java:133 uses lr itself
    // one element of an Adam / AMSGrad update in fp32: new moments and the parameter step
    auto moment_step = [&](float grad, float &m, float &v) -> float {
        m = __builtin_fmaf(BETA1, m, OMB1 * grad);
        const float vn = __builtin_fmaf(BETA2, v, OMB2 * (grad * grad));
        v = OPT == GE_OPT_AMSGRAD ? fmaxf(v, vn) : vn;
        // IEEE operations only (fma, mul, add, correctly rounded sqrt and division: HIP's default for fp32 without fast-math), so that
        // tests/kernel_model.py can restate this arithmetic bit for bit; __fsqrt_rn is the 1-ulp v_sqrt_f32 in this toolchain
        return corr * m * (1.0f / (__builtin_sqrtf(v) + EPS));
    };
    const int lane = threadIdx.x & 63;
    const int32_t D = p.D;
    const int32_t DS = p.DS;
    const uint32_t row_bytes = (uint32_t)p.RW * 4u;
    const int bl_q = (D / VW) >> 6, bl_lane = (D / VW) & 63;     // FAT: the lane and register chunk that hold element [D]
    // FAT: which row (parameter or accumulator) and which component of that lane's vector hold the bias; its accumulator is
    // component 0 of the accumulator row's lane in both layouts
    constexpr bool BIAS_IN_ACC = EMB16;
    constexpr int BIAS_C = EMB16 ? 1 : 0;
    const float lr = p.lr;
    const int wave = rfl((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (wave >= n_workers) return;          // workers pull chunks of the epoch order from one queue
    const int64_t n_chunks = p.n_chunks;
    __shared__ float s_tr[4][2][NCH * 64 * VW];       // per-wave strip for the atomic flush (no block barrier)
    // The streamed rows of the NEXT nonzero land in LDS, not in registers (buffer_load ... lds, 16 bytes per lane): nothing
    // in the register file depends on the load, so it stays in flight across the current nonzero's arithmetic and stores
    // and is waited for only where the next step reads it (with loads into registers the compiler put the copies that merge
    // the two steps' values -- and with them the wait -- right behind the load, and every nonzero paid the full latency).
    // Two images per wave, one per array, so that the wait for one never covers the load into the other.
    // Rows that are multiples of 16 bytes travel 16 bytes per lane; the others (VW < 4) 4 bytes per lane, VW instructions per register chunk.
    constexpr int DMAW = VW == 4 ? 16 : 4;                                            // bytes per lane of one load-to-LDS
    constexpr int N_Q = VW == 4 ? 1 : VW;                                              // such loads per register chunk
    constexpr int IMG = 64 * VW;                                                       // floats per register chunk's image
    constexpr int IMG_R = EMB16 ? ((NCH + 1) / 2) * IMG : NCH * IMG;               // a bf16 row takes half the bytes
    constexpr int IMG_G = IMG_R, IMG_H = IMG_G + NCH * IMG;                          // offsets of the accumulator / second-moment images
    constexpr int IMG_B = IMG_H + (MOM ? NCH * IMG : 0);                             // three 256-byte slots for the bias scalars (not FAT)
    constexpr int SET_FLOATS = IMG_B + (FAT ? 0 : 3 * 64);
    __shared__ float s_setA[4 * SET_FLOATS], s_setB[4 * SET_FLOATS];
    float *const setA = s_setA + (threadIdx.x >> 6) * SET_FLOATS, *const setB = s_setB + (threadIdx.x >> 6) * SET_FLOATS;
    double cost_acc = 0.0;
    bool inr[NCH];                          // lane holds real elements of a row (D % VW == 0)
#pragma unroll
    for (int q = 0; q < NCH; ++q) inr[q] = (lane + q * 64) * VW < D;

    for (;;) {
        unsigned long long ticket = 0;
        if (lane == 0) ticket = atomicAdd(p.queue, 1ull);
        const int64_t tk = ((int64_t)(unsigned)rfl((int)(ticket >> 32)) << 32) | (unsigned)rfl((int)(ticket & 0xFFFFFFFFull));
        if (tk >= p.ticket_end) break;          // (the queue starts at the segment's first ticket, glove_epoch_segment)

        // ---- stage: two nonzeros per lane ------------------------------------------------------
        // key = id of the RESIDENT row (sorted on, kept in registers across a run), oth = id of the
        // STREAMED row.  Blocked order: chunk c of the re-ordered arrays; chunks [0, n_hchunks) are the
        // hub columns in column-major order (resident = context row, hot), the others are the rest of
        // the matrix in row-major order (resident = focus row).  General order (Java permutation /
        // matrix order): the resident side is always the context row, hub columns are keyed ~j.
        int32_t key[2], slot[2], oth[2];
        float ww[2];
        double ll[2];
        int64_t chunk = tk;
        bool res_is_ctx = true;
        int32_t c_lo = 0, c_len = 0, c_meta = -1;
        if (p.blocked) {
            uint32_t x = (uint32_t)tk;
            if (p.order_mode == ORDER_BIJECTION) { do { x = bij_round(x, p); } while ((int64_t)x >= n_chunks); }
            chunk = rfl((int)x);
            res_is_ctx = chunk < p.n_hchunks;
            c_lo = rfl(p.cstart[chunk]); c_len = rfl(p.cstart[chunk + 1]) - c_lo; c_meta = rfl(p.cmeta[chunk]);
        }
        // a hub run is cut every flush_lim nonzeros; a long row's piece (c_meta = its id) runs whole
        const int32_t flush_lim = p.blocked ? (res_is_ctx ? c_meta : RUN_CHUNK) : p.flush_every;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t k = (p.blocked ? (int64_t)c_lo : chunk * RUN_CHUNK) + q * 64 + lane;
            key[q] = KEY_PAD; oth[q] = 0; ww[q] = 0.0f; ll[q] = 0.0;
            slot[q] = q * 64 + lane;
            if (p.blocked) {
                if (q * 64 + lane < c_len) { key[q] = p.bA[k]; oth[q] = p.bB[k]; ww[q] = p.W[k]; ll[q] = p.L[k]; }
            } else if (k < p.N) {
                const int64_t idx = map_index(p, k);
                key[q] = p.J[idx]; oth[q] = p.I[idx]; ww[q] = p.W[idx]; ll[q] = p.L[idx];
            }
        }
        const int n_valid = rfl(__popcll(__ballot(key[0] != KEY_PAD)) + __popcll(__ballot(key[1] != KEY_PAD)));

        // ---- bitonic sort of (key, slot) over positions pos = q*64 + lane ---------------------
#pragma unroll
        for (int k2 = 2; k2 <= RUN_CHUNK; k2 <<= 1) {
#pragma unroll
            for (int j2 = k2 >> 1; j2 >= 1; j2 >>= 1) {
                if (j2 == 64) {
                    if (key[0] > key[1] || (key[0] == key[1] && slot[0] > slot[1])) {   // k2 == 128: ascending everywhere
                        const int32_t tkk = key[0], ts = slot[0];
                        key[0] = key[1]; slot[0] = slot[1]; key[1] = tkk; slot[1] = ts;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int pos = q * 64 + lane;
                        const int32_t pk = __shfl_xor(key[q], j2, 64);
                        const int32_t ps = __shfl_xor(slot[q], j2, 64);
                        const bool asc = (pos & k2) == 0;
                        const bool lower = (lane & j2) == 0;
                        // total order (key, slot): the walk order is the STABLE sort of the chunk by resident row
                        const bool p_less = pk < key[q] || (pk == key[q] && ps < slot[q]);
                        const bool take = (asc == lower) ? p_less : !p_less;
                        if (take) { key[q] = pk; slot[q] = ps; }
                    }
                }
            }
        }

        // ---- table roles for this chunk (wave-uniform) ------------------------------------------
        // The update is symmetric in the two sides: A = resident, B = streamed.
        float *const A_rows = res_is_ctx ? p.context : p.focus, *const A_gs = res_is_ctx ? p.gsc : p.gsf;
        float *const A_bias = res_is_ctx ? p.cbias : p.fbias,  *const A_gsb = res_is_ctx ? p.gscb : p.gsfb;
        float *const B_rows = res_is_ctx ? p.focus : p.context, *const B_gs = res_is_ctx ? p.gsf : p.gsc;
        float *const B_bias = res_is_ctx ? p.fbias : p.cbias,  *const B_gsb = res_is_ctx ? p.gsfb : p.gscb;
        float *const A_m2 = res_is_ctx ? p.m2c : p.m2f,  *const A_m2b = res_is_ctx ? p.m2cb : p.m2fb;     // MOM only
        float *const B_m2 = res_is_ctx ? p.m2f : p.m2c,  *const B_m2b = res_is_ctx ? p.m2fb : p.m2cb;

        // ---- sequential walk ---------------------------------------------------------------------
        // Software pipeline: while nonzero `pos` is computed, the streamed rows of pos+1 and (when the
        // resident row changes there) its resident rows are already in flight.  They are requested BEFORE
        // this nonzero's stores, so waiting for them never waits for a store to retire.
        int32_t cur_id = 0; bool cur_hot = false;
        bool cur_a32 = !EMB16, n_a32 = !EMB16;       // EMB16: resident parameter row lives in hub32 (fp32) for hub chunks
        VT a[NCH], ga[NCH], ha[NCH], a0[NCH], ga0[NCH];     // ha = second moment (MOM); ga0 = gradSq as read (AdaGrad)
        float ab = 0.0f, gab = 0.0f, hab = 0.0f;
        // the resident rows' resources are rebuilt from the row id where they are used (once per run): kept in SGPRs across the
        // walk they cost 24 registers, and what does not fit there spills into vector registers
        int32_t cur_slot = 0, n_slot = 0;            // EMB16 hub chunk: the row's slot in hub32; else the row id
        auto res_rows = [&](int32_t id, int32_t slot, bool a32) -> __amdgpu_buffer_rsrc_t {
            if constexpr (EMB16) {
                if (a32) return make_rsrc(p.hub32 + (int64_t)slot * D, (uint32_t)D * 4u);      // fp32 master rows are plain
                return make_rsrc(reinterpret_cast<uint16_t *>(A_rows) + (int64_t)id * p.ES, (uint32_t)p.D * 2u);
            } else return make_rsrc(A_rows + (int64_t)id * DS, row_bytes);
        };
        auto res_gs = [&](int32_t id) { return make_rsrc(A_gs + (int64_t)id * DS, row_bytes); };
        auto res_m2 = [&](int32_t id) { return make_rsrc(A_m2 + (int64_t)id * DS, MOM ? row_bytes : 0u); };

        auto close_run = [&]() {
            // How a run's result leaves the registers:
            //   STORE   the run owns its rows (whole focus rows of a row chunk; every run of the general order on an
            //           ordinary column): rows, accumulators and bias are stored.
            //   ATOMIC  the row is shared with other workers (hub column, piece of a long focus row) and takes float
            //           atomics: the run's DELTA is added, so no worker overwrites another worker's run.
            //   RMW     a shared FOCUS row that cannot take float atomics (bf16 storage; Adam / AMSGrad): the row is re-read,
            //           the delta added and the sum stored -- the race window shrinks from the run to these few instructions.
            // Adam / AMSGrad take steps of about the learning rate whatever the gradient's size, so concurrent HUB runs must
            // not ADD their moves (measured: the cost climbs again after a few epochs); they are merged last-writer-wins,
            // parameters and moments alike, which is exactly the Java race.  The run is still cut and re-read every
            // flush_lim nonzeros, so workers on one hub stay within that many updates of each other.
            const __amdgpu_buffer_rsrc_t rs_a = res_rows(cur_id, cur_slot, cur_a32), rs_ga = res_gs(cur_id), rs_ha = res_m2(cur_id);
            const bool store_all = !cur_hot || (MOM && res_is_ctx);
            const bool rmw_row = !store_all && !res_is_ctx && (MOM || (EMB16 && !cur_a32));
            if (store_all || (rmw_row && MOM)) {
                if constexpr (FAT) {                       // the bias goes out with its row
#pragma unroll
                    for (int q = 0; q < NCH; ++q)
                        if (q == bl_q && lane == bl_lane) {
                            if (store_all) comp<VW>(BIAS_IN_ACC ? ga[q] : a[q], BIAS_C) = ab;
                            comp<VW>(ga[q], 0) = gab;
                            if constexpr (MOM) comp<VW>(ha[q], 0) = hab;
                        }
                }
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    if (store_all) {
                        if (EMB16 && !cur_a32) emb_store(a[q], rs_a, q);
                        else buf_store<VW>(a[q], rs_a, (lane + q * 64) * VW * 4);
                    }
                    buf_store<VW>(ga[q], rs_ga, (lane + q * 64) * VW * 4);
                    if constexpr (MOM) buf_store<VW>(ha[q], rs_ha, (lane + q * 64) * VW * 4);
                }
            }
            if (rmw_row) {
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    VT cur = (EMB16 && !cur_a32) ? emb_load(rs_a, q) : buf_load<VW, AUX_SC1>(rs_a, (lane + q * 64) * VW * 4);
#pragma unroll
                    for (int t = 0; t < VW; ++t) comp<VW>(cur, t) += comp<VW>(a[q], t) - comp<VW>(a0[q], t);
                    if (EMB16 && !cur_a32) emb_store(cur, rs_a, q);
                    else buf_store<VW>(cur, rs_a, (lane + q * 64) * VW * 4);
                }
            }
            if (!store_all && !MOM) {
                // Publish the run's delta with float atomics.  Lane L holds elements [VW*L, VW*L+VW);
                // staged through this wave's LDS strip so that every atomic wave-instruction covers 64
                // CONSECUTIVE dwords (256 B = four 64-B memory-side requests instead of sixteen).
                float *tr_c = s_tr[threadIdx.x >> 6][0], *tr_g = s_tr[threadIdx.x >> 6][1];
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    VT dc, dg;
#pragma unroll
                    for (int t = 0; t < VW; ++t) {
                        comp<VW>(dc, t) = comp<VW>(a[q], t) - comp<VW>(a0[q], t);
                        comp<VW>(dg, t) = comp<VW>(ga[q], t) - comp<VW>(ga0[q], t);
                    }
                    if (!rmw_row) *reinterpret_cast<VT *>(tr_c + (lane + q * 64) * VW) = dc;
                    *reinterpret_cast<VT *>(tr_g + (lane + q * 64) * VW) = dg;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int k = 0; k < NCH * VW; ++k) {
                    const int e = lane + k * 64;          // element index; past the row the buffer check drops it
                    if (!rmw_row) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(tr_c[e], rs_a, e * 4, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(tr_g[e], rs_ga, e * 4, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
            }
            // The bias takes AdaGrad steps WITHOUT a learning rate (Adagrad.java:88-89): one run alone
            // already moves it most of the way, so concurrent runs must not add up.  The scalars are
            // merged last-writer-wins (what the Java race does), written through (sc1) like every table.
            if constexpr (FAT) {
                if (lane == 0 && !store_all) {             // the rows did not go out whole: the bias slot of each row is stored on its own
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ab), BIAS_IN_ACC ? rs_ga : rs_a, (D + BIAS_C) * 4, 0, AUX_SC1);
                    if (!MOM) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, gab), rs_ga, D * 4, 0, AUX_SC1);
                }
            } else if (lane == 0) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ab), make_rsrc(A_bias + cur_id, 4), 0, 0, AUX_SC1);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, gab), make_rsrc(A_gsb + cur_id, 4), 0, 0, AUX_SC1);
                if constexpr (MOM) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, hab), make_rsrc(A_m2b + cur_id, 4), 0, 0, AUX_SC1);
            }
        };

        // decoded next nonzero + its prefetched rows
        int32_t n_oth = 0, n_key = KEY_PAD; float n_w = 0.0f; double n_l = 0.0;
        VT aN[NCH], gaN[NCH], haN[NCH]; float abN = 0.0f, gabN = 0.0f, habN = 0.0f;
        auto decode = [&](int pos) {
            const int q = pos >> 6, ln = pos & 63;
            n_key = __builtin_amdgcn_readlane(q ? key[1] : key[0], ln);
            const int sl = __builtin_amdgcn_readlane(q ? slot[1] : slot[0], ln);
            const int sq = sl >> 6, sln = sl & 63;
            n_oth = __builtin_amdgcn_readlane(sq ? oth[1] : oth[0], sln);
            n_w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq ? ww[1] : ww[0]), sln));
            const long long lb = __builtin_bit_cast(long long, sq ? ll[1] : ll[0]);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(lb & 0xFFFFFFFFll), sln);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(lb >> 32), sln);
            n_l = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        };
        // Requests the streamed rows of nonzero n_oth into set SET (0: A, 1: B).  `live` false: the same instructions against
        // zero-sized resources -- nothing is read, zeros arrive.  Issued on EVERY step, so that the number of memory
        // instructions between a request and its use is the same on every path and the wait can be counted.
        auto request_streamed = [&](auto SET, const bool live) {
            const uint32_t rbytes = live ? row_bytes : 0u, ebytes = live ? (uint32_t)p.D * 2u : 0u, four = live ? 4u : 0u;
            __amdgpu_buffer_rsrc_t rb;
            if constexpr (EMB16) rb = make_rsrc(reinterpret_cast<uint16_t *>(B_rows) + (int64_t)n_oth * p.ES, ebytes);
            else rb = make_rsrc(B_rows + (int64_t)n_oth * DS, rbytes);
            const __amdgpu_buffer_rsrc_t rg = make_rsrc(B_gs + (int64_t)n_oth * DS, rbytes);
            const __amdgpu_buffer_rsrc_t rh = make_rsrc(B_m2 + (int64_t)n_oth * DS, MOM ? rbytes : 0u);
            float *const img = decltype(SET)::value ? setB : setA;
#pragma unroll
            for (int j = 0; j < (IMG_R / IMG) * N_Q; ++j)
                load_to_lds<(VW == 4 ? 16 : 4)>(rb, img + j * (IMG / N_Q), (lane + j * 64) * DMAW);
#pragma unroll
            for (int j = 0; j < NCH * N_Q; ++j) {
                load_to_lds<(VW == 4 ? 16 : 4)>(rg, img + IMG_G + j * (IMG / N_Q), (lane + j * 64) * DMAW);
                if constexpr (MOM) load_to_lds<(VW == 4 ? 16 : 4)>(rh, img + IMG_H + j * (IMG / N_Q), (lane + j * 64) * DMAW);
            }
            if constexpr (!FAT) {                  // lane 0 is the one lane inside these 4-byte resources
                load_to_lds<4>(make_rsrc(B_bias + n_oth, four), img + IMG_B, lane * 4);
                load_to_lds<4>(make_rsrc(B_gsb + n_oth, four), img + IMG_B + 64, lane * 4);
                if constexpr (MOM) load_to_lds<4>(make_rsrc(B_m2b + n_oth, four), img + IMG_B + 128, lane * 4);
            }
            asm volatile("" ::: "memory");          // issued here, not where the scheduler would like them
        };
        // waits until everything requested so far has arrived
        auto settle = [&]() { __builtin_amdgcn_s_waitcnt(wait_vmcnt(0)); asm volatile("" ::: "memory"); };
        auto request_resident = [&]() {
            const int32_t id = n_key < 0 ? ~n_key : n_key;
            n_slot = id;
            if constexpr (EMB16) {
                n_a32 = res_is_ctx && p.hot_enabled != 0;                  // hub chunk: fp32 master row
                if (n_a32) n_slot = rfl(p.hub_index[id]);
            }
            const __amdgpu_buffer_rsrc_t rsN_a = res_rows(id, n_slot, n_a32), rsN_ga = res_gs(id), rsN_ha = res_m2(id);
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                if constexpr (EMB16) {
                    // the row lives in one of two places (bf16 table / fp32 master row of a hub): BOTH loads are issued, the one that
                    // does not apply against a zero-sized resource (no traffic, zeros arrive) -- an `if` around either would put a
                    // path with no load at all in front of the hand-counted wait of the first step (tests/isa_waits.py)
                    const __amdgpu_buffer_rsrc_t r16 = make_rsrc(reinterpret_cast<uint16_t *>(A_rows) + (int64_t)id * p.ES, n_a32 ? 0u : (uint32_t)p.D * 2u);
                    const __amdgpu_buffer_rsrc_t r32 = make_rsrc(p.hub32 + (int64_t)n_slot * D, n_a32 ? (uint32_t)D * 4u : 0u);
                    VT v16 = emb_load(r16, q), v32 = buf_load<VW, AUX_SC1>(r32, (lane + q * 64) * VW * 4);
#pragma unroll
                    for (int t = 0; t < VW; ++t) comp<VW>(aN[q], t) = n_a32 ? comp<VW>(v32, t) : comp<VW>(v16, t);
                } else aN[q] = buf_load<VW, AUX_SC1>(rsN_a, (lane + q * 64) * VW * 4);
                gaN[q] = buf_load<VW, AUX_SC1>(rsN_ga, (lane + q * 64) * VW * 4);
                if constexpr (MOM) haN[q] = buf_load<VW, AUX_SC1>(rsN_ha, (lane + q * 64) * VW * 4);
            }
            if constexpr (!FAT) {
                abN  = buf_load_f32(make_rsrc(A_bias + id, 4), 0, true);
                gabN = buf_load_f32(make_rsrc(A_gsb + id, 4), 0, true);
                if constexpr (MOM) habN = buf_load_f32(make_rsrc(A_m2b + id, 4), 0, true);
            }
        };

        bool open_new = true;
        int run_len = 0;
        // the staged nonzeros have arrived before the walk starts: inside it only row traffic is pending
        asm volatile("" : "+v"(oth[0]), "+v"(oth[1]), "+v"(ww[0]), "+v"(ww[1]), "+v"(ll[0]), "+v"(ll[1]));
        bool again = false;
        // One nonzero.  CUR (0 / 1) names the set that holds its streamed rows; the other set is requested for the next one first
        // thing, before this step waits for anything.
        auto step = [&](const int pos, auto CUR) {
            constexpr std::integral_constant<int, 1 - decltype(CUR)::value> NXT{};
            const int32_t b_id = n_oth, skey = n_key;
            const float w = n_w; const double l = n_l;
            // the previous nonzero had the same streamed row: what was requested before its stores is stale.  Re-read behind
            // them (same wave, same address: in order) and wait right here, so that on the common path the set is known to be
            // older than those stores and the wait for it leaves them in flight.
            if (again) { request_streamed(CUR, true); settle(); again = false; }
            if (open_new) {                       // the scalars of a new run; its rows follow below
                cur_id = skey < 0 ? ~skey : skey;
                // shared = other workers may hold this row too: hub columns (context side), pieces of a long focus row
                cur_hot = p.blocked ? (res_is_ctx ? p.hot_enabled != 0 : cur_id == c_meta) : skey < 0;
                cur_slot = n_slot;
                cur_a32 = n_a32;
                run_len = 0;
            }
            const bool last = pos + 1 >= n_valid;
            bool next_new = false;
            if (!last) {
                decode(pos + 1);
                again = n_oth == b_id;
                // a long hub run is cut every flush_lim nonzeros: publish the delta, re-read what the other workers published
                next_new = n_key != skey || (cur_hot && run_len + 1 >= flush_lim);
            }
            request_streamed(NXT, !last);
            if (open_new) {
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    a[q] = aN[q]; ga[q] = gaN[q]; a0[q] = aN[q];
                    if constexpr (MOM) ha[q] = haN[q]; else ga0[q] = gaN[q];
                }
                ab = abN; gab = gabN; hab = habN;
                if constexpr (FAT) {
#pragma unroll
                    for (int q = 0; q < NCH; ++q)
                        if (q == bl_q) {
                            ab  = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, comp<VW>(BIAS_IN_ACC ? ga[q] : a[q], BIAS_C)), bl_lane));
                            gab = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, comp<VW>(ga[q], 0)), bl_lane));
                            if constexpr (MOM) hab = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, comp<VW>(ha[q], 0)), bl_lane));
                        }
                }
            }
            if (next_new) request_resident();     // after the copies above: it refills aN
            // this nonzero's streamed rows
            VT bl[NCH], gbl[NCH], hbl[NCH];
            float bb = 0.0f, gbb = 0.0f, hbb = 0.0f;
            // The wait is counted by hand (the compiler's own bookkeeping of loads into LDS loses them across the loop's
            // back edge).  Memory instructions complete in order, and behind this set's request at least N_AFTER more
            // have been issued on every path: the previous step's stores, one per table and register chunk (never
            // skipped: lane 0 of every chunk holds row elements), or before the first step the resident rows' loads, one per
            // table and chunk as well; then the request for the other set a few lines up.
            constexpr int N_TAB = NCH * (2 + (MOM ? 1 : 0)) + (FAT ? 0 : 2 + (MOM ? 1 : 0));
            constexpr int N_DMA = ((IMG_R / IMG) + NCH * (1 + (MOM ? 1 : 0))) * N_Q + (FAT ? 0 : 2 + (MOM ? 1 : 0));
            constexpr int N_AFTER = N_TAB + N_DMA;
            static_assert(N_AFTER < 64, "vmcnt is a 6-bit counter");
            __builtin_amdgcn_s_waitcnt(wait_vmcnt(N_AFTER));       // the builtin, not asm text: the compiler's bookkeeping sees it
            asm volatile("" ::: "memory");
            const float *const img = decltype(CUR)::value ? setB : setA;
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                if constexpr (EMB16) bl[q] = widen_bf16x4(*reinterpret_cast<const float2 *>(img + (lane + q * 64) * 2));
                else bl[q] = *reinterpret_cast<const VT *>(img + (lane + q * 64) * VW);
                gbl[q] = *reinterpret_cast<const VT *>(img + IMG_G + (lane + q * 64) * VW);
                if constexpr (MOM) hbl[q] = *reinterpret_cast<const VT *>(img + IMG_H + (lane + q * 64) * VW);
            }
            if constexpr (!FAT) { bb = img[IMG_B]; gbb = img[IMG_B + 64]; if constexpr (MOM) hbb = img[IMG_B + 128]; }
            VT (&b)[NCH] = bl, (&gb)[NCH] = gbl, (&hb)[NCH] = hbl;
            if constexpr (FAT) {
#pragma unroll
                for (int q = 0; q < NCH; ++q)
                    if (q == bl_q) {
                        bb  = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, comp<VW>(BIAS_IN_ACC ? gb[q] : b[q], BIAS_C)), bl_lane));
                        gbb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, comp<VW>(gb[q], 0)), bl_lane));
                        if constexpr (MOM) hbb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, comp<VW>(hb[q], 0)), bl_lane));
                    }
            }
            // dot product
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < NCH; ++q)
                if (inr[q]) {
#pragma unroll
                    for (int t = 0; t < VW; ++t) part = __builtin_fmaf(comp<VW>(a[q], t), comp<VW>(b[q], t), part);
                }
            part = wave_sum(part);
            const float ic = (float)((double)part + ((double)(ab + bb) - l));
            const float wc = w * ic;
            cost_acc += (0.5 * (double)wc) * (double)ic;
            const float wlr = wc * lr;
            const __amdgpu_buffer_rsrc_t rb = emb_rsrc(B_rows, b_id);
            const __amdgpu_buffer_rsrc_t rg = make_rsrc(B_gs + (int64_t)b_id * DS, row_bytes);
            const __amdgpu_buffer_rsrc_t rh = make_rsrc(B_m2 + (int64_t)b_id * DS, MOM ? row_bytes : 0u);
            // the streamed row's bias takes its step here and goes out with the row (FAT), see below for the arithmetic
            float nbb_fat = 0.0f, ngbb_fat = 0.0f, nhbb_fat = 0.0f;
            if constexpr (FAT) {
                if constexpr (!MOM) { nbb_fat = bb - wc * __frsqrt_rn(gbb); ngbb_fat = gbb + wc * wc; }
                else { float m = gbb, v = hbb; nbb_fat = bb - moment_step(wc, m, v); ngbb_fat = m; nhbb_fat = v; }
            }
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                // one store instruction per table covers the row's lanes AND the lane that holds the bias
                const bool is_bl = FAT && q == bl_q && lane == bl_lane;
                const bool tail = FAT && !inr[q] && (lane + q * 64) * VW < p.RW;     // the bias lane and the padding behind it: stored too, whole lines go out
                // The stores below are issued on EVERY path and by every lane: lanes past the row are dropped by the descriptor's
                // bounds check, not by the exec mask.  A store under `if (lane in range)` sits behind an s_cbranch_execz the
                // hand-counted wait further up cannot see past (tests/isa_waits.py counts the instructions on every path).
                {
                    VT ob{}, ogb{}, ohb{};
                    if (tail) {
                        if (is_bl) {
                            comp<VW>(ogb, 0) = ngbb_fat; comp<VW>(ohb, 0) = nhbb_fat;
                            comp<VW>(BIAS_IN_ACC ? ogb : ob, BIAS_C) = nbb_fat;     // (bf16: this lane's row store lies past the bf16 row and is dropped)
                        }
                    } else if (inr[q]) {
#pragma unroll
                        for (int t = 0; t < VW; ++t) {
                            const float av = comp<VW>(a[q], t), bv = comp<VW>(b[q], t);
                            const float grad_b = wc * av, grad_a = wc * bv;
                            if constexpr (!MOM) {
                                const float sa = comp<VW>(ga[q], t), sb = comp<VW>(gb[q], t);
                                comp<VW>(ob, t)    = __builtin_fmaf(-(wlr * av), __frsqrt_rn(sb), bv);
                                comp<VW>(ogb, t)   = __builtin_fmaf(grad_b, grad_b, sb);
                                comp<VW>(a[q], t)  = __builtin_fmaf(-(wlr * bv), __frsqrt_rn(sa), av);
                                comp<VW>(ga[q], t) = __builtin_fmaf(grad_a, grad_a, sa);
                            } else {
                                float m = comp<VW>(gb[q], t), v = comp<VW>(hb[q], t);
                                comp<VW>(ob, t) = bv - moment_step(grad_b, m, v);
                                comp<VW>(ogb, t) = m; comp<VW>(ohb, t) = v;
                                comp<VW>(a[q], t) = av - moment_step(grad_a, comp<VW>(ga[q], t), comp<VW>(ha[q], t));
                            }
                        }
                    }
                    emb_store(ob, rb, q);
                    buf_store<VW>(ogb, rg, (lane + q * 64) * VW * 4);
                    if constexpr (MOM) buf_store<VW>(ohb, rh, (lane + q * 64) * VW * 4);
                }
            }
            if constexpr (!MOM) {
                const float w2 = wc * wc;
                if constexpr (!FAT) {             // no learning rate on the biases (Adagrad.java:88-89); every lane issues the store, the
                                                  // 4-byte descriptor keeps lane 0's (no exec-mask branch in front of a counted instruction)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, bb - wc * __frsqrt_rn(gbb)), make_rsrc(B_bias + b_id, 4), lane * 4, 0, AUX_SC1);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, gbb + w2), make_rsrc(B_gsb + b_id, 4), lane * 4, 0, AUX_SC1);
                }
                ab = ab - wc * __frsqrt_rn(gab);
                gab = gab + w2;
            } else {                      // Adam.java:127-145: the biases take the same moment step with gradient wc
                const float nbb = bb - moment_step(wc, gbb, hbb);
                if constexpr (!FAT) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, nbb), make_rsrc(B_bias + b_id, 4), lane * 4, 0, AUX_SC1);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, gbb), make_rsrc(B_gsb + b_id, 4), lane * 4, 0, AUX_SC1);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, hbb), make_rsrc(B_m2b + b_id, 4), lane * 4, 0, AUX_SC1);
                }
                ab = ab - moment_step(wc, gab, hab);
            }
            ++run_len;
            if (last || next_new) close_run();
            open_new = next_new;
        };
        constexpr std::integral_constant<int, 0> SET_A{};
        constexpr std::integral_constant<int, 1> SET_B{};
        if (n_valid > 0) { decode(0); request_streamed(SET_A, true); request_resident(); }
        for (int pos = 0; pos < n_valid; pos += 2) {
            step(pos, SET_A);
            if (pos + 1 < n_valid) step(pos + 1, SET_B);
        }
    }
    if (lane == 0 && cost_acc != 0.0) atomicAdd(p.cost_out, cost_acc);
}

// Placement probe: what an epoch does to a record table -- whole records read and written back at random rows, sc1 like the trainer's
// accesses, one wavefront per stream of rows.  The values are written back unchanged (the table is not yet initialised and nobody
// else touches it), so the only result is the time.
__global__ __launch_bounds__(256) void k_probe_records(float *tab, int64_t rows, int64_t ds, int32_t rec_bytes, int32_t iters, uint32_t seed) {
    const int lane = threadIdx.x & 63;
    uint32_t x = (uint32_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 0x9E3779B1u + seed;       // the same in every lane of a wavefront
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const int64_t r = (int64_t)(((uint64_t)(x >> 4) * (uint64_t)rows) >> 28);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(tab + r * ds, (uint32_t)rec_bytes);
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = buf_load<4, AUX_SC1>(rs, (lane + q * 64) * 16);        // past the record: masked by the resource
#pragma unroll
        for (int q = 0; q < 4; ++q) buf_store<4>(v[q], rs, (lane + q * 64) * 16);
    }
}

using hogwild_fn = void (*)(GloveParams, int32_t);


template <int VW, int OPT, bool FAT>
hogwild_fn pick_nch(int nch) {
    switch (nch) {
        case 1: return k_adagrad_runs<VW, 1, OPT, false, FAT>;
        case 2: return k_adagrad_runs<VW, 2, OPT, false, FAT>;
        case 3: return k_adagrad_runs<VW, 3, OPT, false, FAT>;
        case 4: return k_adagrad_runs<VW, 4, OPT, false, FAT>;
        default: return nullptr;
    }
}
template <bool FAT>
hogwild_fn pick_bf16(int nch) {       // bf16 embeddings: AdaGrad, dim % 4 == 0
    switch (nch) {
        case 1: return k_adagrad_runs<4, 1, GE_OPT_ADAGRAD, true, FAT>;
        case 2: return k_adagrad_runs<4, 2, GE_OPT_ADAGRAD, true, FAT>;
        case 3: return k_adagrad_runs<4, 3, GE_OPT_ADAGRAD, true, FAT>;
        case 4: return k_adagrad_runs<4, 4, GE_OPT_ADAGRAD, true, FAT>;
        default: return nullptr;
    }
}
template <int OPT, bool FAT>
hogwild_fn pick_vw(int vw, int nch) { return vw == 4 ? pick_nch<4, OPT, FAT>(nch) : vw == 2 ? pick_nch<2, OPT, FAT>(nch) : pick_nch<1, OPT, FAT>(nch); }
template <int OPT>
hogwild_fn pick_fat(int vw, int nch, bool fat) { return fat ? pick_vw<OPT, true>(vw, nch) : pick_vw<OPT, false>(vw, nch); }
// rows carry their bias at element [D] (fat rows; bf16 rows: behind the accumulator row) when the lane that would hold it lies in
// the row's last 64-lane chunk
inline bool fat_rows_fit(int D) { const int vw = (D % 4 == 0) ? 4 : (D % 2 == 0) ? 2 : 1; return ((D / vw) % 64) != 0; }
// One wavefront spans a row: 64 lanes x VW floats x NCH chunks >= D.
hogwild_fn pick_hogwild(int D, int opt, bool emb16, int *vw_out, int *nch_out) {
    const int vw = (D % 4 == 0) ? 4 : (D % 2 == 0) ? 2 : 1;
    const bool fat = fat_rows_fit(D);
    const int nch = (D + 64 * vw - 1) / (64 * vw);           // the bias lane of a fat row lies inside the last chunk
    hogwild_fn fn = emb16 ? (fat ? pick_bf16<true>(nch) : pick_bf16<false>(nch))
                  : opt == GE_OPT_ADAGRAD ? pick_fat<GE_OPT_ADAGRAD>(vw, nch, fat)
                  : opt == GE_OPT_ADAM ? pick_fat<GE_OPT_ADAM>(vw, nch, fat) : pick_fat<GE_OPT_AMSGRAD>(vw, nch, fat);
    *vw_out = vw; *nch_out = nch;
    return fn;
}

}  // namespace

// =============================================================================================
// handle
// =============================================================================================
struct ge_glove {
    ge_glove_cfg cfg{};
    int32_t rows = 0;                 // focus rows owned = row_end - row_begin
    float *tab[GE_STATE_COUNT] = {};  // device tables (several may point into one allocation: see `owned`)
    int64_t tab_count[GE_STATE_COUNT] = {};
    std::vector<void *> owned;        // every device allocation of the handle
    int32_t *dI = nullptr, *dJ = nullptr, *dperm = nullptr;
    float *dX = nullptr;
    double *dL = nullptr;             // Hogwild: log term per nonzero
    float *dW = nullptr;              // Hogwild: weight per nonzero
    double *dcost = nullptr;          // Hogwild accumulator (+ the chunk queue word behind it)
    float *djob = nullptr;            // deterministic: per-job fp32 costs
    std::vector<int32_t> perm;        // host copy, GE_SHUFFLE_JAVA
    ge::JavaRandom rng{0};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.0f;
    int32_t last_launches = 0;
    int num_cus = 256;
    hogwild_fn hw_fn = nullptr;
    int hw_vw = 0, hw_nch = 0;
    int hw_blocks_per_cu = 4;
    int hw_blocks = 0;
    int hw_workers = 0;
    bool blocked = false;             // Hogwild + DEVICE shuffle: the layout of ge_layout.h
    ge::BlockedLayout lay;            // its device arrays (bA, bB, L, W, border, cstart, cmeta)
    int64_t n_chunks = 0, n_hchunks = 0;
    int flush_every = RUN_CHUNK;
    bool emb16 = false;               // focus/context stored as bf16 (tab[] pointers then address uint16 data)
    bool fat = false;                 // fp32 Hogwild: a row is D + 4 floats with its bias at [D]; tab[*BIAS] are null
    int32_t rw = 0;                   // row width of the fp32 row tables in floats (D, or D + 4 when fat)
    int32_t ds = 0;                   // row stride of the fp32 row tables in floats (rw, or a multiple when the tables interleave)
    int32_t es = 0;                   // bf16 rows: bf16 elements between consecutive embedding rows (dim, or 2 * ds inside records)
    int32_t placements = 0;           // allocations tried for the record tables (both sides)
    float place_best_ms = 0.0f, place_worst_ms = 0.0f;
    float *hub32 = nullptr;           // bf16 build: fp32 master rows of the hub columns
    int32_t *dhub_index = nullptr;
    unsigned long long seg_ticket[64] = {};   // first ticket of each segment of a segmented epoch (source of small async copies)
    hipEvent_t seg_ev[128] = {};              // a segmented epoch: events around every segment's launch (created on first use), so that
    int32_t seg_timed = 0;                    //   last_ms is the kernels' own time, the exchanges between them not included
    std::vector<int32_t> host_hub_index;
    int32_t n_hub = 0;
    std::vector<int32_t> host_key;    // general order: sort key per nonzero (what the kernel stages as `key`)
    int32_t hot_cols = 0;
    int64_t hot_nnz = 0, hot_threshold = 0;

    template <typename T> hipError_t alloc(T **out, size_t n) {
        hipError_t e = hipMalloc((void **)out, sizeof(T) * std::max<size_t>(n, 1));
        if (e == hipSuccess) owned.push_back((void *)*out);
        return e;
    }
};

namespace {

ge_status check_handle(ge_glove *h) {
    if (!h) return ge::fail(GE_ERR_ARG, "null ge_glove handle");
    hipError_t e = hipSetDevice(h->cfg.device);
    if (e != hipSuccess) return ge::fail(GE_ERR_HIP, "hipSetDevice(%d): %s", h->cfg.device, hipGetErrorString(e));
    return GE_OK;
}

void fill_params(const ge_glove *h, GloveParams &p, int32_t iteration) {
    p.focus = h->tab[GE_STATE_FOCUS];     p.context = h->tab[GE_STATE_CONTEXT];
    p.fbias = h->tab[GE_STATE_FBIAS];     p.cbias = h->tab[GE_STATE_CBIAS];
    p.gsf = h->tab[GE_STATE_GSQ_FOCUS];   p.gsc = h->tab[GE_STATE_GSQ_CONTEXT];
    p.gsfb = h->tab[GE_STATE_GSQ_FBIAS];  p.gscb = h->tab[GE_STATE_GSQ_CBIAS];
    p.m2f = h->tab[GE_STATE_M2_FOCUS];    p.m2c = h->tab[GE_STATE_M2_CONTEXT];
    p.m2fb = h->tab[GE_STATE_M2_FBIAS];   p.m2cb = h->tab[GE_STATE_M2_CBIAS];
    p.opt = h->cfg.opt;
    p.hub32 = h->hub32; p.hub_index = h->dhub_index;
    {   // Adam.java:84, evaluated in fp64 from the fp32 constants exactly as the Java expression does
        const float lrf = h->cfg.learning_rate, b1 = 0.9f, b2 = 0.999f;
        p.correction = (double)lrf * std::sqrt(1 - std::pow((double)b2, (double)(iteration + 1))) / (1 - std::pow((double)b1, (double)(iteration + 1)));
    }
    // focus-side tables hold rows [row_begin,row_end): rebase so that kernels index by global row id
    const int64_t off = h->cfg.row_begin;
    if (h->emb16) p.focus = reinterpret_cast<float *>(reinterpret_cast<uint16_t *>(p.focus) - off * h->es);
    else p.focus -= off * h->ds;
    p.gsf -= off * h->ds;
    if (!h->fat) { p.fbias -= off; p.gsfb -= off; }
    if (p.m2f) { p.m2f -= off * h->ds; if (!h->fat) p.m2fb -= off; }
    p.DS = h->ds; p.RW = h->rw; p.ES = h->es;
    p.I = h->dI; p.J = h->dJ; p.X = h->dX; p.perm = h->dperm;
    p.L = h->blocked ? h->lay.L : h->dL; p.W = h->blocked ? h->lay.W : h->dW;
    p.cost_out = h->dcost;
    p.queue = reinterpret_cast<unsigned long long *>(h->dcost + 1);
    p.xmax = h->cfg.xmax; p.N = h->cfg.nnz; p.D = h->cfg.dim;
    p.cost_kind = h->cfg.cost; p.lr = h->cfg.learning_rate;
    p.order_mode = h->cfg.shuffle == GE_SHUFFLE_JAVA ? ORDER_PERM
                 : h->cfg.shuffle == GE_SHUFFLE_DEVICE ? ORDER_BIJECTION : ORDER_IDENTITY;
    p.bA = h->lay.bA; p.bB = h->lay.bB; p.cstart = h->lay.cstart; p.cmeta = h->lay.cmeta; p.n_chunks = h->n_chunks; p.n_hchunks = h->n_hchunks;
    p.ticket_end = h->n_chunks;
    p.blocked = h->blocked ? 1 : 0; p.hot_enabled = h->cfg.hot_columns != GE_HOT_NONE; p.flush_every = h->flush_every;
    const int64_t domain = h->blocked ? h->n_chunks : h->cfg.nnz;      // what the keyed bijection permutes
    uint32_t bits = 0;
    while (bits < 31 && ((int64_t)1 << bits) < domain) ++bits;
    p.bij_mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
    p.bij_shift = bits > 1 ? bits / 2 : 1;
    // SplitMix64 of (seed, iteration) -> four round keys
    uint64_t z = (uint64_t)h->cfg.seed * 0x9E3779B97F4A7C15ULL + (uint64_t)(uint32_t)iteration * 0xD1B54A32D192ED03ULL + 0x632BE59BD9B4E019ULL;
    for (int q = 0; q < 4; ++q) {
        z += 0x9E3779B97F4A7C15ULL;
        uint64_t t = z;
        t = (t ^ (t >> 30)) * 0xBF58476D1CE4E5B9ULL;
        t = (t ^ (t >> 27)) * 0x94D049BB133111EBULL;
        t ^= t >> 31;
        p.bij_key[q] = (uint32_t)t;
    }
}

}  // namespace

extern "C" {

void ge_glove_cfg_default(ge_glove_cfg *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->cost = GE_COST_GLOVE;
    cfg->opt = GE_OPT_ADAGRAD;
    cfg->learning_rate = 0.05f;
    cfg->threads = 1;
    cfg->mode = GE_MODE_HOGWILD;
    cfg->shuffle = GE_SHUFFLE_DEVICE;
}

// Where the API's table `which` lives on the device.  Fat handles keep no bias tables: an fp32 row carries its bias at column
// [dim] (and the accumulator / moment row the bias's accumulator / moment there); a bf16 handle's rows cannot, so both scalars sit
// behind the ACCUMULATOR row: [gradSq (dim) | the bias accumulator | the bias | 2 x 0].
static const int FAT_HOME[GE_STATE_COUNT] = {GE_STATE_FOCUS, GE_STATE_CONTEXT, GE_STATE_FOCUS, GE_STATE_CONTEXT,
                                             GE_STATE_GSQ_FOCUS, GE_STATE_GSQ_CONTEXT, GE_STATE_GSQ_FOCUS, GE_STATE_GSQ_CONTEXT,
                                             GE_STATE_M2_FOCUS, GE_STATE_M2_CONTEXT, GE_STATE_M2_FOCUS, GE_STATE_M2_CONTEXT};
static bool is_bias_table(int which) { return FAT_HOME[which] != which; }
struct Home { float *base; int32_t col0, ncols; int64_t stride; };       // table[r][c] = base[r * stride + col0 + c], c < ncols
static Home home_of(const ge_glove *h, int which) {
    const int32_t D = h->cfg.dim;
    if (!is_bias_table(which)) return {h->tab[which], 0, D, h->ds};
    if (!h->fat) return {h->tab[which], 0, 1, 1};
    if (h->emb16) {
        const bool focus_side = which == GE_STATE_FBIAS || which == GE_STATE_GSQ_FBIAS;
        const bool the_bias = which == GE_STATE_FBIAS || which == GE_STATE_CBIAS;
        return {h->tab[focus_side ? GE_STATE_GSQ_FOCUS : GE_STATE_GSQ_CONTEXT], the_bias ? D + 1 : D, 1, h->ds};
    }
    return {h->tab[FAT_HOME[which]], D, 1, h->ds};
}

static ge_status ge_glove_create_impl(const ge_glove_cfg *cfg, const int32_t *I, const int32_t *J, const float *X,
                          ge_glove **out) {
    if (!out) return ge::fail(GE_ERR_ARG, "out is null");
    *out = nullptr;
    if (!cfg) return ge::fail(GE_ERR_ARG, "cfg is null");
    if (cfg->vocab_size <= 0) return ge::fail(GE_ERR_ARG, "vocab_size must be > 0 (got %d)", cfg->vocab_size);
    if (cfg->dim <= 0) return ge::fail(GE_ERR_ARG, "No dimension specified (dim=%d)", cfg->dim);   // Configuration.check
    if (cfg->nnz < 0 || cfg->nnz > 0x7FFFFFFFLL) return ge::fail(GE_ERR_ARG, "nnz out of range: %lld", (long long)cfg->nnz);
    if ((int64_t)cfg->vocab_size * cfg->dim > 0x7FFFFFFFLL)
        return ge::fail(GE_ERR_ARG, "vocab_size*dim exceeds Java int range (%d x %d)", cfg->vocab_size, cfg->dim);
    if (cfg->nnz > 0 && (!I || !J || !X)) return ge::fail(GE_ERR_ARG, "I/J/X must not be null");
    if (cfg->cost != GE_COST_GLOVE && cfg->cost != GE_COST_PGLOVE) return ge::fail(GE_ERR_ARG, "Invalid cost function %d", cfg->cost);
    if (cfg->opt < GE_OPT_ADAGRAD || cfg->opt > GE_OPT_AMSGRAD) return ge::fail(GE_ERR_ARG, "Invalid optimization method %d", cfg->opt);
    if (cfg->threads < 1) return ge::fail(GE_ERR_ARG, "threads must be >= 1");
    if (cfg->mode != GE_MODE_HOGWILD && cfg->mode != GE_MODE_DETERMINISTIC) return ge::fail(GE_ERR_ARG, "invalid mode %d", cfg->mode);
    if (cfg->shuffle < GE_SHUFFLE_JAVA || cfg->shuffle > GE_SHUFFLE_NONE) return ge::fail(GE_ERR_ARG, "invalid shuffle %d", cfg->shuffle);
    if (cfg->workers < -(1 << 20)) return ge::fail(GE_ERR_ARG, "workers out of range");
    if (cfg->emb_dtype != GE_DTYPE_F32 && cfg->emb_dtype != GE_DTYPE_BF16) return ge::fail(GE_ERR_ARG, "invalid emb_dtype %d", cfg->emb_dtype);
    const bool emb16 = cfg->emb_dtype == GE_DTYPE_BF16;
    if (emb16 && (cfg->mode != GE_MODE_HOGWILD || cfg->shuffle != GE_SHUFFLE_DEVICE || cfg->opt != GE_OPT_ADAGRAD || cfg->dim % 4 != 0))
        return ge::fail(GE_ERR_ARG, "bf16 embeddings need mode=hogwild, shuffle=device, opt=adagrad and dim %% 4 == 0 (the reference path is fp32)");
    if (cfg->hot_columns < GE_HOT_AUTO || cfg->hot_columns > GE_HOT_ALL) return ge::fail(GE_ERR_ARG, "invalid hot_columns %d", cfg->hot_columns);
    if (cfg->hot_theta < 0 || cfg->stale_budget < 0 || cfg->flush_every < 0 || cfg->blocks_per_cu < 0 || (cfg->layout_flags & ~31) != 0)
        return ge::fail(GE_ERR_ARG, "invalid tuning fields (hot_theta %g, stale_budget %g, flush_every %d, blocks_per_cu %d, layout_flags %d)",
                        (double)cfg->hot_theta, (double)cfg->stale_budget, cfg->flush_every, cfg->blocks_per_cu, cfg->layout_flags);
    int32_t rb = cfg->row_begin, re = cfg->row_end;
    if (rb == 0 && re == 0) re = cfg->vocab_size;
    if (rb < 0 || re > cfg->vocab_size || rb >= re) return ge::fail(GE_ERR_ARG, "invalid row range [%d,%d)", rb, re);
    const bool will_block = cfg->mode == GE_MODE_HOGWILD && cfg->shuffle == GE_SHUFFLE_DEVICE;   // blocked layout, built (and range-checked) on the device
    if (!will_block)
        for (int64_t k = 0; k < cfg->nnz; ++k) {
            if (I[k] < rb || I[k] >= re) return ge::fail(GE_ERR_ARG, "I[%lld]=%d outside owned rows [%d,%d)", (long long)k, I[k], rb, re);
            if (J[k] < 0 || J[k] >= cfg->vocab_size) return ge::fail(GE_ERR_ARG, "J[%lld]=%d outside [0,%d)", (long long)k, J[k], cfg->vocab_size);
        }
    ge_status st = ge::select_device(cfg->device);
    if (st != GE_OK) return st;

    ge_glove *h = new (std::nothrow) ge_glove();
    if (!h) return ge::fail(GE_ERR_OOM, "host allocation failed");
    h->cfg = *cfg;
    h->emb16 = emb16;
    h->fat = cfg->mode == GE_MODE_HOGWILD && fat_rows_fit(cfg->dim);
    h->cfg.row_begin = rb; h->cfg.row_end = re;
    h->rows = re - rb;
    h->stream = (hipStream_t)cfg->stream;
    const int32_t V = cfg->vocab_size, D = cfg->dim;
    const int64_t N = cfg->nnz;
    const bool moments = cfg->opt != GE_OPT_ADAGRAD;      // Adam / AMSGrad keep M2* next to M1*
    const bool interleave = cfg->mode == GE_MODE_HOGWILD && (cfg->layout_flags & GE_LAYOUT_SEPARATE_TABLES) == 0;
    const bool packed = (cfg->layout_flags & GE_LAYOUT_PACKED_RECORDS) != 0;
    h->rw = h->fat ? D + 4 : D;
    // An fp32 fat row as wide as the whole 64-byte lines that hold it (D = 200: 208 floats): every row store then writes whole
    // lines and a row shares no line with its accumulator row.  More bytes, less time: D = 100 (112 instead of 104 floats) 29.8 /
    // 30.4 -> 27.5 / 27.5 ms, D = 200 48.9 / 53.8 -> 47.8 / 47.8 ms (tools/r02/rw_probe.sh).  Not for bf16 rows (no
    // gain measured) and not where the padding would exceed 10 %.
    if (h->fat && !emb16 && !packed && D % 4 == 0) {
        const int32_t lines = (D + 1 + 15) / 16 * 16;
        if ((int64_t)lines * 10 <= (int64_t)(D + 4) * 11) h->rw = lines;
    }
    h->ds = h->rw * (interleave ? (moments ? 3 : 2) : 1);
    // bf16 rows in records: [bf16 row, padded to 16 bytes | fp32 accumulator row]; e16 = bf16 elements of the padded row
    const int32_t e16 = (D + 7) / 8 * 8;
    if (emb16) h->ds = interleave ? e16 / 2 + h->rw : h->rw;                                             // rw: the accumulator row (fat: + its two scalars)
    // A record starts on a 64-byte boundary: a wave's 16-byte-per-lane row access then covers whole 64-byte requests.  Records of
    // 1 632 bytes (D = 200) start on odd 32-byte sectors half of the time; rounding them up to 1 664 took the epoch from 60.9 to
    // 49.2 ms on one box and from 60.8 to 54.6 on another (tools/r02/align_probe.sh, kernel_ab.sh).  Rounding further
    // (128 bytes and more) gains nothing at D = 200 and loses 5 % on the 1 216-byte records of bf16 rows.  The padding is never touched.
    if (interleave && !packed) h->ds = (h->ds + 15) / 16 * 16;
    if (const char *pad = std::getenv("GE_RECORD_PAD_LINES")) if (interleave && !packed) h->ds += 16 * std::max(0, std::atoi(pad));   // experiment: record stride + n x 64 B
    if (emb16) h->es = interleave ? 2 * h->ds : D;

    // GE_GLOVE_TIMING=1: where ge_glove_create's time goes, to stderr (host clock; the stream is drained at each lap)
    struct CreateClock {
        bool on = std::getenv("GE_GLOVE_TIMING") != nullptr;
        hipStream_t stream = nullptr;
        std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
        void lap(const char *what) {
            if (!on) return;
            (void)hipStreamSynchronize(stream);
            const auto n = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[ge_glove_create] %-34s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
            t = n;
        }
    } clk;
    clk.stream = h->stream;
    // every failure below frees what the handle owns so far (ge_glove_destroy walks h->owned)
#define GE_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { ge_status _s = ge::fail(_e == hipErrorOutOfMemory ? GE_ERR_OOM : GE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); ge_glove_destroy(h); return _s; } } while (0)

    hipDeviceProp_t prop;
    GE_TRY(hipGetDeviceProperties(&prop, cfg->device));
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    GE_TRY(hipEventCreate(&h->ev0));
    GE_TRY(hipEventCreate(&h->ev1));

    // ---- tables, allocated once, in their final layout ------------------------------------------------------------
    // fp32 row tables: `rw` floats per row (fat rows carry the bias at [D]); unless GE_LAYOUT_SEPARATE_TABLES a side is ONE
    // allocation of records [row | accumulator row (| second moment row)], stride ds = 2 or 3 rw.  bf16 rows: records
    // [bf16 row padded to 16 bytes | fp32 accumulator row, fat: gradSq (D) | its bias accumulator | the bias | 2 x 0].
    const int64_t counts[GE_STATE_COUNT] = {
        (int64_t)h->rows * D, (int64_t)V * D, h->rows, V, (int64_t)h->rows * D, (int64_t)V * D, h->rows, V,
        moments ? (int64_t)h->rows * D : 0, moments ? (int64_t)V * D : 0, moments ? h->rows : 0, moments ? V : 0};
    for (int t = 0; t < GE_STATE_COUNT; ++t) h->tab_count[t] = counts[t];
    {
        static const int ROWT[2][3] = {{GE_STATE_FOCUS, GE_STATE_GSQ_FOCUS, GE_STATE_M2_FOCUS}, {GE_STATE_CONTEXT, GE_STATE_GSQ_CONTEXT, GE_STATE_M2_CONTEXT}};
        static const int BIAST[2][3] = {{GE_STATE_FBIAS, GE_STATE_GSQ_FBIAS, GE_STATE_M2_FBIAS}, {GE_STATE_CBIAS, GE_STATE_GSQ_CBIAS, GE_STATE_M2_CBIAS}};
        const int n_aux = moments ? 3 : 2;
        for (int side = 0; side < 2; ++side) {
            const size_t nr = side == 0 ? (size_t)h->rows : (size_t)V;
            if (interleave) {
                // Where the driver places a table decides which of two epoch times the handle gets (same process, same
                // virtual address, 48 or 54 ms at the bench size).  So a large table is allocated up to six times, each candidate
                // while the earlier ones are still held (else the driver hands the same pages back), a millisecond of what an epoch
                // does to it is timed on each (0.91 - 0.97 ms on the good placements, 1.04 - 1.11 ms on the others), the fastest is kept.
                float *blk = nullptr;
                const size_t bytes = nr * (size_t)h->ds * sizeof(float);
                const char *alloc_env = std::getenv("GE_TABLE_ALLOC");
                const bool contiguous = alloc_env && std::strcmp(alloc_env, "contiguous") == 0;
                // Bounds: the candidates of one table together stay under 24 GB (allocating and freeing an 8 GB table costs a quarter
                // of a second) AND under half of what the device has free right now (another process may share the GPU; a candidate
                // that cannot be allocated ends the search with what there is); a table too large for a second candidate gets one.
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
                const size_t budget = std::min<size_t>((size_t)24 << 30, free_b / 2);
                const int tries = ((cfg->layout_flags & GE_LAYOUT_FIRST_PLACEMENT) || bytes < ((size_t)64 << 20)) ? 1
                                : (int)std::max<size_t>(1, std::min<size_t>(6, budget / bytes));
                float *cand[6] = {}; float cand_ms[6] = {};
                int n_cand = 0, best = 0;
                for (int t = 0; t < tries; ++t) {
                    hipError_t me = hipErrorUnknown;
                    if (contiguous) {                                       // experiment (GE_TABLE_ALLOC=contiguous): physically contiguous VRAM
                        me = hipExtMallocWithFlags((void **)&cand[t], bytes, hipDeviceMallocContiguous);
                        if (me != hipSuccess) { (void)hipGetLastError(); cand[t] = nullptr; }
                    }
                    if (me != hipSuccess) me = hipMalloc((void **)&cand[t], bytes);
                    if (me != hipSuccess) { (void)hipGetLastError(); cand[t] = nullptr; break; }     // no room for another candidate: keep what there is
                    ++n_cand;
                    h->owned.push_back((void *)cand[t]);               // the handle owns every candidate until the losers are freed below
                    if (tries > 1) {
                        const int32_t rec_bytes = (int32_t)std::min<int64_t>((int64_t)h->ds * 4, 4096);
                        float ms_min = 1e30f;
                        for (int rep = 0; rep < 3; ++rep) {
                            GE_TRY(hipEventRecord(h->ev0, h->stream));
                            hipLaunchKernelGGL(k_probe_records, dim3((unsigned)h->num_cus * 5), dim3(256), 0, h->stream, cand[t], (int64_t)nr, (int64_t)h->ds, rec_bytes, 384, 0x5EEDu + rep);
                            GE_TRY(hipEventRecord(h->ev1, h->stream));
                            GE_TRY(hipEventSynchronize(h->ev1));
                            float ms = 0.0f; GE_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
                            ms_min = std::min(ms_min, ms);
                        }
                        cand_ms[t] = ms_min;
                        if (cand_ms[t] < cand_ms[best]) best = t;
                        float slowest = cand_ms[0];
                        for (int u = 1; u <= t; ++u) slowest = std::max(slowest, cand_ms[u]);
                        if (cand_ms[best] < 0.88f * slowest) break;        // both kinds seen and a fast one in hand (they lie 15 - 20 % apart)
                        if (t >= 2 && cand_ms[best] > 0.98f * slowest) break;  // three candidates within 2 %: a box that has one kind only
                    }
                }
                if (n_cand == 0) { ge_status _s = ge::fail(GE_ERR_OOM, "hipMalloc of a %zu-byte record table failed", bytes); ge_glove_destroy(h); return _s; }
                float worst = cand_ms[best];
                for (int t = 0; t < n_cand; ++t) {
                    worst = std::max(worst, cand_ms[t]);
                    if (t == best) continue;
                    h->owned.erase(std::find(h->owned.begin(), h->owned.end(), (void *)cand[t]));
                    (void)hipFree(cand[t]);
                }
                blk = cand[best];
                h->placements += n_cand; h->place_best_ms += cand_ms[best]; h->place_worst_ms += worst;
                if (emb16) { h->tab[ROWT[side][0]] = blk; h->tab[ROWT[side][1]] = blk + e16 / 2; }     // the bf16 row leads its record
                else for (int a = 0; a < n_aux; ++a) h->tab[ROWT[side][a]] = blk + (size_t)a * h->rw;
            } else {
                for (int a = 0; a < n_aux; ++a) {
                    if (a == 0 && emb16) { uint16_t *t16 = nullptr; GE_TRY(h->alloc(&t16, nr * (size_t)D)); h->tab[ROWT[side][0]] = reinterpret_cast<float *>(t16); }
                    else GE_TRY(h->alloc(&h->tab[ROWT[side][a]], nr * (size_t)h->ds));
                }
            }
            if (!h->fat) for (int a = 0; a < n_aux; ++a) GE_TRY(h->alloc(&h->tab[BIAST[side][a]], nr));
        }
    }
    clk.lap("tables (placement search)");
    const size_t nn = (size_t)std::max<int64_t>(N, 1);
    GE_TRY(h->alloc(&h->dcost, 2));
    GE_TRY(h->alloc(&h->djob, (size_t)cfg->threads));

    if (cfg->mode == GE_MODE_HOGWILD) {
        h->hw_fn = pick_hogwild(D, cfg->opt, emb16, &h->hw_vw, &h->hw_nch);
        if (!h->hw_fn) { ge_glove_destroy(h); return ge::fail(GE_ERR_ARG, "dim %d not supported by the Hogwild kernel (max 1024 for dim%%4==0, 512 for other even dims, 256 for odd dims)", D); }
        // One wavefront = one sequential worker.  Never more workers than N/2048: a small matrix must
        // not degenerate into one giant stale batch (the JVM has at most #cores updates in flight).
        const int groups_per_block = 4;
        const int64_t chunks = (N + RUN_CHUNK - 1) / RUN_CHUNK;
        int occ = 0;
        h->hw_blocks_per_cu = cfg->blocks_per_cu > 0 ? cfg->blocks_per_cu : 4;
        if (cfg->blocks_per_cu == 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(h->hw_fn), 256, 0) == hipSuccess && occ > 0)
            h->hw_blocks_per_cu = occ;                   // every worker resident: one wave of blocks
        int64_t blocks = std::min<int64_t>((chunks + 3) / 4, (int64_t)h->num_cus * h->hw_blocks_per_cu);
        blocks = std::min<int64_t>(blocks, std::max<int64_t>(1, N / 2048 / groups_per_block));
        h->hw_blocks = (int)std::max<int64_t>(blocks, 1);
        h->hw_workers = h->hw_blocks * 4;
        if (cfg->workers > 0) {                       // explicit worker count (tests, reproducibility)
            h->hw_workers = (int)std::min<int64_t>(cfg->workers, (int64_t)h->num_cus * 32);
            h->hw_blocks = (h->hw_workers + 3) / 4;
        } else if (cfg->workers < 0) {                // fill the device but leave -workers wavefront slots to kernels
            h->hw_blocks = (int)std::max<int64_t>(1, (int64_t)h->hw_blocks - (-(int64_t)cfg->workers + 3) / 4);   // running beside (collectives)
            h->hw_workers = h->hw_blocks * 4;
        }
        // (0.05 -- five times the hub columns -- takes a fifth off the Hogwild lag, C2 epoch 32: 1.068 -> 1.055 of the sequential oracle's
        // cost, and is free on the faster kind of placement (48.0 -> 48.3 ms) but costs 4 - 7 % on the slower kind (54.9 -> 57 - 58.7 ms,
        // tools/r03/theta_modes.py): the default stays 0.25)
        const double theta = cfg->hot_theta > 0 ? (double)cfg->hot_theta : 0.25;
        const double stale_budget = cfg->stale_budget > 0 ? (double)cfg->stale_budget : 2000.0;
        h->blocked = cfg->shuffle == GE_SHUFFLE_DEVICE;
        if (h->blocked) {
            ge::LayoutRequest rq{};
            rq.V = V; rq.row_begin = rb; rq.row_end = re; rq.N = N; rq.cost = cfg->cost; rq.xmax = cfg->xmax;
            rq.hot_columns = cfg->hot_columns; rq.hot_theta = theta; rq.stale_budget = stale_budget; rq.flush_every = cfg->flush_every;
            rq.workers = h->hw_workers;
            rq.shared_rows = (cfg->layout_flags & GE_LAYOUT_PLAIN_LONG_ROWS) ? 0 : 1;
            rq.pack_rows = (cfg->layout_flags & GE_LAYOUT_FIXED_CUTS) ? 0 : 1;
            rq.want_hub_index = emb16;
            st = ge::build_blocked_layout(rq, I, J, X, h->stream, &h->lay);
            if (st != GE_OK) { ge_glove_destroy(h); return st; }
            h->n_chunks = h->lay.n_chunks; h->n_hchunks = h->lay.n_hchunks;
            h->hot_cols = h->lay.hot_cols; h->hot_nnz = h->lay.hot_nnz; h->hot_threshold = h->lay.hot_threshold;
            h->flush_every = h->lay.flush_min;
            if (emb16) { h->host_hub_index.swap(h->lay.hub_index); h->n_hub = h->lay.n_hub; }
        } else {
            // general order (Java permutation / matrix order): the resident side is always the context row, hub columns are keyed ~j
            std::vector<int32_t> cnt((size_t)V, 0);
            std::vector<uint8_t> hotcol((size_t)V, 0);
            if (N > 0 && cfg->hot_columns != GE_HOT_NONE) {
                const int64_t thr = cfg->hot_columns == GE_HOT_ALL ? 0
                                  : std::max<int64_t>(2, (int64_t)std::ceil(theta * (double)N / (double)h->hw_workers));
                for (int64_t k = 0; k < N; ++k) ++cnt[(size_t)J[k]];
                for (int32_t v = 0; v < V; ++v)
                    if (cnt[(size_t)v] >= thr && cnt[(size_t)v] > 0) { hotcol[(size_t)v] = 1; ++h->hot_cols; h->hot_nnz += cnt[(size_t)v]; }
                h->hot_threshold = thr;
            }
            h->flush_every = cfg->flush_every > 0 ? std::min<int32_t>(cfg->flush_every, RUN_CHUNK) : RUN_CHUNK;
            if (cfg->flush_every == 0)
                for (int32_t v = 0; v < V; ++v) if (hotcol[(size_t)v]) {
                    const double K = std::max(1.0, (double)cnt[(size_t)v] * (double)h->hw_workers / (double)std::max<int64_t>(N, 1));
                    h->flush_every = std::min<int>(h->flush_every, (int)std::min<double>(RUN_CHUNK, std::max<double>(4.0, std::floor(stale_budget / K))));
                }
            h->host_key.assign(J, J + N);
            for (int64_t k = 0; k < N; ++k) if (hotcol[(size_t)J[k]]) h->host_key[(size_t)k] = ~J[k];
            h->n_chunks = chunks; h->n_hchunks = 0;
        }
    } else if ((size_t)D * sizeof(float) > 64 * 1024) {
        ge_glove_destroy(h);
        return ge::fail(GE_ERR_ARG, "dim %d too large for deterministic mode", D);
    }
    clk.lap("epoch layout");
    if (!h->blocked) {
        GE_TRY(h->alloc(&h->dI, nn)); GE_TRY(h->alloc(&h->dJ, nn)); GE_TRY(h->alloc(&h->dX, nn));
        if (N > 0) {
            GE_TRY(hipMemcpyAsync(h->dI, I, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, h->stream));
            GE_TRY(hipMemcpyAsync(h->dJ, h->host_key.empty() ? J : h->host_key.data(), sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, h->stream));
            GE_TRY(hipMemcpyAsync(h->dX, X, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, h->stream));
            if (cfg->mode == GE_MODE_HOGWILD) {
                GE_TRY(h->alloc(&h->dL, nn)); GE_TRY(h->alloc(&h->dW, nn));
                const int blocks = (int)std::min<int64_t>((N + 255) / 256, 8192);
                hipLaunchKernelGGL(k_cost_terms, dim3(blocks), dim3(256), 0, h->stream, h->dX, N, cfg->cost, cfg->xmax, h->dL, h->dW);
            }
            GE_TRY(hipStreamSynchronize(h->stream));
        }
    }
    if (cfg->shuffle == GE_SHUFFLE_JAVA) {
        GE_TRY(h->alloc(&h->dperm, nn));
        h->perm.resize((size_t)N);
        for (int64_t k = 0; k < N; ++k) h->perm[(size_t)k] = (int32_t)k;     // Permutation ctor
    }

    // --- parameter init in the reference's draw order, straight into the final layout; the context side covers all V
    //     rows, the focus side only the owned rows (same values a single-GPU run would hold there).
    const uint64_t s0 = ge::JavaRandom::scramble(cfg->seed);
    if (emb16) {
        GE_TRY(h->alloc(&h->dhub_index, (size_t)V));
        GE_TRY(hipMemcpyAsync(h->dhub_index, h->host_hub_index.data(), sizeof(int32_t) * (size_t)V, hipMemcpyHostToDevice, h->stream));
        GE_TRY(h->alloc(&h->hub32, (size_t)std::max<int64_t>((int64_t)h->n_hub * D, 1)));
    }
    {
        // Adagrad ctor: gradSq = 1 (Adagrad.java:27-33); Adam / AMSGrad ctors: every moment = 0 (new float[]).  Before the parameter
        // init: a bf16 handle's bias lives in the accumulator row and must survive the fill.
        const float v0 = moments ? 0.0f : 1.0f;
        for (int t : {GE_STATE_GSQ_FOCUS, GE_STATE_GSQ_CONTEXT, GE_STATE_M2_FOCUS, GE_STATE_M2_CONTEXT}) {
            if (!h->tab[t]) continue;
            const int64_t nr = h->tab_count[t] / D;
            const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((nr * h->rw + 255) / 256, 8192));
            hipLaunchKernelGGL(k_fill_rows, dim3(blocks), dim3(256), 0, h->stream, h->tab[t], nr, (int64_t)h->ds, h->fat ? D + 1 : D, h->rw, v0);
        }
        for (int t : {GE_STATE_GSQ_FBIAS, GE_STATE_GSQ_CBIAS, GE_STATE_M2_FBIAS, GE_STATE_M2_CBIAS}) {
            if (!h->tab[t]) continue;
            const int64_t n = h->tab_count[t];
            hipLaunchKernelGGL(k_fill, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096))), dim3(256), 0, h->stream, h->tab[t], n, v0);
        }
        const int64_t stride = emb16 ? h->es : h->ds;
        const int32_t bias_col = (h->fat && !emb16) ? D : -1;        // fp32 fat rows: the bias is a column of the row being written
        const Home fb = home_of(h, GE_STATE_FBIAS), cb = home_of(h, GE_STATE_CBIAS);
        void *foc = h->tab[GE_STATE_FOCUS], *ctx = h->tab[GE_STATE_CONTEXT];
        auto launch = [&](void *f, void *c, int32_t row0, int32_t nrows) {
            const dim3 g((unsigned)((nrows + 127) / 128)), b(128);
            if (emb16) hipLaunchKernelGGL(k_init_java<true>, g, b, 0, h->stream, f, c, fb.base + fb.col0, cb.base + cb.col0, (int64_t)fb.stride, rb,
                                          row0, nrows, D, stride, bias_col, h->rw, s0, (const int32_t *)h->dhub_index, h->hub32);
            else hipLaunchKernelGGL(k_init_java<false>, g, b, 0, h->stream, f, c, fb.base ? fb.base + fb.col0 : nullptr, cb.base ? cb.base + cb.col0 : nullptr, (int64_t)fb.stride, rb,
                                    row0, nrows, D, stride, bias_col, h->rw, s0, (const int32_t *)nullptr, (float *)nullptr);
        };
        if (h->rows == V) launch(foc, ctx, 0, V);
        else { launch(nullptr, ctx, 0, V); launch(foc, nullptr, rb, h->rows); }
    }
    GE_TRY(hipGetLastError());
    GE_TRY(hipStreamSynchronize(h->stream));
    h->rng.s = ge::JavaRandom::jump(s0, (uint64_t)V * (uint64_t)(2 + 2 * D));

    clk.lap("init and the rest");
#undef GE_TRY
    *out = h;
    return GE_OK;
}

static ge_status ge_glove_epoch_impl(ge_glove *h, int32_t iteration, double *cost_sum) {
    ge_status st = check_handle(h);
    if (st != GE_OK) return st;
    const int64_t N = h->cfg.nnz;
    if (h->cfg.shuffle == GE_SHUFFLE_JAVA && N > 0) {
        // ExtendedRandom.shuffle(int[]), cumulative on the same array (J/util/rnd/ExtendedRandom.java:398-407)
        int32_t *a = h->perm.data();
        const int32_t n = (int32_t)N;
        for (int32_t i = 0; i < n; ++i) {
            const int32_t r = i + h->rng.next_int(n - i);
            const int32_t t = a[i]; a[i] = a[r]; a[r] = t;
        }
        GE_HIP(hipMemcpyAsync(h->dperm, a, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, h->stream));
    }
    GloveParams p;
    fill_params(h, p, iteration);
    double total = 0.0;
    h->last_launches = 0;
    if (h->cfg.mode == GE_MODE_DETERMINISTIC) {
        const int T = h->cfg.threads;
        const int64_t per = N / T;
        GE_HIP(hipMemsetAsync(h->djob, 0, sizeof(float) * (size_t)T, h->stream));
        GE_HIP(hipEventRecord(h->ev0, h->stream));
        for (int t = 0; t < T; ++t) {
            const int64_t off = per * t;                                  // Adagrad.java:47
            const int64_t lines = (t == T - 1) ? per + N % T : per;       // Optimizer.java:59-63
            if (lines <= 0) continue;
            hipLaunchKernelGGL(k_adagrad_exact, dim3(1), dim3(64), sizeof(float) * (size_t)h->cfg.dim, h->stream,
                               p, off, off + lines, h->djob + t);
            ++h->last_launches;
        }
        GE_HIP(hipEventRecord(h->ev1, h->stream));
        GE_HIP(hipGetLastError());
        std::vector<float> jc((size_t)T);
        GE_HIP(hipMemcpyAsync(jc.data(), h->djob, sizeof(float) * (size_t)T, hipMemcpyDeviceToHost, h->stream));
        GE_HIP(hipStreamSynchronize(h->stream));
        for (int t = 0; t < T; ++t) total += (double)jc[(size_t)t];       // localCost += job result (Optimizer.java:89)
    } else {
        GE_HIP(hipMemsetAsync(h->dcost, 0, 2 * sizeof(double), h->stream));
        GE_HIP(hipEventRecord(h->ev0, h->stream));
        if (N > 0) {
            hipLaunchKernelGGL(h->hw_fn, dim3(h->hw_blocks), dim3(256), 0, h->stream, p, (int32_t)h->hw_workers);
            ++h->last_launches;
        }
        GE_HIP(hipEventRecord(h->ev1, h->stream));
        GE_HIP(hipGetLastError());
        GE_HIP(hipMemcpyAsync(&total, h->dcost, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        GE_HIP(hipStreamSynchronize(h->stream));
    }
    GE_HIP(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    if (cost_sum) *cost_sum = total;
    return GE_OK;
}

// bf16 build: an fp32 device copy of FOCUS or CONTEXT as the caller sees it (hub rows from their fp32 masters)
// fat build: any table as the API shows it (rows [n x D] or a bias vector [n]) gathered out of the fat rows
// does the API view of table `which` differ from how the handle stores it?  (fat rows: every fp32 table; interleaved
// records: the row tables; bf16: the two embedding tables)
static bool stored_strided(const ge_glove *h, int which) {
    if (h->emb16 && (which == GE_STATE_FOCUS || which == GE_STATE_CONTEXT)) return false;     // bf16: converted, not gathered
    return is_bias_table(which) ? h->fat : h->ds != h->cfg.dim;
}
static ge_status materialize_f32(ge_glove *h, int which, float **out) {
    const int64_t n = h->tab_count[which];
    float *d = nullptr;
    GE_HIP(hipMalloc((void **)&d, sizeof(float) * (size_t)std::max<int64_t>(n, 1)));
    if (stored_strided(h, which)) {
        const Home hm = home_of(h, which);
        hipLaunchKernelGGL(k_fat_gather, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 16384))), dim3(256), 0, h->stream,
                           hm.base, n / hm.ncols, (int32_t)hm.stride, hm.col0, hm.ncols, d);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { (void)hipFree(d); return ge::fail(GE_ERR_HIP, "row gather failed: %s", hipGetErrorString(e)); }
        *out = d;
        return GE_OK;
    }
    hipLaunchKernelGGL(k_bf16_to_f32, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, h->stream,
                       reinterpret_cast<const uint16_t *>(h->tab[which]), d, n, h->cfg.dim, (int64_t)h->es);
    if (which == GE_STATE_CONTEXT && h->n_hub > 0)
        hipLaunchKernelGGL(k_hub_rows, dim3((unsigned)h->cfg.vocab_size), dim3(64), 0, h->stream, d, h->hub32, h->dhub_index, h->cfg.vocab_size, h->cfg.dim, 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(d); return ge::fail(GE_ERR_HIP, "bf16 -> fp32 conversion failed: %s", hipGetErrorString(e)); }
    *out = d;
    return GE_OK;
}

static ge_status extract_impl(ge_glove *h, void *out, bool f64) {
    ge_status st = check_handle(h);
    if (st != GE_OK) return st;
    if (!out) return ge::fail(GE_ERR_ARG, "out is null");
    if (h->rows != h->cfg.vocab_size)
        return ge::fail(GE_ERR_STATE, "extract needs all focus rows on this handle (owned [%d,%d) of %d); gather shards first",
                        h->cfg.row_begin, h->cfg.row_end, h->cfg.vocab_size);
    const int64_t n = (int64_t)h->cfg.vocab_size * h->cfg.dim;
    const size_t bytes = (size_t)n * (f64 ? sizeof(double) : sizeof(float));
    void *d = nullptr;
    GE_HIP(hipMalloc(&d, bytes));
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 8192);
    float *foc = h->tab[GE_STATE_FOCUS], *ctx = h->tab[GE_STATE_CONTEXT];
    const bool temp = h->emb16 || stored_strided(h, GE_STATE_FOCUS);
    if (temp) {
        foc = ctx = nullptr;
        ge_status s1 = materialize_f32(h, GE_STATE_FOCUS, &foc);
        ge_status s2 = s1 == GE_OK ? materialize_f32(h, GE_STATE_CONTEXT, &ctx) : s1;
        if (s2 != GE_OK) { if (foc) (void)hipFree(foc); (void)hipFree(d); return s2; }
    }
    if (f64) hipLaunchKernelGGL(k_extract<double>, dim3(blocks), dim3(256), 0, h->stream, foc, ctx, (double *)d, n);
    else     hipLaunchKernelGGL(k_extract<float>,  dim3(blocks), dim3(256), 0, h->stream, foc, ctx, (float *)d, n);
    hipError_t e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (temp) { (void)hipFree(foc); (void)hipFree(ctx); }
    (void)hipFree(d);
    if (e != hipSuccess) return ge::fail(GE_ERR_HIP, "extract copy failed: %s", hipGetErrorString(e));
    return GE_OK;
}
ge_status ge_glove_extract_f32(ge_glove *h, float *out) { return extract_impl(h, out, false); }
ge_status ge_glove_extract_f64(ge_glove *h, double *out) { return extract_impl(h, out, true); }

ge_status ge_glove_get_state(ge_glove *h, int32_t which, float *out, int64_t count) {
    ge_status st = check_handle(h);
    if (st != GE_OK) return st;
    if (which < 0 || which >= GE_STATE_COUNT || !out) return ge::fail(GE_ERR_ARG, "invalid state id %d or null buffer", which);
    if (count != h->tab_count[which]) return ge::fail(GE_ERR_ARG, "state %d holds %lld floats, caller passed %lld", which, (long long)h->tab_count[which], (long long)count);
    if (h->tab_count[which] == 0) return GE_OK;
    if (stored_strided(h, which) || (h->emb16 && (which == GE_STATE_FOCUS || which == GE_STATE_CONTEXT))) {
        float *d = nullptr;
        st = materialize_f32(h, which, &d);
        if (st != GE_OK) return st;
        hipError_t e = hipMemcpyAsync(out, d, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        (void)hipFree(d);
        if (e != hipSuccess) return ge::fail(GE_ERR_HIP, "state copy failed: %s", hipGetErrorString(e));
        return GE_OK;
    }
    GE_HIP(hipMemcpyAsync(out, h->tab[which], sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, h->stream));
    GE_HIP(hipStreamSynchronize(h->stream));
    return GE_OK;
}

ge_status ge_glove_set_state(ge_glove *h, int32_t which, const float *in, int64_t count) {
    ge_status st = check_handle(h);
    if (st != GE_OK) return st;
    if (which < 0 || which >= GE_STATE_COUNT || !in) return ge::fail(GE_ERR_ARG, "invalid state id %d or null buffer", which);
    if (count != h->tab_count[which]) return ge::fail(GE_ERR_ARG, "state %d holds %lld floats, caller passed %lld", which, (long long)h->tab_count[which], (long long)count);
    if (count == 0) return GE_OK;
    if (stored_strided(h, which)) {
        float *d = nullptr;
        GE_HIP(hipMalloc((void **)&d, sizeof(float) * (size_t)count));
        hipError_t e = hipMemcpyAsync(d, in, sizeof(float) * (size_t)count, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            const Home hm = home_of(h, which);
            hipLaunchKernelGGL(k_fat_scatter, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((count + 255) / 256, 16384))), dim3(256), 0, h->stream,
                               hm.base, count / hm.ncols, (int32_t)hm.stride, hm.col0, hm.ncols, d);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        }
        (void)hipFree(d);
        if (e != hipSuccess) return ge::fail(GE_ERR_HIP, "state copy failed: %s", hipGetErrorString(e));
        return GE_OK;
    }
    if (h->emb16 && (which == GE_STATE_FOCUS || which == GE_STATE_CONTEXT)) {
        float *d = nullptr;
        GE_HIP(hipMalloc((void **)&d, sizeof(float) * (size_t)std::max<int64_t>(count, 1)));
        hipError_t e = hipMemcpyAsync(d, in, sizeof(float) * (size_t)count, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            if (which == GE_STATE_CONTEXT && h->n_hub > 0)
                hipLaunchKernelGGL(k_hub_rows, dim3((unsigned)h->cfg.vocab_size), dim3(64), 0, h->stream, d, h->hub32, h->dhub_index, h->cfg.vocab_size, h->cfg.dim, 0);
            hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)std::min<int64_t>((count + 255) / 256, 8192)), dim3(256), 0, h->stream,
                               d, reinterpret_cast<uint16_t *>(h->tab[which]), count, h->cfg.dim, (int64_t)h->es);
            e = hipStreamSynchronize(h->stream);
        }
        (void)hipFree(d);
        if (e != hipSuccess) return ge::fail(GE_ERR_HIP, "state copy failed: %s", hipGetErrorString(e));
        return GE_OK;
    }
    GE_HIP(hipMemcpyAsync(h->tab[which], in, sizeof(float) * (size_t)count, hipMemcpyHostToDevice, h->stream));
    GE_HIP(hipStreamSynchronize(h->stream));
    return GE_OK;
}

ge_status ge_glove_device_ptr(ge_glove *h, int32_t which, void **dptr, int64_t *count) {
    if (!h) return ge::fail(GE_ERR_ARG, "null ge_glove handle");
    if (which < 0 || which >= GE_STATE_COUNT || !dptr) return ge::fail(GE_ERR_ARG, "invalid state id %d or null out", which);
    if (h->emb16 && (which == GE_STATE_FOCUS || which == GE_STATE_CONTEXT))
        return ge::fail(GE_ERR_STATE, "table %d is stored as bf16 (+ fp32 hub rows); use ge_glove_get_state/set_state", which);
    if (stored_strided(h, which)) {
        // a row table is [n x row_stride]; a bias "table" is one column of its home table (home_of): the pointer is row 0's
        // scalar, consecutive rows are row_stride floats apart.  *count = floats from the returned pointer to the end of the last row's part.
        const Home hm = home_of(h, which);
        *dptr = hm.base + hm.col0;
        const int64_t nr = h->tab_count[which] / hm.ncols;
        if (count) *count = nr > 0 ? (nr - 1) * hm.stride + (is_bias_table(which) ? 1 : h->rw) : 0;
        return GE_OK;
    }
    *dptr = h->tab[which];
    if (count) *count = h->tab_count[which];
    return GE_OK;
}

ge_status ge_glove_context_layout(ge_glove *h, ge_context_layout *out) {
    if (!h || !out) return ge::fail(GE_ERR_ARG, "null argument");
    out->table = h->tab[GE_STATE_CONTEXT];
    out->dtype = h->emb16 ? GE_DTYPE_BF16 : GE_DTYPE_F32;
    out->hub_rows = h->emb16 ? h->hub32 : nullptr;
    out->hub_index = h->emb16 ? h->dhub_index : nullptr;
    out->n_hub = h->emb16 ? h->n_hub : 0;
    out->vocab_size = h->cfg.vocab_size; out->dim = h->cfg.dim;
    out->row_stride = h->emb16 ? h->es : h->ds;
    out->accum = h->tab[GE_STATE_GSQ_CONTEXT];
    out->accum_stride = h->ds;
    const Home cb = home_of(h, GE_STATE_CBIAS), gcb = home_of(h, GE_STATE_GSQ_CBIAS);
    out->bias = cb.base + cb.col0; out->bias_stride = (int32_t)cb.stride;
    out->accum_bias = gcb.base + gcb.col0; out->accum_bias_stride = (int32_t)gcb.stride;
    return GE_OK;
}

ge_status ge_glove_get_perm(ge_glove *h, int32_t *out, int64_t count) {
    if (!h || !out) return ge::fail(GE_ERR_ARG, "null argument");
    if (h->cfg.shuffle != GE_SHUFFLE_JAVA) return ge::fail(GE_ERR_STATE, "no permutation array unless shuffle == GE_SHUFFLE_JAVA");
    if (count != h->cfg.nnz) return ge::fail(GE_ERR_ARG, "perm holds %lld entries", (long long)h->cfg.nnz);
    std::memcpy(out, h->perm.data(), sizeof(int32_t) * (size_t)count);
    return GE_OK;
}

static ge_status ge_glove_epoch_order_impl(ge_glove *h, int32_t iteration, int32_t *out, int64_t count) {
    if (!h || !out) return ge::fail(GE_ERR_ARG, "null argument");
    if (h->cfg.mode != GE_MODE_HOGWILD) return ge::fail(GE_ERR_STATE, "epoch order is defined for GE_MODE_HOGWILD handles");
    const int64_t N = h->cfg.nnz;
    if (count != N) return ge::fail(GE_ERR_ARG, "the epoch visits %lld nonzeros", (long long)N);
    GloveParams p;
    fill_params(h, p, iteration);
    std::vector<int32_t> key, border, cstart;        // blocked layout: copied back from the device (a testing aid, not a hot path)
    if (h->blocked) {
        ge_status st = check_handle(h);
        if (st != GE_OK) return st;
        key.resize((size_t)std::max<int64_t>(N, 1)); border.resize(key.size()); cstart.resize((size_t)h->n_chunks + 1);
        if (N > 0) {
            GE_HIP(hipMemcpy(key.data(), h->lay.bA, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost));
            GE_HIP(hipMemcpy(border.data(), h->lay.border, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost));
        }
        GE_HIP(hipMemcpy(cstart.data(), h->lay.cstart, sizeof(int32_t) * cstart.size(), hipMemcpyDeviceToHost));
    }
    std::vector<std::pair<int32_t, int32_t>> ent;      // (key, original nonzero) of one chunk, in staging order
    int64_t w = 0;
    for (int64_t t = 0; t < h->n_chunks; ++t) {
        ent.clear();
        if (h->blocked) {
            uint32_t x = (uint32_t)t;
            do { x = bij_round(x, p); } while ((int64_t)x >= h->n_chunks);
            for (int64_t k = cstart[(size_t)x]; k < cstart[(size_t)x + 1]; ++k) ent.push_back({key[(size_t)k], border[(size_t)k]});
        } else {
            for (int64_t k = t * RUN_CHUNK; k < std::min<int64_t>((t + 1) * RUN_CHUNK, N); ++k) {
                const int64_t idx = h->cfg.shuffle == GE_SHUFFLE_JAVA ? h->perm[(size_t)k] : k;
                ent.push_back({h->host_key[(size_t)idx], (int32_t)idx});
            }
        }
        std::stable_sort(ent.begin(), ent.end(), [](const std::pair<int32_t, int32_t> &a, const std::pair<int32_t, int32_t> &b) { return a.first < b.first; });
        for (auto &e : ent) out[w++] = e.second;
    }
    if (w != N) return ge::fail(GE_ERR_STATE, "internal: epoch order covers %lld of %lld nonzeros", (long long)w, (long long)N);
    return GE_OK;
}

ge_status ge_glove_rng_state(ge_glove *h, uint64_t *state) {
    if (!h || !state) return ge::fail(GE_ERR_ARG, "null argument");
    *state = h->rng.s;
    return GE_OK;
}

ge_status ge_glove_last_kernel_ms(ge_glove *h, float *ms, int32_t *launches) {
    if (!h) return ge::fail(GE_ERR_ARG, "null ge_glove handle");
    if (ms) *ms = h->last_ms;
    if (launches) *launches = h->last_launches;
    return GE_OK;
}

ge_status ge_glove_get_info(ge_glove *h, ge_glove_info *info) {
    if (!h || !info) return ge::fail(GE_ERR_ARG, "null argument");
    std::memset(info, 0, sizeof(*info));
    info->group_width = 64; info->vector_width = h->hw_vw; info->chunks_per_lane = h->hw_nch;
    info->blocks = h->cfg.mode == GE_MODE_HOGWILD ? h->hw_blocks : 1;
    info->groups_in_flight = h->cfg.mode == GE_MODE_HOGWILD ? h->hw_workers : 1;
    info->hot_columns = h->hot_cols; info->hot_nonzeros = h->hot_nnz; info->hot_threshold = h->hot_threshold;
    info->chunks = h->n_chunks; info->hub_chunks = h->n_hchunks; info->long_rows = h->lay.long_rows; info->shared_chunks = h->lay.shared_chunks;
    info->flush_min = h->flush_every; info->row_stride = h->ds;
    info->placements = h->placements; info->placement_best_ms = h->place_best_ms; info->placement_worst_ms = h->place_worst_ms;
    if (h->blocked) {
        // one row access = the bytes a wavefront's row instruction moves: the row width of the table it touches
        // (the row as the update needs it: dim + 4 floats when fat; what the line-aligned layout pads it with is not counted)
        const int64_t acc = 4ll * (h->cfg.dim + (h->fat ? 4 : 0)), emb = h->emb16 ? 2ll * h->cfg.dim : acc;
        const int64_t aux = h->cfg.opt == GE_OPT_ADAGRAD ? 1 : 2;                 // accumulator rows per side
        const int64_t pair = 2 * (emb + aux * acc) + (h->fat ? 0 : 2 * 4 * (1 + aux));   // load + store of a row, its accumulator row(s) and, unless fat, its scalars
        info->runs = h->lay.n_runs;
        info->schedule_bytes = h->cfg.nnz * (20 + pair) + h->lay.n_runs * pair;
    }
    return GE_OK;
}

}  // extern "C"
namespace ge {
// A Hogwild epoch in `nseg` launches (ge_sync_epoch: the hub rows of a sharded run are reconciled between them).  The chunks of
// the epoch are handed out by ticket through a keyed bijection, so tickets [n seg / nseg, n (seg + 1) / nseg) are a random nseg-th
// of the epoch; the cost accumulates on the device over the segments.  Nothing here blocks the host: glove_epoch_finish does.
// leave_blocks: workgroups NOT launched (their wavefront slots stay free for kernels running beside the epoch: the live hub-row exchange
// of a sharded run); after_reset: recorded once the ticket counter holds this segment's first ticket (a host that watches the counter
// waits for it, so that it never reads the previous epoch's value).
ge_status glove_epoch_segment(ge_glove *h, int32_t iteration, int32_t seg, int32_t nseg, int32_t leave_blocks, hipEvent_t after_reset) {
    ge_status st = check_handle(h);
    if (st != GE_OK) return st;
    if (h->cfg.mode != GE_MODE_HOGWILD || h->cfg.shuffle == GE_SHUFFLE_JAVA) return ge::fail(GE_ERR_STATE, "a segmented epoch needs a GE_MODE_HOGWILD handle with a device-side order");
    if (nseg < 1 || nseg > 64 || seg < 0 || seg >= nseg) return ge::fail(GE_ERR_ARG, "segment %d of %d", seg, nseg);
    GloveParams p;
    fill_params(h, p, iteration);
    const int64_t n = h->cfg.nnz > 0 ? h->n_chunks : 0;
    const int64_t begin = n * seg / nseg, end = n * (seg + 1) / nseg;
    if (seg == 0) {
        GE_HIP(hipMemsetAsync(h->dcost, 0, 2 * sizeof(double), h->stream));
        GE_HIP(hipEventRecord(h->ev0, h->stream));
        h->last_launches = 0;
        h->seg_timed = 0;
    }
    if (end > begin) {
        h->seg_ticket[seg] = (unsigned long long)begin;
        GE_HIP(hipMemcpyAsync(p.queue, &h->seg_ticket[seg], sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
        p.ticket_end = end;
        if (after_reset) GE_HIP(hipEventRecord(after_reset, h->stream));
        hipEvent_t *ev = &h->seg_ev[2 * h->seg_timed];
        for (int k = 0; k < 2; ++k) if (!ev[k]) GE_HIP(hipEventCreate(&ev[k]));
        GE_HIP(hipEventRecord(ev[0], h->stream));
        hipLaunchKernelGGL(h->hw_fn, dim3((unsigned)std::max(1, h->hw_blocks - std::max(0, leave_blocks))), dim3(256), 0, h->stream, p, (int32_t)h->hw_workers);
        GE_HIP(hipEventRecord(ev[1], h->stream));
        ++h->seg_timed;
        ++h->last_launches;
        GE_HIP(hipGetLastError());
    }
    else if (after_reset) GE_HIP(hipEventRecord(after_reset, h->stream));
    if (seg == nseg - 1) GE_HIP(hipEventRecord(h->ev1, h->stream));
    return GE_OK;
}
// the epoch's ticket counter (device memory; tickets [0, *tickets) are the epoch's chunks) and the event behind the last launch
ge_status glove_epoch_progress(ge_glove *h, const unsigned long long **counter, int64_t *tickets, hipEvent_t *done) {
    ge_status st = check_handle(h);
    if (st != GE_OK) return st;
    if (counter) *counter = reinterpret_cast<const unsigned long long *>(h->dcost + 1);
    if (tickets) *tickets = h->cfg.nnz > 0 ? h->n_chunks : 0;
    if (done) *done = h->ev1;
    return GE_OK;
}
// the columns this handle's epoch kernel treats as hubs (resident runs that publish the row and its accumulator row by float atomics)
const std::vector<int32_t> *glove_kernel_hubs(const ge_glove *h) { return (h && h->blocked && h->cfg.hot_columns != GE_HOT_NONE) ? &h->lay.hubs : nullptr; }
const std::vector<int32_t> *glove_hub_counts(const ge_glove *h) { return h ? &h->lay.heavy_count : nullptr; }
ge_status glove_epoch_finish(ge_glove *h, double *cost_sum) {
    ge_status st = check_handle(h);
    if (st != GE_OK) return st;
    double total = 0.0;
    GE_HIP(hipMemcpyAsync(&total, h->dcost, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    GE_HIP(hipStreamSynchronize(h->stream));
    // ge_glove_last_kernel_ms: the launches' own time summed (what the caller's exchanges between them took is the caller's to time)
    h->last_ms = 0.0f;
    for (int32_t k = 0; k < h->seg_timed; ++k) { float ms = 0.0f; GE_HIP(hipEventElapsedTime(&ms, h->seg_ev[2 * k], h->seg_ev[2 * k + 1])); h->last_ms += ms; }
    if (cost_sum) *cost_sum = total;
    return GE_OK;
}
// the busy columns of this handle's shard (ascending; ge_layout.h `heavy`), for the small exchanges of a sharded run
const std::vector<int32_t> *glove_hub_columns(const ge_glove *h) { return h ? &h->lay.heavy : nullptr; }
// what sync.hip needs to know about a handle (struct ge_glove is private to this file)
ge_status glove_sync_view(ge_glove *h, int32_t *opt, int32_t *mode, void **stream, int32_t *device) {
    if (!h) return ge::fail(GE_ERR_ARG, "null ge_glove handle");
    *opt = h->cfg.opt; *mode = h->cfg.mode; *stream = (void *)h->stream; *device = h->cfg.device;
    return GE_OK;
}
}  // namespace ge
extern "C" {

void ge_glove_destroy(ge_glove *h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    for (hipEvent_t e : h->seg_ev) if (e) (void)hipEventDestroy(e);
    for (void *q : h->owned) (void)hipFree(q);
    h->lay.release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
}

// ---- guarded entry points (bodies above allocate on the host) ----
ge_status ge_glove_create(const ge_glove_cfg *cfg, const int32_t *I, const int32_t *J, const float *X, ge_glove **out) {
    GE_GUARD(ge_glove_create_impl(cfg, I, J, X, out));
}
ge_status ge_glove_epoch(ge_glove *h, int32_t iteration, double *cost_sum) { GE_GUARD(ge_glove_epoch_impl(h, iteration, cost_sum)); }
ge_status ge_glove_epoch_order(ge_glove *h, int32_t iteration, int32_t *out, int64_t count) { GE_GUARD(ge_glove_epoch_order_impl(h, iteration, out, count)); }

}  // extern "C"
