// glove.hip -- GloVe / pGloVe AdaGrad trainer for MI355X (gfx950): kernels + C ABI.
//
// Reference semantics (J/ = src/main/java/org/uu/nl/embedding/ of Phaken/graph-embeddings):
//   Optimizer ctor           J/opt/Optimizer.java:34-64     -> k_init_java / k_fill
//   Adagrad.createJob        J/opt/grad/Adagrad.java:42-98  -> k_adagrad_exact (bit-exact) / k_adagrad_hogwild
//   GloveCost / PGloveCost   J/opt/GloveCost.java:7-20, J/opt/PGloveCost.java:7-20 -> cost_terms()
//   Optimizer.extractResult  J/opt/Optimizer.java:129-140   -> k_extract
//
// This file is compiled with -ffp-contract=off: the exact kernel must not fuse a*b+c
// (Java never does); the Hogwild kernel asks for FMAs explicitly where it wants them.
//
// Roofline: HBM.  Algorithmic bytes per pair-update (fp32): read 16*D+28, write 16*D+16
// (two embedding rows, two AdaGrad accumulator rows, four bias/accumulator scalars, one
// 12-byte (i,j,X) triple); SURVEY.md 8(d), DESIGN.md.  No MFMA: sparse gather + length-D dot.

#include "ge_common.h"
#include "ge_javarand.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

enum { ORDER_IDENTITY = 0, ORDER_PERM = 1, ORDER_BIJECTION = 2 };

struct GloveParams {
    float *focus, *context, *fbias, *cbias;
    float *gsf, *gsc, *gsfb, *gscb;
    const int32_t *I, *J;
    const float *X;
    const int32_t *perm;
    double *cost_out;      // Hogwild: one double accumulator
    double xmax;
    int64_t N;
    int32_t D;
    int32_t cost_kind;
    float lr;
    int32_t order_mode;
    uint32_t bij_mask, bij_shift;
    uint32_t bij_key[4];
};

// ---- per-nonzero constants of the cost functions -------------------------------------
// GloveCost:  ic = s + (fB+cB) - log(X);  wc = (X > max) ? ic : (float)pow(X/max, 0.75) * ic
// PGloveCost: ic = s + (fB+cB) - log(X/(1-X)) [fp32 division];  wc = X * ic
// Both reduce to  ic = (float)((double)s + ((double)(fB+cB) - l)),  wc = w * ic.
__device__ __forceinline__ void cost_terms(int kind, float x, double xmax, double &l, float &w) {
    if (kind == GE_COST_GLOVE) {
        l = log((double)x);
        w = ((double)x > xmax) ? 1.0f : (float)pow((double)x / xmax, 0.75);
    } else {
        l = log((double)(x / (1.0f - x)));
        w = x;
    }
}

// ---- epoch order ----------------------------------------------------------------------
// Keyed bijection of [0, 2^b) (odd multiply, xor-shift, add key: each step invertible),
// cycle-walked into [0, N).  2^b < 2N so the expected number of rounds is < 2.
__device__ __forceinline__ uint32_t bij_round(uint32_t x, const GloveParams &p) {
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
__global__ void k_init_java(float *focus, float *context, float *fbias, float *cbias,
                            int32_t row0, int32_t rows, int32_t D, uint64_t seed_state) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int64_t i = (int64_t)row0 + r;
    ge::JavaRandom rng{ge::JavaRandom::jump(seed_state, (uint64_t)i * (uint64_t)(2 + 2 * D))};
    const float fd = (float)D;
    fbias[r] = (float)((double)rng.next_float() - 0.5) / fd;
    cbias[r] = (float)((double)rng.next_float() - 0.5) / fd;
    float *f = focus + (int64_t)r * D, *c = context + (int64_t)r * D;
    for (int32_t d = 0; d < D; ++d) {
        f[d] = (float)((double)rng.next_float() - 0.5) / fd;
        c[d] = (float)((double)rng.next_float() - 0.5) / fd;
    }
}

__global__ void k_fill(float *p, int64_t n, float v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// Optimizer.extractResult: (focus + context) / 2 in fp32, widened on store for the f64 form.
template <typename OUT>
__global__ void k_extract(const float *focus, const float *context, OUT *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (OUT)((focus[i] + context[i]) / 2.0f);
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
        cost_terms(p.cost_kind, x, p.xmax, l, w);
        ic = (float)((double)ic + ((double)(p.fbias[bu] + p.cbias[bv]) - l));
        float wc = w * ic;
        cost = (float)((double)cost + (0.5 * (double)wc) * (double)ic);
        __syncthreads();
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
        __syncthreads();   // the next nonzero may read what this one wrote (same wave, program order)
    }
    if (lane == 0) *job_cost = cost;
}

// ---- Hogwild kernel -----------------------------------------------------------------------
// Block = 256 threads = 256/G lane groups; a group of G lanes owns one nonzero at a time
// (G = 16: one DPP row, four nonzeros in flight per wavefront).  Per tile of 256 nonzeros the
// block first stages (i, j, w, l) in LDS -- one nonzero per lane, so the fp64 log/pow of the cost
// function cost 1/64 of a wave-instruction per nonzero -- then each group walks its share:
// 16-byte coalesced loads of the four rows, fp32 dot reduced across the group, fused AdaGrad
// update, plain (lock-free) stores.  Rows are read once and written once per update.
template <int VW> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<1> { using T = float; };

template <int VW> __device__ __forceinline__ float &comp(typename Vec<VW>::T &v, int c);
template <> __device__ __forceinline__ float &comp<4>(float4 &v, int c) { return (&v.x)[c]; }
template <> __device__ __forceinline__ float &comp<2>(float2 &v, int c) { return (&v.x)[c]; }
template <> __device__ __forceinline__ float &comp<1>(float &v, int) { return v; }

constexpr int HW_TILE = 256;

template <int G, int VW, int NCH>
__global__ __launch_bounds__(256) void k_adagrad_hogwild(GloveParams p, int64_t k_begin, int64_t k_end) {
    using V = typename Vec<VW>::T;
    constexpr int NG = 256 / G;
    __shared__ int32_t s_i[HW_TILE], s_j[HW_TILE];
    __shared__ float   s_w[HW_TILE];
    __shared__ double  s_l[HW_TILE];
    __shared__ double  s_cost[256 / 64];

    const int tid = threadIdx.x;
    const int gl = tid % G;          // lane within the group
    const int grp = tid / G;
    const int32_t D = p.D;
    const float lr = p.lr;
    double cost_acc = 0.0;

    const int64_t n_tiles = (k_end - k_begin + HW_TILE - 1) / HW_TILE;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        {   // stage: one nonzero per lane
            const int64_t k = k_begin + tile * HW_TILE + tid;
            int32_t bu = -1, bv = 0; float w = 0.0f; double l = 0.0;
            if (k < k_end) {
                const int64_t idx = map_index(p, k);
                bu = p.I[idx]; bv = p.J[idx];
                cost_terms(p.cost_kind, p.X[idx], p.xmax, l, w);
            }
            s_i[tid] = bu; s_j[tid] = bv; s_w[tid] = w; s_l[tid] = l;
        }
        __syncthreads();
        for (int e = grp; e < HW_TILE; e += NG) {
            const int32_t bu = s_i[e];
            if (bu < 0) break;                       // tail of the last tile
            const int32_t bv = s_j[e];
            const float w = s_w[e];
            const double l = s_l[e];
            float *foc = p.focus + (int64_t)bu * D, *ctx = p.context + (int64_t)bv * D;
            float *g1s = p.gsf + (int64_t)bu * D,   *g2s = p.gsc + (int64_t)bv * D;

            V f[NCH], c[NCH], gf[NCH], gc[NCH];
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int off = (gl + q * G) * VW;
                if (off < D) {
                    f[q]  = *reinterpret_cast<const V *>(foc + off);
                    c[q]  = *reinterpret_cast<const V *>(ctx + off);
                    gf[q] = *reinterpret_cast<const V *>(g1s + off);
                    gc[q] = *reinterpret_cast<const V *>(g2s + off);
                }
            }
            const float fb = p.fbias[bu], cb = p.cbias[bv];
            const float gfb = p.gsfb[bu], gcb = p.gscb[bv];
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int off = (gl + q * G) * VW;
                if (off < D) {
#pragma unroll
                    for (int t = 0; t < VW; ++t) part = __builtin_fmaf(comp<VW>(f[q], t), comp<VW>(c[q], t), part);
                }
            }
#pragma unroll
            for (int m = G / 2; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);

            const float ic = (float)((double)part + ((double)(fb + cb) - l));
            const float wc = w * ic;
            if (gl == 0) cost_acc += (0.5 * (double)wc) * (double)ic;
            const float wlr = wc * lr;
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int off = (gl + q * G) * VW;
                if (off < D) {
                    V nf, nc, ngf, ngc;
#pragma unroll
                    for (int t = 0; t < VW; ++t) {
                        const float fv = comp<VW>(f[q], t), cv = comp<VW>(c[q], t);
                        const float a = comp<VW>(gf[q], t), b = comp<VW>(gc[q], t);
                        const float grad1 = wc * cv, grad2 = wc * fv;
                        comp<VW>(nf, t)  = __builtin_fmaf(-(wlr * cv), __frsqrt_rn(a), fv);
                        comp<VW>(nc, t)  = __builtin_fmaf(-(wlr * fv), __frsqrt_rn(b), cv);
                        comp<VW>(ngf, t) = __builtin_fmaf(grad1, grad1, a);
                        comp<VW>(ngc, t) = __builtin_fmaf(grad2, grad2, b);
                    }
                    *reinterpret_cast<V *>(foc + off) = nf;
                    *reinterpret_cast<V *>(ctx + off) = nc;
                    *reinterpret_cast<V *>(g1s + off) = ngf;
                    *reinterpret_cast<V *>(g2s + off) = ngc;
                }
            }
            if (gl == 0) {
                p.fbias[bu] = fb - wc * __frsqrt_rn(gfb);      // no learning rate on the biases (Adagrad.java:88-89)
                p.cbias[bv] = cb - wc * __frsqrt_rn(gcb);
                const float w2 = wc * wc;
                p.gsfb[bu] = gfb + w2;
                p.gscb[bv] = gcb + w2;
            }
        }
        __syncthreads();
    }

    // block-reduce the cost, one fp64 atomic per block
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cost_acc += __shfl_xor(cost_acc, m, 64);
    if ((tid & 63) == 0) s_cost[tid >> 6] = cost_acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int q = 0; q < 256 / 64; ++q) t += s_cost[q];
        if (t != 0.0) atomicAdd(p.cost_out, t);
    }
}

using hogwild_fn = void (*)(GloveParams, int64_t, int64_t);

template <int G, int VW>
hogwild_fn pick_nch(int nch) {
    switch (nch) {
        case 1: return k_adagrad_hogwild<G, VW, 1>;
        case 2: return k_adagrad_hogwild<G, VW, 2>;
        case 3: return k_adagrad_hogwild<G, VW, 3>;
        case 4: return k_adagrad_hogwild<G, VW, 4>;
        case 5: return k_adagrad_hogwild<G, VW, 5>;
        case 6: return k_adagrad_hogwild<G, VW, 6>;
        case 7: return k_adagrad_hogwild<G, VW, 7>;
        case 8: return k_adagrad_hogwild<G, VW, 8>;
        default: return nullptr;
    }
}
template <int G>
hogwild_fn pick_vw(int vw, int nch) {
    switch (vw) {
        case 4: return pick_nch<G, 4>(nch);
        case 2: return pick_nch<G, 2>(nch);
        default: return pick_nch<G, 1>(nch);
    }
}
// Chooses the lane-group width G and the chunks per lane for a given D.
hogwild_fn pick_hogwild(int D, int forced_g, int *g_out, int *vw_out, int *nch_out) {
    const int vw = (D % 4 == 0) ? 4 : (D % 2 == 0) ? 2 : 1;
    const int chunks = (D + vw - 1) / vw;
    int cand[3] = {16, 32, 64};
    for (int ci = 0; ci < 3; ++ci) {
        const int g = cand[ci];
        if (forced_g && g != forced_g) continue;
        const int nch = (chunks + g - 1) / g;
        if (nch > 8) continue;
        hogwild_fn fn = g == 16 ? pick_vw<16>(vw, nch) : g == 32 ? pick_vw<32>(vw, nch) : pick_vw<64>(vw, nch);
        if (fn) { *g_out = g; *vw_out = vw; *nch_out = nch; return fn; }
    }
    return nullptr;
}

}  // namespace

// =============================================================================================
// handle
// =============================================================================================
struct ge_glove {
    ge_glove_cfg cfg{};
    int32_t rows = 0;                 // focus rows owned = row_end - row_begin
    float *tab[GE_STATE_COUNT] = {};  // device tables
    int64_t tab_count[GE_STATE_COUNT] = {};
    int32_t *dI = nullptr, *dJ = nullptr, *dperm = nullptr;
    float *dX = nullptr;
    double *dcost = nullptr;          // Hogwild accumulator
    float *djob = nullptr;            // deterministic: per-job fp32 costs
    std::vector<int32_t> perm;        // host copy, GE_SHUFFLE_JAVA
    ge::JavaRandom rng{0};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.0f;
    int32_t last_launches = 0;
    int num_cus = 256;
    hogwild_fn hw_fn = nullptr;
    int hw_g = 0, hw_vw = 0, hw_nch = 0;
    int hw_blocks_per_cu = 8;
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
    // focus-side tables hold rows [row_begin,row_end): rebase so that kernels index by global row id
    const int64_t off = h->cfg.row_begin;
    p.focus -= off * h->cfg.dim;  p.gsf -= off * h->cfg.dim;
    p.fbias -= off;               p.gsfb -= off;
    p.I = h->dI; p.J = h->dJ; p.X = h->dX; p.perm = h->dperm;
    p.cost_out = h->dcost;
    p.xmax = h->cfg.xmax; p.N = h->cfg.nnz; p.D = h->cfg.dim;
    p.cost_kind = h->cfg.cost; p.lr = h->cfg.learning_rate;
    p.order_mode = h->cfg.shuffle == GE_SHUFFLE_JAVA ? ORDER_PERM
                 : h->cfg.shuffle == GE_SHUFFLE_DEVICE ? ORDER_BIJECTION : ORDER_IDENTITY;
    uint32_t bits = 0;
    while (bits < 31 && ((int64_t)1 << bits) < h->cfg.nnz) ++bits;
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

ge_status ge_glove_create(const ge_glove_cfg *cfg, const int32_t *I, const int32_t *J, const float *X,
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
    if (cfg->opt != GE_OPT_ADAGRAD) return ge::fail(GE_ERR_ARG, "Invalid optimization method %d (adagrad only)", cfg->opt);
    if (cfg->threads < 1) return ge::fail(GE_ERR_ARG, "threads must be >= 1");
    if (cfg->mode != GE_MODE_HOGWILD && cfg->mode != GE_MODE_DETERMINISTIC) return ge::fail(GE_ERR_ARG, "invalid mode %d", cfg->mode);
    if (cfg->shuffle < GE_SHUFFLE_JAVA || cfg->shuffle > GE_SHUFFLE_NONE) return ge::fail(GE_ERR_ARG, "invalid shuffle %d", cfg->shuffle);
    int32_t rb = cfg->row_begin, re = cfg->row_end;
    if (rb == 0 && re == 0) re = cfg->vocab_size;
    if (rb < 0 || re > cfg->vocab_size || rb >= re) return ge::fail(GE_ERR_ARG, "invalid row range [%d,%d)", rb, re);
    for (int64_t k = 0; k < cfg->nnz; ++k) {
        if (I[k] < rb || I[k] >= re) return ge::fail(GE_ERR_ARG, "I[%lld]=%d outside owned rows [%d,%d)", (long long)k, I[k], rb, re);
        if (J[k] < 0 || J[k] >= cfg->vocab_size) return ge::fail(GE_ERR_ARG, "J[%lld]=%d outside [0,%d)", (long long)k, J[k], cfg->vocab_size);
    }
    ge_status st = ge::select_device(cfg->device);
    if (st != GE_OK) return st;

    ge_glove *h = new (std::nothrow) ge_glove();
    if (!h) return ge::fail(GE_ERR_OOM, "host allocation failed");
    h->cfg = *cfg;
    h->cfg.row_begin = rb; h->cfg.row_end = re;
    h->rows = re - rb;
    h->stream = (hipStream_t)cfg->stream;
    const int32_t V = cfg->vocab_size, D = cfg->dim;
    const int64_t N = cfg->nnz;

#define GE_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { ge_status _s = ge::fail(_e == hipErrorOutOfMemory ? GE_ERR_OOM : GE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); ge_glove_destroy(h); return _s; } } while (0)

    hipDeviceProp_t prop;
    GE_TRY(hipGetDeviceProperties(&prop, cfg->device));
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    const int64_t counts[GE_STATE_COUNT] = {
        (int64_t)h->rows * D, (int64_t)V * D, h->rows, V, (int64_t)h->rows * D, (int64_t)V * D, h->rows, V};
    for (int t = 0; t < GE_STATE_COUNT; ++t) {
        h->tab_count[t] = counts[t];
        GE_TRY(hipMalloc((void **)&h->tab[t], sizeof(float) * (size_t)std::max<int64_t>(counts[t], 1)));
    }
    const size_t nn = (size_t)std::max<int64_t>(N, 1);
    GE_TRY(hipMalloc((void **)&h->dI, sizeof(int32_t) * nn));
    GE_TRY(hipMalloc((void **)&h->dJ, sizeof(int32_t) * nn));
    GE_TRY(hipMalloc((void **)&h->dX, sizeof(float) * nn));
    GE_TRY(hipMalloc((void **)&h->dcost, sizeof(double)));
    GE_TRY(hipMalloc((void **)&h->djob, sizeof(float) * (size_t)cfg->threads));
    if (N > 0) {
        GE_TRY(hipMemcpyAsync(h->dI, I, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, h->stream));
        GE_TRY(hipMemcpyAsync(h->dJ, J, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, h->stream));
        GE_TRY(hipMemcpyAsync(h->dX, X, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, h->stream));
    }
    if (cfg->shuffle == GE_SHUFFLE_JAVA) {
        GE_TRY(hipMalloc((void **)&h->dperm, sizeof(int32_t) * nn));
        h->perm.resize((size_t)N);
        for (int64_t k = 0; k < N; ++k) h->perm[(size_t)k] = (int32_t)k;     // Permutation ctor
    }
    GE_TRY(hipEventCreate(&h->ev0));
    GE_TRY(hipEventCreate(&h->ev1));

    // --- parameter init in the reference's draw order; context side covers all V rows, the
    //     focus side only the owned rows (same values a single-GPU run would hold there).
    const uint64_t s0 = ge::JavaRandom::scramble(cfg->seed);
    {
        // k_init_java writes focus+context+both biases of a row; run it per side with scratch for the other side.
        // Single-GPU (rows == V): one launch fills everything.
        if (h->rows == V) {
            hipLaunchKernelGGL(k_init_java, dim3((V + 127) / 128), dim3(128), 0, h->stream,
                               h->tab[GE_STATE_FOCUS], h->tab[GE_STATE_CONTEXT], h->tab[GE_STATE_FBIAS], h->tab[GE_STATE_CBIAS],
                               0, V, D, s0);
        } else {
            float *scr_tab = nullptr, *scr_b = nullptr;
            GE_TRY(hipMalloc((void **)&scr_tab, sizeof(float) * (size_t)V * D));
            GE_TRY(hipMalloc((void **)&scr_b, sizeof(float) * (size_t)V));
            // context side: all rows (focus outputs go to scratch)
            hipLaunchKernelGGL(k_init_java, dim3((V + 127) / 128), dim3(128), 0, h->stream,
                               scr_tab, h->tab[GE_STATE_CONTEXT], scr_b, h->tab[GE_STATE_CBIAS], 0, V, D, s0);
            // focus side: owned rows (context outputs go to scratch)
            hipLaunchKernelGGL(k_init_java, dim3((h->rows + 127) / 128), dim3(128), 0, h->stream,
                               h->tab[GE_STATE_FOCUS], scr_tab, h->tab[GE_STATE_FBIAS], scr_b, rb, h->rows, D, s0);
            GE_TRY(hipStreamSynchronize(h->stream));
            (void)hipFree(scr_tab); (void)hipFree(scr_b);
        }
        for (int t = GE_STATE_GSQ_FOCUS; t <= GE_STATE_GSQ_CBIAS; ++t) {    // Adagrad ctor: gradSq = 1
            const int64_t n = h->tab_count[t];
            const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
            hipLaunchKernelGGL(k_fill, dim3(std::max(blocks, 1)), dim3(256), 0, h->stream, h->tab[t], n, 1.0f);
        }
    }
    GE_TRY(hipGetLastError());
    GE_TRY(hipStreamSynchronize(h->stream));
    h->rng.s = ge::JavaRandom::jump(s0, (uint64_t)V * (uint64_t)(2 + 2 * D));

    if (cfg->mode == GE_MODE_HOGWILD) {
        int forced = 0;
        if (const char *e = std::getenv("GE_GLOVE_GROUP")) forced = std::atoi(e);
        if (const char *e = std::getenv("GE_GLOVE_BLOCKS_PER_CU")) h->hw_blocks_per_cu = std::max(1, std::atoi(e));
        h->hw_fn = pick_hogwild(D, forced, &h->hw_g, &h->hw_vw, &h->hw_nch);
        if (!h->hw_fn) { ge_glove_destroy(h); return ge::fail(GE_ERR_ARG, "dim %d not supported by the Hogwild kernel (max 2048 for dim%%4==0)", D); }
    } else if ((size_t)D * sizeof(float) > 64 * 1024) {
        ge_glove_destroy(h);
        return ge::fail(GE_ERR_ARG, "dim %d too large for deterministic mode", D);
    }
#undef GE_TRY
    *out = h;
    return GE_OK;
}

ge_status ge_glove_epoch(ge_glove *h, int32_t iteration, double *cost_sum) {
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
        GE_HIP(hipMemsetAsync(h->dcost, 0, sizeof(double), h->stream));
        GE_HIP(hipEventRecord(h->ev0, h->stream));
        if (N > 0) {
            const int64_t tiles = (N + HW_TILE - 1) / HW_TILE;
            const int blocks = (int)std::min<int64_t>(tiles, (int64_t)h->num_cus * h->hw_blocks_per_cu);
            hipLaunchKernelGGL(h->hw_fn, dim3(blocks), dim3(256), 0, h->stream, p, (int64_t)0, N);
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
    if (f64) hipLaunchKernelGGL(k_extract<double>, dim3(blocks), dim3(256), 0, h->stream, h->tab[GE_STATE_FOCUS], h->tab[GE_STATE_CONTEXT], (double *)d, n);
    else     hipLaunchKernelGGL(k_extract<float>,  dim3(blocks), dim3(256), 0, h->stream, h->tab[GE_STATE_FOCUS], h->tab[GE_STATE_CONTEXT], (float *)d, n);
    hipError_t e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
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
    GE_HIP(hipMemcpyAsync(out, h->tab[which], sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, h->stream));
    GE_HIP(hipStreamSynchronize(h->stream));
    return GE_OK;
}

ge_status ge_glove_set_state(ge_glove *h, int32_t which, const float *in, int64_t count) {
    ge_status st = check_handle(h);
    if (st != GE_OK) return st;
    if (which < 0 || which >= GE_STATE_COUNT || !in) return ge::fail(GE_ERR_ARG, "invalid state id %d or null buffer", which);
    if (count != h->tab_count[which]) return ge::fail(GE_ERR_ARG, "state %d holds %lld floats, caller passed %lld", which, (long long)h->tab_count[which], (long long)count);
    GE_HIP(hipMemcpyAsync(h->tab[which], in, sizeof(float) * (size_t)count, hipMemcpyHostToDevice, h->stream));
    GE_HIP(hipStreamSynchronize(h->stream));
    return GE_OK;
}

ge_status ge_glove_device_ptr(ge_glove *h, int32_t which, void **dptr, int64_t *count) {
    if (!h) return ge::fail(GE_ERR_ARG, "null ge_glove handle");
    if (which < 0 || which >= GE_STATE_COUNT || !dptr) return ge::fail(GE_ERR_ARG, "invalid state id %d or null out", which);
    *dptr = h->tab[which];
    if (count) *count = h->tab_count[which];
    return GE_OK;
}

ge_status ge_glove_get_perm(ge_glove *h, int32_t *out, int64_t count) {
    if (!h || !out) return ge::fail(GE_ERR_ARG, "null argument");
    if (h->cfg.shuffle != GE_SHUFFLE_JAVA) return ge::fail(GE_ERR_STATE, "no permutation array unless shuffle == GE_SHUFFLE_JAVA");
    if (count != h->cfg.nnz) return ge::fail(GE_ERR_ARG, "perm holds %lld entries", (long long)h->cfg.nnz);
    std::memcpy(out, h->perm.data(), sizeof(int32_t) * (size_t)count);
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

void ge_glove_destroy(ge_glove *h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    for (int t = 0; t < GE_STATE_COUNT; ++t) if (h->tab[t]) (void)hipFree(h->tab[t]);
    if (h->dI) (void)hipFree(h->dI);
    if (h->dJ) (void)hipFree(h->dJ);
    if (h->dX) (void)hipFree(h->dX);
    if (h->dperm) (void)hipFree(h->dperm);
    if (h->dcost) (void)hipFree(h->dcost);
    if (h->djob) (void)hipFree(h->djob);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
}

}  // extern "C"
