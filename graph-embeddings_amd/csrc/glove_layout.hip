// glove_layout.hip -- builds the blocked epoch layout of the Hogwild trainer on the device (ge_layout.h).
//
// Replaces four serial O(N) host passes of round 1 (range check, column count, placement, flush limits).  Pipeline, all on
// the handle's stream (N = nonzeros, V = vocabulary, rows = focus rows owned):
//   upload I, J, X -> k_scan_input (range check + column histogram, wave-aggregated atomics)
//   -> host: hub columns, their dense rank, their flush limits                      O(V)
//   -> k_sort_keys (key = hub ? rank(j) : n_hub + i, per-row count of the rest)     O(N)
//   -> rocprim::radix_sort_pairs (stable: column-major hubs, then the rest grouped by row in matrix order)
//   -> host: rows packed whole into chunks, long rows cut into pieces               O(rows + chunks)
//   -> k_place (position, bA, bB, L, W, border per nonzero; the cost terms are computed here, once per nonzero)
// Roofline: HBM streaming + one radix sort; runs once per handle.
#include "ge_layout.h"
#include "ge_cost.h"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

namespace {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// one atomicAdd per distinct key per wavefront (the hub columns / the row a wavefront sits in would otherwise
// serialise 64 atomics on one address)
__device__ __forceinline__ void wave_count(int32_t *cnt, int32_t key, bool active) {
    bool todo = active;
    for (;;) {
        const unsigned long long m = __ballot(todo);
        if (!m) break;
        const int leader = __ffsll((long long)m) - 1;
        const int32_t k0 = __shfl(key, leader, 64);
        const bool same = todo && key == k0;
        const unsigned long long ms = __ballot(same);
        if (lane_id() == leader) atomicAdd(cnt + k0, (int32_t)__popcll(ms));
        todo = todo && !same;
    }
}

__global__ __launch_bounds__(256) void k_scan_input(const int32_t *I, const int32_t *J, int64_t N, int32_t rb, int32_t re,
                                                    int32_t V, int32_t *col_cnt, unsigned long long *first_bad) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < N; base += stride) {
        const int64_t k = base + threadIdx.x;
        const bool act = k < N;
        const int32_t i = act ? I[k] : rb, j = act ? J[k] : 0;
        const bool bad = act && (i < rb || i >= re || j < 0 || j >= V);
        if (bad) atomicMin(first_bad, (unsigned long long)k);
        wave_count(col_cnt, j, act && !bad);
    }
}

__global__ __launch_bounds__(256) void k_sort_keys(const int32_t *I, const int32_t *J, int64_t N, int32_t rb,
                                                   const int32_t *hub_rank, int32_t n_hub, uint32_t *key, int32_t *val, int32_t *row_cnt) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < N; base += stride) {
        const int64_t k = base + threadIdx.x;
        const bool act = k < N;
        int32_t r = 0; bool rest = false;
        if (act) {
            const int32_t h = hub_rank[J[k]];
            r = I[k] - rb;
            rest = h < 0;
            key[k] = rest ? (uint32_t)n_hub + (uint32_t)r : (uint32_t)h;
            val[k] = (int32_t)k;
        }
        wave_count(row_cnt, r, rest);
    }
}

__global__ __launch_bounds__(256) void k_place(const int32_t *I, const int32_t *J, const float *X, int64_t N, int32_t rb,
                                               const uint32_t *skey, const int32_t *sval, int32_t n_hub, int64_t nH,
                                               const int32_t *row_src, const int32_t *row_dst, int kind, double xmax,
                                               int32_t *bA, int32_t *bB, double *L, float *W, int32_t *border) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += stride) {
        const int32_t k = sval[t];
        const int32_t i = I[k], j = J[k];
        int64_t pos; int32_t a, b;
        if (skey[t] < (uint32_t)n_hub) { pos = t; a = j; b = i; }                 // H: column-major, positions [0, nH)
        else { const int32_t r = i - rb; pos = (int64_t)row_dst[r] + (t - nH - row_src[r]); a = i; b = j; }
        double l; float w;
        cost_terms<false>(kind, X[k], xmax, l, w);
        bA[pos] = a; bB[pos] = b; L[pos] = l; W[pos] = w; border[pos] = k;
    }
}

struct Dev {            // frees every temporary of one build
    std::vector<void *> p;
    ~Dev() { for (void *q : p) (void)hipFree(q); }
    template <typename T> hipError_t alloc(T **out, size_t n) {
        hipError_t e = hipMalloc((void **)out, sizeof(T) * std::max<size_t>(n, 1));
        if (e == hipSuccess) p.push_back((void *)*out);
        return e;
    }
};

int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 16384)); }

}  // namespace

namespace ge {

void BlockedLayout::release() {
    for (void *q : {(void *)bA, (void *)bB, (void *)border, (void *)L, (void *)W, (void *)cstart, (void *)cmeta}) if (q) (void)hipFree(q);
    bA = bB = border = cstart = cmeta = nullptr; L = nullptr; W = nullptr;
}

ge_status build_blocked_layout(const LayoutRequest &rq, const int32_t *I, const int32_t *J, const float *X,
                               hipStream_t stream, BlockedLayout *out) {
    struct Clock {           // GE_GLOVE_TIMING=1 (see ge_glove_create)
        bool on = std::getenv("GE_GLOVE_TIMING") != nullptr;
        hipStream_t stream = nullptr;
        std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
        void lap(const char *what) {
            if (!on) return;
            (void)hipStreamSynchronize(stream);
            const auto n = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[ge_glove_create]   layout: %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
            t = n;
        }
    } clk;
    clk.stream = stream;
    const int64_t N = rq.N;
    const int32_t V = rq.V, rb = rq.row_begin, rows = rq.row_end - rq.row_begin;
    Dev tmp;
    int32_t *dI = nullptr, *dJ = nullptr; float *dX = nullptr;
    int32_t *d_col = nullptr, *d_row = nullptr; unsigned long long *d_bad = nullptr;
    GE_HIP(tmp.alloc(&dI, (size_t)N)); GE_HIP(tmp.alloc(&dJ, (size_t)N)); GE_HIP(tmp.alloc(&dX, (size_t)N));
    GE_HIP(tmp.alloc(&d_col, (size_t)V)); GE_HIP(tmp.alloc(&d_row, (size_t)rows)); GE_HIP(tmp.alloc(&d_bad, 1));
    if (N > 0) {
        GE_HIP(hipMemcpyAsync(dI, I, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, stream));
        GE_HIP(hipMemcpyAsync(dJ, J, sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, stream));
        GE_HIP(hipMemcpyAsync(dX, X, sizeof(float) * (size_t)N, hipMemcpyHostToDevice, stream));
    }
    GE_HIP(hipMemsetAsync(d_col, 0, sizeof(int32_t) * (size_t)V, stream));
    GE_HIP(hipMemsetAsync(d_row, 0, sizeof(int32_t) * (size_t)rows, stream));
    GE_HIP(hipMemsetAsync(d_bad, 0xFF, sizeof(unsigned long long), stream));
    if (N > 0) hipLaunchKernelGGL(k_scan_input, dim3(grid_for(N)), dim3(256), 0, stream, dI, dJ, N, rb, rq.row_end, V, d_col, d_bad);
    GE_HIP(hipGetLastError());
    std::vector<int32_t> cnt((size_t)V);
    unsigned long long bad = ~0ull;
    GE_HIP(hipMemcpyAsync(cnt.data(), d_col, sizeof(int32_t) * (size_t)V, hipMemcpyDeviceToHost, stream));
    GE_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, stream));
    GE_HIP(hipStreamSynchronize(stream));
    clk.lap("upload + column counts");
    if (bad != ~0ull) {
        const int64_t k = (int64_t)bad;
        if (I[k] < rb || I[k] >= rq.row_end) return ge::fail(GE_ERR_ARG, "I[%lld]=%d outside owned rows [%d,%d)", (long long)k, I[k], rb, rq.row_end);
        return ge::fail(GE_ERR_ARG, "J[%lld]=%d outside [0,%d)", (long long)k, J[k], V);
    }

    // ---- hub columns: count(j) * workers >= theta * N; their flush limits ----
    std::vector<int32_t> hub_rank((size_t)V, -1);
    std::vector<int32_t> hubs;                     // hub columns, ascending
    out->hot_cols = 0; out->hot_nnz = 0; out->hot_threshold = 0; out->n_runs = 0;
    // The columns a SHARDED run reconciles inside the epoch (ge_sync_epoch): busy enough that a rank pushes them hard within one epoch.
    // Independent of this handle's worker count (that is about concurrency INSIDE the GPU): count >= N / 20 480 -- what the hub rule
    // gives a full device, 0.25 N / 5 120 -- and at least 256.
    int64_t heavy_thr;
    {
        int64_t div = 20480, floor_n = 256;
        if (const char *e = std::getenv("GE_SYNC_HEAVY_DIV")) div = std::max<int64_t>(1, std::atoll(e));          // experiments (tools/r03/heavy_probe.sh)
        if (const char *e = std::getenv("GE_SYNC_HEAVY_MIN")) floor_n = std::max<int64_t>(1, std::atoll(e));
        // more ranks, more contributors to a row that is summed once per epoch: from four ranks on the set grows with the number of ranks
        // (this handle's share of the rows tells it: 1/8 of them -> twice the divisor).  Six ranks with the eight-rank threshold of the
        // plain rule were 7 % behind the single GPU at epoch 12, with their own 3 % (profiles/r03_six_ranks_hub_threshold.json); the epoch
        // kernel does not mind twice the hub columns on a shard (48.0 ms either way at the bench size).
        const int64_t shards = rq.row_end > rq.row_begin ? std::max<int64_t>(1, (int64_t)V / (int64_t)(rq.row_end - rq.row_begin)) : 1;
        if (shards > 4 && !std::getenv("GE_SYNC_HEAVY_DIV")) div = div * shards / 4;
        heavy_thr = std::max<int64_t>(floor_n, N / div);
    }
    if (N > 0 && rq.hot_columns != GE_HOT_NONE) {
        int64_t thr = rq.hot_columns == GE_HOT_ALL ? 0
                    : std::max<int64_t>(2, (int64_t)std::ceil(rq.hot_theta * (double)N / (double)std::max(rq.workers, 1)));
        // a shard of a larger run: every column the exchange inside the epoch covers is a hub HERE too, i.e. moved by atomic adds only,
        // so that the other ranks' deltas can be added to it while the epoch kernel runs (the live exchange, sync.hip)
        if (rq.row_end - rq.row_begin < V) thr = std::min(thr, heavy_thr);
        for (int32_t v = 0; v < V; ++v)
            if (cnt[(size_t)v] >= thr && cnt[(size_t)v] > 0) { hub_rank[(size_t)v] = (int32_t)hubs.size(); hubs.push_back(v); out->hot_nnz += cnt[(size_t)v]; }
        out->hot_cols = (int32_t)hubs.size();
        out->hot_threshold = thr;
    }
    const int32_t n_hub = (int32_t)hubs.size();
    // Concurrent runs on one hub column add their deltas; each delta is stale by the length of the run.  Summing K
    // concurrent runs of m updates behaves like one step K*m times too long and diverges once K*m*(lr*w*|row|^2) passes
    // ~1 (measured: K*m = 39k diverges, 10k is stable at the bench scale).  A hub run is therefore cut -- delta
    // published, row re-read -- every m_j updates with K_j * m_j <= stale_budget, K_j = count_j * workers / N.
    auto flush_limit = [&](int32_t col) -> int32_t {
        if (rq.flush_every > 0) return std::min<int32_t>(rq.flush_every, LAYOUT_CHUNK);
        const double K = std::max(1.0, (double)cnt[(size_t)col] * (double)rq.workers / (double)std::max<int64_t>(N, 1));
        return (int32_t)std::min<double>(LAYOUT_CHUNK, std::max<double>(4.0, std::floor(rq.stale_budget / K)));
    };
    out->flush_min = rq.flush_every > 0 ? std::min<int32_t>(rq.flush_every, LAYOUT_CHUNK) : LAYOUT_CHUNK;
    for (int32_t c : hubs) out->flush_min = std::min(out->flush_min, flush_limit(c));
    if (rq.want_hub_index) { out->hub_index = hub_rank; out->n_hub = n_hub; }
    out->hubs = hubs;
    out->heavy.clear(); out->heavy_count.clear();
    for (int32_t v = 0; v < V; ++v) if (cnt[(size_t)v] >= heavy_thr) { out->heavy.push_back(v); out->heavy_count.push_back(cnt[(size_t)v]); }

    // ---- stable sort: hubs column-major, the rest grouped by row ----
    clk.lap("hub columns (host)");
    int32_t *d_rank = nullptr; uint32_t *d_key = nullptr, *d_skey = nullptr; int32_t *d_val = nullptr, *d_sval = nullptr;
    GE_HIP(tmp.alloc(&d_rank, (size_t)V));
    GE_HIP(tmp.alloc(&d_key, (size_t)N)); GE_HIP(tmp.alloc(&d_skey, (size_t)N));
    GE_HIP(tmp.alloc(&d_val, (size_t)N)); GE_HIP(tmp.alloc(&d_sval, (size_t)N));
    GE_HIP(hipMemcpyAsync(d_rank, hub_rank.data(), sizeof(int32_t) * (size_t)V, hipMemcpyHostToDevice, stream));
    std::vector<int32_t> rcnt((size_t)rows, 0);
    if (N > 0) {
        hipLaunchKernelGGL(k_sort_keys, dim3(grid_for(N)), dim3(256), 0, stream, dI, dJ, N, rb, d_rank, n_hub, d_key, d_val, d_row);
        GE_HIP(hipGetLastError());
        unsigned bits = 1;
        while (bits < 32 && ((uint64_t)1 << bits) < (uint64_t)n_hub + (uint64_t)rows) ++bits;
        size_t tmp_bytes = 0;
        GE_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_key, d_skey, d_val, d_sval, (size_t)N, 0, bits, stream));
        void *d_tmp = nullptr;
        GE_HIP(tmp.alloc((char **)&d_tmp, tmp_bytes));
        GE_HIP(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_key, d_skey, d_val, d_sval, (size_t)N, 0, bits, stream));
        GE_HIP(hipMemcpyAsync(rcnt.data(), d_row, sizeof(int32_t) * (size_t)rows, hipMemcpyDeviceToHost, stream));
    }
    GE_HIP(hipStreamSynchronize(stream));

    // ---- chunk table ----
    clk.lap("sort");
    const int64_t nH = out->hot_nnz, nR = N - nH;
    std::vector<int32_t> cfill, cmeta;                   // per chunk: positions in it, meta
    // H: fixed cuts; a chunk's flush limit is the smallest limit among the columns inside it
    {
        int64_t pos = 0; size_t hc = 0; int64_t col_end = n_hub ? cnt[(size_t)hubs[0]] : 0;
        while (pos < nH) {
            const int64_t end = std::min<int64_t>(pos + LAYOUT_CHUNK, nH);
            int32_t m = LAYOUT_CHUNK;
            int64_t q = pos;
            int32_t seg[LAYOUT_CHUNK]; int n_seg = 0;      // nonzeros of each column inside the chunk
            while (q < end) {
                while (col_end <= q) { ++hc; col_end += cnt[(size_t)hubs[hc]]; }
                m = std::min(m, flush_limit(hubs[hc]));
                const int64_t q2 = std::min(end, col_end);
                seg[n_seg++] = (int32_t)(q2 - q);
                q = q2;
            }
            for (int k = 0; k < n_seg; ++k) out->n_runs += (seg[k] + m - 1) / m;     // a run per column, cut every m nonzeros
            cfill.push_back((int32_t)(end - pos)); cmeta.push_back(m);
            pos = end;
        }
    }
    out->n_hchunks = (int64_t)cfill.size();
    // R: rows packed whole (best fit over a few open chunks); a long row takes consecutive chunks of its own and leaves
    // its last, partial one open for whole rows behind it
    std::vector<int32_t> row_src((size_t)rows), row_chunk((size_t)rows, -1), row_off((size_t)rows, 0);
    out->long_rows = 0; out->shared_chunks = 0;
    {
        int64_t src = 0;
        constexpr int OPEN = 16;
        int32_t open[OPEN]; int n_open = 0;
        int32_t cur = -1;                                 // pack_rows == 0: the one chunk being filled
        for (int32_t r = 0; r < rows; ++r) {
            row_src[(size_t)r] = (int32_t)src;
            const int32_t len = rcnt[(size_t)r];
            src += len;
            if (len == 0) continue;
            if (!rq.pack_rows) {                          // round-1 layout: cut every 128 positions, rows may straddle chunks
                if (cur < 0 || cfill[(size_t)cur] == LAYOUT_CHUNK) { cur = (int32_t)cfill.size(); cfill.push_back(0); cmeta.push_back(-1); }
                row_chunk[(size_t)r] = cur; row_off[(size_t)r] = cfill[(size_t)cur];
                int32_t left = len;
                while (left > 0) {
                    if (cfill[(size_t)cur] == LAYOUT_CHUNK) { cur = (int32_t)cfill.size(); cfill.push_back(0); cmeta.push_back(-1); }
                    const int32_t take = std::min(left, LAYOUT_CHUNK - cfill[(size_t)cur]);
                    cfill[(size_t)cur] += take; left -= take; ++out->n_runs;
                }
                continue;
            }
            if (len > LAYOUT_CHUNK) {
                ++out->long_rows;
                const int32_t id = rq.shared_rows ? rb + r : -1;
                row_chunk[(size_t)r] = (int32_t)cfill.size(); row_off[(size_t)r] = 0;
                int32_t left = len;
                while (left > 0) {
                    const int32_t take = std::min(left, LAYOUT_CHUNK);
                    cfill.push_back(take); cmeta.push_back(id); ++out->shared_chunks; ++out->n_runs;
                    left -= take;
                }
                if (cfill.back() < LAYOUT_CHUNK) {        // the partial last piece stays open
                    if (n_open == OPEN) { int f = 0; for (int o = 1; o < OPEN; ++o) if (cfill[(size_t)open[o]] > cfill[(size_t)open[f]]) f = o; open[f] = open[--n_open]; }
                    open[n_open++] = (int32_t)cfill.size() - 1;
                }
                continue;
            }
            int best = -1;
            for (int o = 0; o < n_open; ++o) {
                const int32_t room = LAYOUT_CHUNK - cfill[(size_t)open[o]];
                if (room >= len && (best < 0 || room < LAYOUT_CHUNK - cfill[(size_t)open[best]])) best = o;
            }
            if (best < 0) {
                if (n_open == OPEN) { int f = 0; for (int o = 1; o < OPEN; ++o) if (cfill[(size_t)open[o]] > cfill[(size_t)open[f]]) f = o; open[f] = open[--n_open]; }
                open[n_open] = (int32_t)cfill.size(); best = n_open++;
                cfill.push_back(0); cmeta.push_back(-1);
            }
            const int32_t c = open[best];
            row_chunk[(size_t)r] = c; row_off[(size_t)r] = cfill[(size_t)c];
            cfill[(size_t)c] += len; ++out->n_runs;
            if (cfill[(size_t)c] == LAYOUT_CHUNK) open[best] = open[--n_open];
        }
        if (src != nR) return ge::fail(GE_ERR_STATE, "internal: row counts cover %lld of %lld nonzeros", (long long)src, (long long)nR);
    }
    const int64_t n_chunks = (int64_t)cfill.size();
    std::vector<int32_t> cstart((size_t)n_chunks + 1);
    {
        int64_t pos = 0;
        for (int64_t c = 0; c < n_chunks; ++c) { cstart[(size_t)c] = (int32_t)pos; pos += cfill[(size_t)c]; }
        cstart[(size_t)n_chunks] = (int32_t)pos;
        if (pos != N) return ge::fail(GE_ERR_STATE, "internal: chunks cover %lld of %lld nonzeros", (long long)pos, (long long)N);
    }
    std::vector<int32_t> row_dst((size_t)rows, 0);
    for (int32_t r = 0; r < rows; ++r) if (row_chunk[(size_t)r] >= 0) row_dst[(size_t)r] = cstart[(size_t)row_chunk[(size_t)r]] + row_off[(size_t)r];

    // ---- placement ----
    clk.lap("chunk table (host)");
    int32_t *d_src = nullptr, *d_dst = nullptr;
    GE_HIP(tmp.alloc(&d_src, (size_t)rows)); GE_HIP(tmp.alloc(&d_dst, (size_t)rows));
    GE_HIP(hipMemcpyAsync(d_src, row_src.data(), sizeof(int32_t) * (size_t)rows, hipMemcpyHostToDevice, stream));
    GE_HIP(hipMemcpyAsync(d_dst, row_dst.data(), sizeof(int32_t) * (size_t)rows, hipMemcpyHostToDevice, stream));
    const size_t np = (size_t)std::max<int64_t>(N, 1);
    hipError_t e = hipMalloc((void **)&out->bA, sizeof(int32_t) * np);
    if (e == hipSuccess) e = hipMalloc((void **)&out->bB, sizeof(int32_t) * np);
    if (e == hipSuccess) e = hipMalloc((void **)&out->border, sizeof(int32_t) * np);
    if (e == hipSuccess) e = hipMalloc((void **)&out->L, sizeof(double) * np);
    if (e == hipSuccess) e = hipMalloc((void **)&out->W, sizeof(float) * np);
    if (e == hipSuccess) e = hipMalloc((void **)&out->cstart, sizeof(int32_t) * ((size_t)n_chunks + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&out->cmeta, sizeof(int32_t) * (size_t)std::max<int64_t>(n_chunks, 1));
    if (e == hipSuccess) e = hipMemcpyAsync(out->cstart, cstart.data(), sizeof(int32_t) * ((size_t)n_chunks + 1), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && n_chunks) e = hipMemcpyAsync(out->cmeta, cmeta.data(), sizeof(int32_t) * (size_t)n_chunks, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && N > 0) {
        hipLaunchKernelGGL(k_place, dim3(grid_for(N)), dim3(256), 0, stream, dI, dJ, dX, N, rb, d_skey, d_sval, n_hub, nH, d_src, d_dst,
                           rq.cost, rq.xmax, out->bA, out->bB, out->L, out->W, out->border);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    clk.lap("placement kernel");
    if (e != hipSuccess) {
        out->release();
        return ge::fail(e == hipErrorOutOfMemory ? GE_ERR_OOM : GE_ERR_HIP, "blocked layout build failed: %s", hipGetErrorString(e));
    }
    out->P = N; out->n_chunks = n_chunks;
    return GE_OK;
}

}  // namespace ge
