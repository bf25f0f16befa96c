// ge_layout.h -- the blocked epoch layout of the Hogwild trainer (built on the device by glove_layout.hip, consumed by
// k_adagrad_runs in glove.hip).  Internal to libgeglove.so.
//
// The nonzeros (i, j, X) of one handle are re-ordered once, at create time, into `P = N` positions cut into CHUNKS of at
// most 128 positions -- the unit a worker (one wavefront) pulls from the epoch queue:
//   * H part, chunks [0, n_hchunks): the nonzeros of the HUB columns, column-major (stable by original index), cut every
//     128 positions.  Resident row of a run = the context row j; always published by delta (atomics).
//   * R part, the rest: grouped by focus row i (stable), rows packed whole into chunks (best fit over a few open chunks),
//     so a focus row is resident in ONE worker per epoch.  A row with more than 128 such nonzeros is cut into pieces of
//     128 that occupy consecutive chunks; the pieces run concurrently, so they publish the row by delta like hub columns
//     do (cmeta = the row's id) instead of storing it -- no worker ever overwrites another worker's run.
// Per position: bA = resident row id (j in H, i in R), bB = streamed row id, L = log term (fp64), W = weight (fp32),
// border = index of the nonzero in the caller's arrays.
#pragma once
#include "ge_common.h"
#include <vector>

namespace ge {

constexpr int LAYOUT_CHUNK = 128;

struct LayoutRequest {
    int32_t V, row_begin, row_end;
    int64_t N;
    int32_t cost;            // GE_COST_*
    double  xmax;
    int32_t hot_columns;     // GE_HOT_*
    double  hot_theta;       // column j is a hub when count(j) >= hot_theta * N / workers
    double  stale_budget;    // a hub run is cut every m_j updates, K_j * m_j <= stale_budget
    int32_t flush_every;     // > 0: that limit for every hub run instead
    int32_t workers;         // sequential workers in flight
    int32_t shared_rows;     // 1: pieces of long focus rows publish by delta (default); 0: plain stores (ablation / tests)
    int32_t pack_rows;       // 1: rows packed whole into chunks (default); 0: fixed cuts every 128 positions (ablation / tests)
    bool    want_hub_index;  // bf16 rows: dense index of the hub columns
};

struct BlockedLayout {
    // device arrays
    int32_t *bA = nullptr, *bB = nullptr, *border = nullptr;
    double  *L = nullptr;
    float   *W = nullptr;
    int32_t *cstart = nullptr;      // n_chunks + 1 positions
    int32_t *cmeta = nullptr;       // n_chunks: H chunk: flush limit of its runs; R chunk: id of the row that publishes by delta, or -1
    int64_t P = 0, n_chunks = 0, n_hchunks = 0;
    // what was decided
    int32_t hot_cols = 0; int64_t hot_nnz = 0, hot_threshold = 0;
    int32_t flush_min = LAYOUT_CHUNK;
    int64_t long_rows = 0, shared_chunks = 0;
    int64_t n_runs = 0;              // runs per epoch: times a resident row (+ its accumulator row) is loaded and published
    std::vector<int32_t> hubs;       // the hub columns, ascending
    std::vector<int32_t> heavy;      // columns with count >= max(256, N / 20 480), ascending: what a sharded run reconciles inside an epoch
    std::vector<int32_t> heavy_count; // their nonzero counts on this handle
    std::vector<int32_t> hub_index;  // [V] dense hub index or -1 (want_hub_index)
    int32_t n_hub = 0;
    void release();                  // frees the device arrays
};

// I, J, X: host arrays of the caller (N entries).  Validates the index ranges (same messages as the host loop it replaces).
ge_status build_blocked_layout(const LayoutRequest &rq, const int32_t *I, const int32_t *J, const float *X,
                               hipStream_t stream, BlockedLayout *out);

}  // namespace ge
