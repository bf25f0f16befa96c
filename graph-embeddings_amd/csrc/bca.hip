// bca.hip -- Bookmark Coloring co-occurrence builder for MI355X (gfx950): kernels + C ABI.
//
// Reference semantics (J/ = src/main/java/org/uu/nl/embedding/ of Phaken/graph-embeddings):
//   BookmarkColoring ctor            J/bca/BookmarkColoring.java:32-120   -> ge_bca_build
//   DirectedWeighted.doWork          J/bca/jobs/DirectedWeighted.java:31-101
//   UndirectedWeighted.doWork        J/bca/jobs/UndirectedWeighted.java:31-114   -> bca_pass()
//   BCAJob.call (fwd + forced reverse, merged)   J/bca/util/BCAJob.java:31-36
//   BCV add / merge / toUnity / toCounts / max   J/bca/util/BCV.java:35-107
//
// One wavefront runs one bookmark at a time (V independent jobs, pulled from a queue).  The paint
// that is still "wet" lives in a per-wave open-addressing table keyed by node id (fp64, as
// PaintedNode.paint) plus a compact list of the ids currently in the TreeMap; the next node is the
// LOWEST id in that list (TreeMap.pollFirstEntry), found with a wave-wide min.  A pop spreads to up
// to 64 neighbours per step, one neighbour per lane (neighbour lists are unique, so lanes never
// collide on a node).  All fp64/fp32 operations are the reference's, in the reference's order
// (-ffp-contract=off), so the values are bit-exact; the row is emitted in java.util.HashMap
// iteration order (bucket = (k ^ k>>>16) & (cap-1); putVal appends at the bin tail, merge() links
// new keys at the bin head and resizes before the lookup).  Rows in which a bin could grow to 8 nodes --
// where the JDK resizes early (treeifyBin below 64 buckets) or builds a red-black tree bin -- are
// replayed exactly by one lane (ge_jhashmap_dev.h).
//
// Roofline: none of HBM/MFMA -- this is latency-bound pointer chasing over a small working set; it
// runs once per graph, against minutes for the JVM.  Measured in DESIGN.md.

#include "ge_common.h"
#include "ge_jhashmap_dev.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <functional>
#include <memory>
#include <new>
#include <thread>
#include <vector>
#include <sys/mman.h>

namespace {

constexpr int32_t KEY_EMPTY = -1;

struct BcaGraph {
    const int64_t *out_ptr, *in_ptr;
    const int32_t *out_idx, *in_idx;
    const float *out_w, *in_w;
    const double *tot_out, *tot_in, *tot_und;
    int32_t V;
};

struct BcaWork {           // per-wave workspace (struct of arrays, `hc` slots each)
    int32_t *hkey;         // node id or KEY_EMPTY
    int32_t *hin;          // 1 while the node is in the TreeMap
    double  *hpaint;       // PaintedNode.paint
    float   *fval, *rval;  // BCV value of the forward / reverse pass
    int32_t *fseq, *rseq;  // BCV insertion sequence (-1 = not in that BCV)
    int32_t *touched;      // slots in use (hc/2)
    int32_t *alist;        // ids currently in the TreeMap (ac)
    unsigned long long *okey;   // sort keys (hc/2)
    int32_t *oslot;        // slot per row entry (hc/2)
    int32_t *rpos;         // reverse-BCV iteration position -> slot (hc/2)
    int32_t *rnew;         // reverse-BCV iteration position -> 1 if the key is new to the forward BCV
    // exact java.util.HashMap replay (ge_jhashmap_dev.h), entries e = index into touched[]
    int32_t *jkey, *jnext, *jprev, *jparent, *jleft, *jright, *jred;   // hc/2 each
    int32_t *ordf, *ordr;  // entries in forward / reverse put order (hc/2 each)
    int32_t *jhead, *jtree;   // per bin (hc each); jhead doubles as the bin histogram of bins_may_treeify
    int32_t *sidx;         // slot -> entry (hc)
};

struct BcaParams {
    BcaGraph g;
    double alpha, epsilon;
    int32_t directed, normalize;
    int32_t row_begin, row_end;
    int32_t hc, hc_log2, ac;           // table slots (power of two), active-list capacity
    char *work;                        // n_waves * work_stride bytes
    int64_t work_stride;
    // outputs
    int32_t *row_n;                    // per bookmark (relative to row_begin)
    int64_t *row_off;
    float *row_max;                    // bcv.max() after normalisation
    int32_t *outJ; float *outX;        // row pool
    int64_t out_cap;
    unsigned long long *pool_used;     // bump allocator
    unsigned long long *queue;         // next bookmark
    unsigned long long *prof;          // GE_BCA_TIMING: [0] shader-clock ticks spent in the passes, [1] in the emission, summed over the wavefronts (else null)
    int32_t *status;                   // 0 ok, 1 table overflow, 2 active-list overflow, 3 pool overflow
    const int32_t *redo;               // second launch: the bookmarks (relative to row_begin) whose rows did not fit
    int32_t n_jobs;                    // number of jobs in this launch (all rows, or the redo list)
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// orders the wave's LDS traffic only: global stores (the list of used slots) stay in flight across it
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}
// minimum of an unsigned value over the wavefront: four DPP steps inside each row of 16 lanes, then the four rows through SGPRs
// (a 64-bit butterfly of ds_bpermute pairs costs six LDS-crossbar round trips -- a quarter of a pop)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    auto step = [](uint32_t x, int ctrl) -> uint32_t {
        uint32_t o;
        switch (ctrl) {
            case 0:  o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0xB1, 0xF, 0xF, false); break;    // quad_perm [1,0,3,2]
            case 1:  o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
            case 2:  o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x141, 0xF, 0xF, false); break;   // row_half_mirror
            default: o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x140, 0xF, 0xF, false); break;   // row_mirror
        }
        return o < x ? o : x;
    };
    v = step(v, 0); v = step(v, 1); v = step(v, 2); v = step(v, 3);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
    return a < b ? a : b;
}
__device__ __forceinline__ unsigned long long lanemask_lt() {
    const unsigned lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}
__device__ __forceinline__ uint32_t java_hash(int32_t k) { const uint32_t h = (uint32_t)k; return h ^ (h >> 16); }
// capacity a java.util.HashMap reaches after `n` putVal insertions (16, load factor 0.75)
__device__ __forceinline__ int32_t java_cap_after_puts(int32_t n) { int32_t c = 16; while (n > (c / 4) * 3) c <<= 1; return c; }

__device__ __forceinline__ BcaWork carve(const BcaParams &p, int wave) {
    char *b = p.work + (int64_t)wave * p.work_stride;
    BcaWork w;
    const int64_t hc = p.hc, half = p.hc / 2;
    w.hpaint = (double *)b;                 b += 8 * hc;
    w.okey = (unsigned long long *)b;       b += 8 * half;
    w.hkey = (int32_t *)b;                  b += 4 * hc;
    w.hin = (int32_t *)b;                   b += 4 * hc;
    w.fval = (float *)b;                    b += 4 * hc;
    w.rval = (float *)b;                    b += 4 * hc;
    w.fseq = (int32_t *)b;                  b += 4 * hc;
    w.rseq = (int32_t *)b;                  b += 4 * hc;
    w.touched = (int32_t *)b;               b += 4 * half;
    w.oslot = (int32_t *)b;                 b += 4 * half;
    w.rpos = (int32_t *)b;                  b += 4 * half;
    w.rnew = (int32_t *)b;                  b += 4 * half;
    w.jkey = (int32_t *)b;                  b += 4 * half;
    w.jnext = (int32_t *)b;                 b += 4 * half;
    w.jprev = (int32_t *)b;                 b += 4 * half;
    w.jparent = (int32_t *)b;               b += 4 * half;
    w.jleft = (int32_t *)b;                 b += 4 * half;
    w.jright = (int32_t *)b;                b += 4 * half;
    w.jred = (int32_t *)b;                  b += 4 * half;
    w.ordf = (int32_t *)b;                  b += 4 * half;
    w.ordr = (int32_t *)b;                  b += 4 * half;
    w.jhead = (int32_t *)b;                 b += 4 * hc;
    w.jtree = (int32_t *)b;                 b += 4 * hc;
    w.sidx = (int32_t *)b;                  b += 4 * hc;
    w.alist = (int32_t *)b;
    return w;
}
inline int64_t work_bytes(int64_t hc, int64_t ac) { return 8 * hc + 8 * (hc / 2) + 4 * hc * 6 + 4 * (hc / 2) * 4 + 4 * (hc / 2) * 9 + 4 * hc * 3 + 4 * ac; }

// find the slot of `key` (must exist); wave-uniform key
__device__ __forceinline__ int32_t table_find(const BcaWork &w, const BcaParams &p, int32_t key) {
    uint32_t slot = ((uint32_t)key * 2654435761u) >> (32 - p.hc_log2);
    for (;;) {
        if (__hip_atomic_load(w.hkey + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == key) return (int32_t)slot;
        slot = (slot + 1) & (uint32_t)(p.hc - 1);
    }
}

// One pass = DirectedWeighted.doWork(reverse) or UndirectedWeighted.doWork.
// mode 0: out-neighbours, 1: in-neighbours, 2: undirected (out then in, one shared total).
// Returns false on overflow.  n_touched / n_seq are wave-uniform running counters.
__device__ bool bca_pass(const BcaParams &p, const BcaWork &w, int32_t bookmark, int mode, bool reverse_bcv,
                         int32_t &n_touched, int32_t &n_seq, int32_t *status) {
    const int lane = threadIdx.x & 63;
    const double alpha = p.alpha, epsilon = p.epsilon;
    float *val = reverse_bcv ? w.rval : w.fval;
    int32_t *seq = reverse_bcv ? w.rseq : w.fseq;
    int32_t an = 0;

    // lane-parallel: each active lane adds paint `pt` to node `nb` (TreeMap containsKey/get/put)
    auto tree_add = [&](bool act, int32_t nb, double pt) -> bool {
        int32_t slot = 0; bool inserted = false;
        if (act) {
            uint32_t s = ((uint32_t)nb * 2654435761u) >> (32 - p.hc_log2);
            for (int probe = 0;; ++probe) {
                const int32_t k = __hip_atomic_load(w.hkey + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (k == nb) break;
                if (k == KEY_EMPTY) {
                    int32_t expected = KEY_EMPTY;
                    if (__hip_atomic_compare_exchange_strong(w.hkey + s, &expected, nb, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WAVEFRONT)) { inserted = true; break; }
                    if (expected == nb) break;
                }
                s = (s + 1) & (uint32_t)(p.hc - 1);
                if (probe > p.hc) break;
            }
            slot = (int32_t)s;
        }
        // register new slots
        const unsigned long long mi = __ballot(inserted);
        if (inserted) {
            const int pos = n_touched + __popcll(mi & lanemask_lt());
            if (pos < p.hc / 2) w.touched[pos] = slot;
            w.hin[slot] = 0; w.fseq[slot] = -1; w.rseq[slot] = -1;
        }
        n_touched += __popcll(mi);
        if (n_touched > p.hc / 2) { if (lane == 0) *status = 1; return false; }
        // paint
        bool fresh = false;
        if (act) {
            if (w.hin[slot]) w.hpaint[slot] = w.hpaint[slot] + pt;      // nodeTree.get(n).addPaint(p)
            else { w.hpaint[slot] = pt; w.hin[slot] = 1; fresh = true; }  // nodeTree.put(n, new PaintedNode(n, p))
        }
        const unsigned long long mf = __ballot(fresh);
        if (fresh) {
            const int pos = an + __popcll(mf & lanemask_lt());
            if (pos < p.ac) w.alist[pos] = nb;
        }
        an += __popcll(mf);
        if (an > p.ac) { if (lane == 0) *status = 2; return false; }
        wave_sync();
        return true;
    };

    if (!tree_add(lane == 0, bookmark, 1.0)) return false;           // nodeTree.put(bookmark, PaintedNode(bookmark, 1))

    while (an > 0) {
        // pollFirstEntry(): lowest node id
        unsigned long long best = ~0ull;
        for (int i = lane; i < an; i += 64) {
            const unsigned long long c = ((unsigned long long)(uint32_t)w.alist[i] << 32) | (uint32_t)i;
            best = c < best ? c : best;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(best, m, 64); best = o < best ? o : best; }
        const int32_t focus = rfl((int)(best >> 32));
        const int32_t fpos = rfl((int)(best & 0xFFFFFFFFull));
        if (lane == 0) w.alist[fpos] = w.alist[an - 1];
        --an;
        const int32_t fslot = table_find(w, p, focus);
        const double wet = w.hpaint[fslot];
        const int32_t old_seq = seq[fslot];
        wave_sync();
        if (lane == 0) {
            w.hin[fslot] = 0;
            // bcv.add(focus, (float)(alpha * wet)): put(key, getOrDefault(key, 0f) + value)
            const float add = (float)(alpha * wet);
            if (old_seq < 0) { seq[fslot] = n_seq; val[fslot] = 0.0f + add; }
            else val[fslot] = val[fslot] + add;
        }
        if (old_seq < 0) ++n_seq;
        wave_sync();
        if (wet < epsilon) continue;

        double total;
        int64_t ob = 0, oe = 0, ib = 0, ie = 0;
        if (mode == 0) { ob = p.g.out_ptr[focus]; oe = p.g.out_ptr[focus + 1]; total = p.g.tot_out[focus]; }
        else if (mode == 1) { ob = p.g.in_ptr[focus]; oe = p.g.in_ptr[focus + 1]; total = p.g.tot_in[focus]; }
        else { ob = p.g.out_ptr[focus]; oe = p.g.out_ptr[focus + 1]; ib = p.g.in_ptr[focus]; ie = p.g.in_ptr[focus + 1]; total = p.g.tot_und[focus]; }
        if (mode != 2) {
            if (oe == ob) continue;          // neighbors.length == 0
            if (total == 0) continue;        // totalWeight == 0 (directed version only)
        }
        const double spread = (1 - alpha) * wet;
        const int32_t *idx0 = mode == 1 ? p.g.in_idx : p.g.out_idx;
        const float *w0 = mode == 1 ? p.g.in_w : p.g.out_w;
        for (int64_t k = ob; k < oe; k += 64) {
            const int64_t kk = k + lane;
            bool act = kk < oe; int32_t nb = 0; double pt = 0;
            if (act) { nb = idx0[kk]; const float weight = w0[kk]; pt = spread * ((double)weight / total); act = !(pt < epsilon); }
            if (!tree_add(act, nb, pt)) return false;
        }
        for (int64_t k = ib; k < ie; k += 64) {   // undirected: in-neighbours after the out-neighbours
            const int64_t kk = k + lane;
            bool act = kk < ie; int32_t nb = 0; double pt = 0;
            if (act) { nb = p.g.in_idx[kk]; const float weight = p.g.in_w[kk]; pt = spread * ((double)weight / total); act = !(pt < epsilon); }
            if (!tree_add(act, nb, pt)) return false;
        }
    }
    return true;
}

// ---- the same pass with its hot state in LDS ------------------------------------------------------------------------------------
// A pop is a chain of dependent table accesses (lowest id of the active list, probe, paint, BCV value, then per neighbour probe /
// insert / paint); with the tables in global memory every link is a round trip to L2 or HBM -- the per-wave tables of 4 096 waves
// are 600 MB, far past the caches -- plus a store acknowledgement at every wavefront fence: about twelve round trips, 6.8 us per
// pop (profiles/r03_bca_*).  Here the table of one bookmark (1 024 slots: key, wet paint, the BCV value and sequence of the pass
// under way) and the active list live in 15 KB of LDS per wavefront (ten wavefronts per CU); what is left in global memory is the
// graph itself and the write-only list of used slots.  paint > 0 <=> the node is in the TreeMap (every paint ever added is >= epsilon > 0).
// A bookmark that outgrows these bounds (more than 512 nodes touched, or 384 at once in the TreeMap: 0.5 % of the rows of a
// DBLP-like graph) is left to the global-memory kernel: status 4.
constexpr int LDS_HC = 768, LDS_MAX = 512, LDS_AC = 384;       // slots (load <= 2/3), nodes per bookmark, nodes at once in the TreeMap
struct BcaHot {
    int32_t *hkey; double *paint; float *val; int16_t *seq; int32_t *alist; int16_t *aslot /* the table slot of alist[i] */;
    int32_t *touched /* global: written here, read by the hand-over */;
};
__device__ __forceinline__ uint32_t lds_home(int32_t key) { return (uint32_t)(((unsigned long long)((uint32_t)key * 2654435761u) * (unsigned long long)LDS_HC) >> 32); }
__device__ __forceinline__ int32_t lds_find(const BcaHot &t, int32_t key) {
    uint32_t slot = lds_home(key);
    for (;;) {
        if (t.hkey[slot] == key) return (int32_t)slot;
        slot = slot + 1 == (uint32_t)LDS_HC ? 0u : slot + 1;
    }
}
__device__ bool bca_pass_lds(const BcaParams &p, const BcaHot &t, int32_t bookmark, int mode, int32_t &n_touched, int32_t &n_seq, int32_t *status) {
    const int lane = threadIdx.x & 63;
    const double alpha = p.alpha, epsilon = p.epsilon;
    int32_t an = 0;
    auto tree_add = [&](bool act, int32_t nb, double pt) -> bool {
        int32_t slot = 0; bool inserted = false;
        if (act) {
            uint32_t s = lds_home(nb);
            for (int probe = 0;; ++probe) {                  // one LDS atomic per probe: claim the slot if it is empty, else learn who holds it
                int32_t expected = KEY_EMPTY;
                if (__hip_atomic_compare_exchange_strong(t.hkey + s, &expected, nb, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)) { inserted = true; break; }
                if (expected == nb) break;
                s = s + 1 == (uint32_t)LDS_HC ? 0u : s + 1;
                if (probe > LDS_HC) break;
            }
            slot = (int32_t)s;
        }
        const unsigned long long mi = __ballot(inserted);
        if (inserted) {
            const int pos = n_touched + __popcll(mi & lanemask_lt());
            if (pos < LDS_MAX) t.touched[pos] = slot;
            t.paint[slot] = -1.0; t.seq[slot] = -1;
        }
        n_touched += __popcll(mi);
        if (n_touched > LDS_MAX) { if (lane == 0) *status = 4; return false; }
        bool fresh = false;
        if (act) {
            const double cur = t.paint[slot];
            if (cur > 0) t.paint[slot] = cur + pt;                        // nodeTree.get(n).addPaint(p)
            else { t.paint[slot] = pt; fresh = true; }                    // nodeTree.put(n, new PaintedNode(n, p))
        }
        const unsigned long long mf = __ballot(fresh);
        if (fresh) {
            const int pos = an + __popcll(mf & lanemask_lt());
            if (pos < LDS_AC) { t.alist[pos] = nb; t.aslot[pos] = (int16_t)slot; }
        }
        an += __popcll(mf);
        if (an > LDS_AC) { if (lane == 0) *status = 4; return false; }
        wave_sync_lds();
        return true;
    };

    if (!tree_add(lane == 0, bookmark, 1.0)) return false;
    while (an > 0) {
        // pollFirstEntry(): the lowest node id of the active list (ids are unique in it, so exactly one lane holds the minimum)
        uint32_t bid = 0xFFFFFFFFu; int32_t bpos = 0;
        for (int i = lane; i < an; i += 64) { const uint32_t id = (uint32_t)t.alist[i]; if (id < bid) { bid = id; bpos = i; } }
        const uint32_t mid = wave_min_u32(bid);
        const int src = __ffsll((long long)__ballot(bid == mid)) - 1;
        const int32_t focus = (int32_t)mid;
        const int32_t fpos = __builtin_amdgcn_readlane(bpos, src);
        const int32_t fslot = t.aslot[fpos];                 // the node's table slot rides beside its id: no probe for the pop
        // the graph reads of this pop do not depend on the table: issued first, they are under way while the table is worked on.
        // (One 64-byte line per vertex holding degree, total and the first six neighbours -- one access instead of two dependent ones
        // -- was tried: bit-identical, and 6 % SLOWER, 97.8 against 92.0 ms; the second read is not what a pop waits for.)
        double total;
        int64_t ob = 0, oe = 0, ib = 0, ie = 0;
        if (mode == 0) { ob = p.g.out_ptr[focus]; oe = p.g.out_ptr[focus + 1]; total = p.g.tot_out[focus]; }
        else if (mode == 1) { ob = p.g.in_ptr[focus]; oe = p.g.in_ptr[focus + 1]; total = p.g.tot_in[focus]; }
        else { ob = p.g.out_ptr[focus]; oe = p.g.out_ptr[focus + 1]; ib = p.g.in_ptr[focus]; ie = p.g.in_ptr[focus + 1]; total = p.g.tot_und[focus]; }
        if (lane == 0) { t.alist[fpos] = t.alist[an - 1]; t.aslot[fpos] = t.aslot[an - 1]; }
        --an;
        const double wet = t.paint[fslot];
        const int32_t old_seq = t.seq[fslot];
        wave_sync_lds();
        if (lane == 0) {
            t.paint[fslot] = -1.0;                                        // pollFirstEntry(): out of the TreeMap
            const float add = (float)(alpha * wet);                       // bcv.add(focus, (float)(alpha * wet))
            if (old_seq < 0) { t.seq[fslot] = (int16_t)n_seq; t.val[fslot] = 0.0f + add; }
            else t.val[fslot] = t.val[fslot] + add;
        }
        if (old_seq < 0) ++n_seq;
        wave_sync_lds();
        if (wet < epsilon) continue;
        if (mode != 2) {
            if (oe == ob) continue;
            if (total == 0) continue;
        }
        const double spread = (1 - alpha) * wet;
        const int32_t *idx0 = mode == 1 ? p.g.in_idx : p.g.out_idx;
        const float *w0 = mode == 1 ? p.g.in_w : p.g.out_w;
        for (int64_t k = ob; k < oe; k += 64) {
            const int64_t kk = k + lane;
            bool act = kk < oe; int32_t nb = 0; double pt = 0;
            if (act) { nb = idx0[kk]; const float weight = w0[kk]; pt = spread * ((double)weight / total); act = !(pt < epsilon); }
            if (!tree_add(act, nb, pt)) return false;
        }
        for (int64_t k = ib; k < ie; k += 64) {   // undirected: in-neighbours after the out-neighbours
            const int64_t kk = k + lane;
            bool act = kk < ie; int32_t nb = 0; double pt = 0;
            if (act) { nb = p.g.in_idx[kk]; const float weight = p.g.in_w[kk]; pt = spread * ((double)weight / total); act = !(pt < epsilon); }
            if (!tree_add(act, nb, pt)) return false;
        }
    }
    return true;
}

// Float.compare-based max/min over a row (BCV.max / BCV.min)
__device__ __forceinline__ int float_compare(float a, float b) {
    if (a < b) return -1;
    if (a > b) return 1;
    const int ia = (a != a) ? 0x7fc00000 : __builtin_bit_cast(int, a);
    const int ib = (b != b) ? 0x7fc00000 : __builtin_bit_cast(int, b);
    return ia == ib ? 0 : (ia < ib ? -1 : 1);
}

// Could any bin of a java.util.HashMap reach 8 nodes while the map receives its keys?  idx(e) = position of entry e in
// the order the keys enter the map (-1: not a key of it).  With the table growing by load factor alone, the key at
// position k enters while size = k <= threshold, so at capacity c exactly the keys with idx <= 0.75 c have entered by
// the time the table leaves that capacity: a histogram of their bins per capacity level finds the first bin that
// reaches 8 (merge() treeifies with the 8th node, putVal with the 9th; 8 covers both).  Until that first event the
// load-factor trajectory IS the map's trajectory, so "no level reaches 8" is exact, not a heuristic; a row that does
// is replayed sequentially (ge_jhashmap_dev.h).  O(n) per row.  Wave-uniform result.
// LOCAL: the histogram (w.jhead) lives in LDS, so the two fences per level need not wait for global stores in flight.
template <bool LOCAL, class IDX>
__device__ bool bins_may_treeify(const BcaParams &p, const BcaWork &w, int32_t n_touched, int32_t n_total, IDX idx) {
    const int lane = threadIdx.x & 63;
    bool flag = false;
    auto sync = [] { if constexpr (LOCAL) wave_sync_lds(); else wave_sync(); };
    for (int32_t c = 16; n_total >= 8; c <<= 1) {
        const int32_t thr = (c / 4) * 3;
        for (int i = lane; i < c; i += 64) w.jhead[i] = 0;
        sync();
        for (int e = lane; e < n_touched; e += 64) {
            const int32_t k = idx(e);
            if (k >= 0 && k <= thr) {
                const int32_t b = (int32_t)(java_hash(w.jkey[e]) & (uint32_t)(c - 1));
                flag |= __hip_atomic_fetch_add(w.jhead + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) + 1 >= 8;
            }
        }
        sync();
        if (__ballot(flag) != 0 || n_total - 1 <= thr || c >= p.hc) break;
    }
    return __ballot(flag) != 0;
}
__device__ __forceinline__ gejm::Map exact_map(const BcaParams &p, const BcaWork &w) {
    gejm::Map m;
    m.key = w.jkey; m.next = w.jnext; m.prev = w.jprev; m.parent = w.jparent; m.left = w.jleft; m.right = w.jright; m.red = w.jred;
    m.head = w.jhead; m.tree = w.jtree; m.max_cap = p.hc;
    gejm::reset(m);
    return m;
}

// LDS: the passes run on the LDS tables above and hand their result to the global workspace `w`, where the emission below finds
// it exactly as the global-memory passes leave it (p.hc = LDS_HC then).
template <bool LDS>
__global__ __launch_bounds__(64) void k_bca(BcaParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x;
    BcaWork w = carve(p, wave);
    // LDS, two views of one buffer: during the passes the hot table of the bookmark (paint 6 144 B | keys 3 072 | BCV values 3 072 |
    // sequences 1 536 | active list 1 536 + its slots 768); during the emission -- the passes' result has been handed to the global workspace by then --
    // the sort keys of the row (4 096 B), the bin histogram / bin heads of the exact replay (4 096) and the row itself in iteration
    // order (values 2 048, keys 2 048), so that ranking, the treeify check and the sequential folds of the normalisation run on LDS.
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[LDS ? 16128 : 16];
    double *const s_paint = reinterpret_cast<double *>(s_raw);
    int32_t *const s_hkey = reinterpret_cast<int32_t *>(s_raw + (LDS ? 6144 : 0));
    float *const s_val = reinterpret_cast<float *>(s_raw + (LDS ? 9216 : 0));
    int16_t *const s_seq = reinterpret_cast<int16_t *>(s_raw + (LDS ? 12288 : 0));
    int32_t *const s_alist = reinterpret_cast<int32_t *>(s_raw + (LDS ? 13824 : 0));
    int16_t *const s_aslot = reinterpret_cast<int16_t *>(s_raw + (LDS ? 15360 : 0));
    float *const e_x = reinterpret_cast<float *>(s_raw + (LDS ? 8192 : 0));
    int32_t *const e_j = reinterpret_cast<int32_t *>(s_raw + (LDS ? 10240 : 0));
    static_assert(LDS_HC == 768 && LDS_AC == 384 && LDS_MAX == 512, "the two LDS views are laid out for these sizes");
    if constexpr (LDS) { w.okey = reinterpret_cast<unsigned long long *>(s_raw); w.jhead = reinterpret_cast<int32_t *>(s_raw + 4096); }
    const BcaHot hot{s_hkey, s_paint, s_val, s_seq, s_alist, s_aslot, w.touched};
    // table starts empty
    if constexpr (LDS) { for (int i = lane; i < LDS_HC; i += 64) s_hkey[i] = KEY_EMPTY; }
    else { for (int i = lane; i < p.hc; i += 64) w.hkey[i] = KEY_EMPTY; }
    wave_sync();
    for (;;) {
        unsigned long long ticket = 0;
        if (lane == 0) ticket = atomicAdd(p.queue, 1ull);
        const int32_t job = rfl((int)ticket);
        if (job >= p.n_jobs || ticket >= (unsigned long long)p.n_jobs) break;
        const int32_t r = p.redo ? p.redo[job] : job;
        const unsigned long long t_start = p.prof ? __builtin_amdgcn_s_memtime() : 0ull;
        const int32_t bookmark = p.row_begin + r;
        int32_t n_touched = 0, nf = 0, nr = 0;
        int32_t status = 0;
        bool ok;
        if constexpr (LDS) {
            ok = bca_pass_lds(p, hot, bookmark, p.directed ? 0 : 2, n_touched, nf, &status);
            const int32_t nt_f = n_touched;
            wave_sync();                                   // the list of used slots (global) is complete and visible
            if (ok) {                                      // the forward BCV leaves LDS: slot list, keys, values, sequences
                for (int e = lane; e < nt_f; e += 64) {
                    const int32_t sl = w.touched[e];
                    w.hkey[sl] = s_hkey[sl]; w.fval[sl] = s_val[sl]; w.fseq[sl] = s_seq[sl]; w.rseq[sl] = -1;
                    s_seq[sl] = -1;                        // the reverse pass keeps its own BCV
                }
                wave_sync();
            }
            if (ok && p.directed) {
                ok = bca_pass_lds(p, hot, bookmark, 1, n_touched, nr, &status);   // DirectedWeighted: reverse = true always
                wave_sync();
                if (ok) {
                    for (int e = lane; e < n_touched; e += 64) {
                        const int32_t sl = w.touched[e];
                        if (e >= nt_f) { w.hkey[sl] = s_hkey[sl]; w.fseq[sl] = -1; }   // first seen by the reverse pass
                        w.rval[sl] = s_val[sl]; w.rseq[sl] = s_seq[sl];
                    }
                    wave_sync();
                }
            }
        } else {
            ok = bca_pass(p, w, bookmark, p.directed ? 0 : 2, false, n_touched, nf, &status);
            if (ok && p.directed) ok = bca_pass(p, w, bookmark, 1, true, n_touched, nr, &status);   // DirectedWeighted: reverse = true always
        }
        status = rfl(status);
        const unsigned long long t_passes = p.prof ? __builtin_amdgcn_s_memtime() : 0ull;
        if (LDS && !ok && status == 4) {                   // outgrew the LDS tables: the global-memory kernel runs this bookmark
            if (lane == 0) { p.row_n[r] = 0; p.row_off[r] = -2; p.row_max[r] = 1.0f; atomicMax(p.status, 4); }
            for (int i = lane; i < LDS_HC; i += 64) s_hkey[i] = KEY_EMPTY;
            wave_sync();
            continue;
        }
        int32_t n_out = 0; float row_max = 1.0f;
        int64_t off = 0;
        if (ok) {
            // ---- reverse BCV iteration order + merge bookkeeping (directed) ----
            int32_t M = 0, cap;
            // entries in the order their keys entered each BCV (for the exact replay), slot -> entry
            for (int e = lane; e < n_touched; e += 64) {
                const int32_t s = w.touched[e];
                w.sidx[s] = e; w.jkey[e] = w.hkey[s];
                if (w.fseq[s] >= 0) w.ordf[w.fseq[s]] = e;
                if (w.rseq[s] >= 0) w.ordr[w.rseq[s]] = e;
            }
            wave_sync();
            bool exact_overflow = false;
            if (p.directed) {
                const int32_t cap_r = java_cap_after_puts(nr);
                const bool rev_exact = bins_may_treeify<LDS>(p, w, n_touched, nr, [&](int e) { return w.rseq[w.touched[e]]; });
                if (rev_exact) {
                    // the reverse BCV is a map built by putVal alone; replay it and walk its bins
                    if (lane == 0) {
                        gejm::Map m = exact_map(p, w);
                        for (int32_t k = 0; k < nr && !m.overflow; ++k) gejm::put_new(m, w.ordr[k]);
                        int32_t t = 0;
                        if (!m.overflow)
                            for (int32_t b = 0; b < m.cap; ++b)
                                for (int32_t e = m.head[b]; e >= 0; e = m.next[e]) {
                                    const int32_t s = w.touched[e];
                                    w.rpos[t] = s; w.rnew[t] = w.fseq[s] < 0; ++t;
                                }
                        if (m.overflow) status = 1;
                    }
                    wave_sync();
                    exact_overflow = rfl(status) == 1;
                } else {
                // position of each reverse entry in the reverse map's iteration order: rank of (bin, seq)
                for (int e = lane; e < n_touched; e += 64) {
                    const int32_t s = w.touched[e];
                    const int32_t sq = w.rseq[s];
                    w.okey[e] = sq < 0 ? ~0ull : (((unsigned long long)(java_hash(w.hkey[s]) & (uint32_t)(cap_r - 1)) << 32) | (uint32_t)sq);
                }
                wave_sync();
                for (int e = lane; e < n_touched; e += 64) {
                    const unsigned long long me = w.okey[e];
                    if (me == ~0ull) continue;
                    int32_t rank = 0;
                    for (int o = 0; o < n_touched; ++o) rank += w.okey[o] < me;
                    const int32_t s = w.touched[e];
                    w.rpos[rank] = s;
                    w.rnew[rank] = w.fseq[s] < 0;
                }
                wave_sync();
                }
                // exclusive prefix of `new` over iteration positions -> merge sequence of new keys
                int32_t running = 0;
                for (int base = 0; base < nr; base += 64) {
                    const int t = base + lane;
                    const int32_t flag = t < nr ? w.rnew[t] : 0;
                    const unsigned long long mk = __ballot(flag != 0);
                    if (t < nr) w.rnew[t] = flag ? running + __popcll(mk & lanemask_lt()) : -1;
                    running += __popcll(mk);
                }
                M = running;
                wave_sync();
                // size before the last merge() call decides the final capacity (resize happens BEFORE the lookup)
                int32_t size_before_last = nf;
                if (nr > 0) size_before_last = nf + M - (w.rnew[nr - 1] >= 0 ? 1 : 0);
                cap = java_cap_after_puts(nf);
                while (size_before_last > (cap / 4) * 3) cap <<= 1;
                if (nr == 0) cap = java_cap_after_puts(nf);
                // merged values; a key new to the forward map remembers its merge sequence m as fseq = -(m+2)
                for (int t = lane; t < nr; t += 64) {
                    const int32_t s = w.rpos[t];
                    const int32_t m = w.rnew[t];
                    if (m < 0) w.fval[s] = w.fval[s] + w.rval[s];                     // Float.sum(old, value)
                    else { w.fval[s] = w.rval[s]; w.fseq[s] = -(m + 2); }
                }
                wave_sync();
                n_out = nf + M;
                // final order: bins ascending; inside a bin the merged-new keys (linked at the bin HEAD, so in
                // reverse merge order) come before the forward keys (appended at the TAIL, in put order)
                for (int e = lane; e < n_touched; e += 64) {
                    const int32_t s = w.touched[e];
                    const uint32_t bin = java_hash(w.hkey[s]) & (uint32_t)(cap - 1);
                    const int32_t fs = w.fseq[s];
                    unsigned long long key = ~0ull;
                    if (fs >= 0) key = ((unsigned long long)bin << 33) | (1ull << 32) | (uint32_t)fs;
                    else if (fs <= -2) key = ((unsigned long long)bin << 33) | (uint32_t)(M - 1 - (-fs - 2));
                    w.okey[e] = key;
                }
                wave_sync();
            } else {
                cap = java_cap_after_puts(nf);
                n_out = nf;
                for (int e = lane; e < n_touched; e += 64) {
                    const int32_t s = w.touched[e];
                    const uint32_t bin = java_hash(w.hkey[s]) & (uint32_t)(cap - 1);
                    w.okey[e] = w.fseq[s] >= 0 ? (((unsigned long long)bin << 33) | (uint32_t)w.fseq[s]) : ~0ull;
                }
                wave_sync();
            }
            // The forward BCV (+ the merged keys): position of a key in the order the keys entered the map = its put
            // sequence, or nf + its merge sequence.  A row that could fill a bin is replayed exactly instead: puts in
            // sequence, then BCV.merge over the reverse BCV's iteration order, then -- normalised rows --
            // remove(rootNode), which in a tree bin is not a plain unlink.  okey becomes the key's iteration position;
            // the root of a normalised row is ranked first, where the code below finds and drops it.
            const bool fwd_exact = !exact_overflow && bins_may_treeify<LDS>(p, w, n_touched, n_out, [&](int e) {
                const int32_t fs = w.fseq[w.touched[e]];
                return fs >= 0 ? fs : (fs <= -2 ? nf + (-fs - 2) : -1);
            });
            if (fwd_exact) {
                if (lane == 0) {
                    gejm::Map m = exact_map(p, w);
                    for (int32_t k = 0; k < nf && !m.overflow; ++k) gejm::put_new(m, w.ordf[k]);
                    if (p.directed)
                        for (int32_t t = 0; t < nr && !m.overflow; ++t) gejm::merge(m, w.sidx[w.rpos[t]], w.rnew[t] >= 0);
                    if (!m.overflow) {
                        const bool drop = p.normalize != GE_NORM_NONE;
                        const int32_t root_e = w.ordf[0];                    // the bookmark is the first key of the forward BCV
                        if (drop) gejm::remove(m, root_e);
                        for (int32_t e = 0; e < n_touched; ++e) w.okey[e] = ~0ull;
                        unsigned long long pos = drop ? 1 : 0;
                        if (drop) w.okey[root_e] = 0;
                        for (int32_t b = 0; b < m.cap; ++b)
                            for (int32_t e = m.head[b]; e >= 0; e = m.next[e]) w.okey[e] = pos++;
                    }
                    if (m.overflow) status = 1;
                }
                wave_sync();
                exact_overflow = rfl(status) == 1;
            }
            if (exact_overflow) { ok = false; status = 1; }
            const int32_t n_cand = n_touched;
            // ---- normalisation needs the row in iteration order: rank every candidate ----
            const bool drop_root = p.normalize != GE_NORM_NONE;
            if (lane == 0) {
                const unsigned long long o = atomicAdd(p.pool_used, (unsigned long long)n_out);
                off = (int64_t)o;
            }
            off = ((int64_t)(unsigned)rfl((int)(off >> 32)) << 32) | (unsigned)rfl((int)(off & 0xFFFFFFFFll));
            bool fits = true;
            if (off + n_out > p.out_cap) { fits = false; }      // the row size is known: it is re-run alone into an exact pool
            if (!fits) {
                if (lane == 0) { p.row_n[r] = n_out; p.row_off[r] = -1; p.row_max[r] = 1.0f; atomicMax(p.status, 3); }
                if constexpr (LDS) { for (int i = lane; i < LDS_HC; i += 64) s_hkey[i] = KEY_EMPTY; }       // (the emission has used the LDS)
                else { for (int e = lane; e < n_touched && e < p.hc / 2; e += 64) w.hkey[w.touched[e]] = KEY_EMPTY; }
                wave_sync();
                continue;
            }
            if (ok && LDS) {
                // the row in iteration order, in LDS
                for (int e = lane; e < n_cand; e += 64) {
                    const unsigned long long me = w.okey[e];
                    if (me == ~0ull) continue;
                    int32_t rank = 0;
                    for (int o = 0; o < n_cand; ++o) rank += w.okey[o] < me;
                    const int32_t s = w.touched[e];
                    e_j[rank] = w.hkey[s];
                    e_x[rank] = w.fval[s];
                }
                wave_sync_lds();
                int32_t n = n_out;
                if (p.normalize != GE_NORM_NONE) {
                    // BCV.toCounts / remove(rootNode) / toUnity: sequential float folds in iteration order, by one lane, on LDS
                    if (lane == 0) {
                        if (p.normalize == GE_NORM_COUNTS) {
                            float aMax = 1.0f, aMin = 0.0f;
                            for (int k = 0; k < n; ++k) {
                                const float v = e_x[k];
                                if (k == 0) { aMax = v; aMin = v; }
                                else { aMax = float_compare(aMax, v) >= 0 ? aMax : v; aMin = float_compare(aMin, v) <= 0 ? aMin : v; }
                            }
                            for (int k = 0; k < n; ++k) e_x[k] = (e_x[k] / ((aMax - aMin) / (1000.0f - 1.0f))) + 1.0f;
                        }
                        int k = 0;
                        while (k < n && e_j[k] != bookmark) ++k;                   // remove(rootNode)
                        for (int q = k; q + 1 < n; ++q) { e_j[q] = e_j[q + 1]; e_x[q] = e_x[q + 1]; }
                        if (k < n) --n;
                        if (p.normalize == GE_NORM_UNITY) {
                            float sum = 0.0f;
                            for (int q = 0; q < n; ++q) sum = q == 0 ? e_x[q] : sum + e_x[q];      // reduce(Float::sum)
                            for (int q = 0; q < n; ++q) e_x[q] = e_x[q] / sum - 1e-6f;
                        }
                    }
                    wave_sync_lds();
                    n = rfl(n);
                }
                // BCV.max(): the maximum under Float.compare is a total-order maximum -- any association gives the sequential fold's result
                float mx = 0.0f; bool have = false;
                for (int k = lane; k < n; k += 64) { const float v = e_x[k]; mx = !have ? v : (float_compare(mx, v) >= 0 ? mx : v); have = true; }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    const float o = __shfl_xor(mx, m, 64); const int oh = __shfl_xor((int)have, m, 64);
                    if (oh) { mx = !have ? o : (float_compare(mx, o) >= 0 ? mx : o); have = true; }
                }
                row_max = have ? mx : 1.0f;                                   // max().orElse(1f)
                for (int k = lane; k < n; k += 64) { p.outJ[off + k] = e_j[k]; p.outX[off + k] = e_x[k]; }
                n_out = n;
            }
            if (ok && !LDS) {
                for (int e = lane; e < n_cand; e += 64) {
                    const unsigned long long me = w.okey[e];
                    if (me == ~0ull) continue;
                    int32_t rank = 0;
                    for (int o = 0; o < n_cand; ++o) rank += w.okey[o] < me;
                    const int32_t s = w.touched[e];
                    p.outJ[off + rank] = w.hkey[s];
                    p.outX[off + rank] = w.fval[s];
                }
                wave_sync();
                __threadfence();
                // ---- BCV.toUnity / toCounts (sequential float folds in iteration order), BCV.max ----
                if (lane == 0) {
                    int32_t *oj = p.outJ + off; float *ox = p.outX + off;
                    int32_t n = n_out;
                    if (p.normalize == GE_NORM_COUNTS) {
                        float aMax = 1.0f, aMin = 0.0f;
                        for (int k = 0; k < n; ++k) {
                            const float v = __hip_atomic_load(ox + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (k == 0) { aMax = v; aMin = v; }
                            else { aMax = float_compare(aMax, v) >= 0 ? aMax : v; aMin = float_compare(aMin, v) <= 0 ? aMin : v; }
                        }
                        for (int k = 0; k < n; ++k) {
                            const float v = __hip_atomic_load(ox + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ox[k] = (v / ((aMax - aMin) / (1000.0f - 1.0f))) + 1.0f;
                        }
                    }
                    if (drop_root) {                                   // remove(rootNode)
                        int k = 0;
                        while (k < n && __hip_atomic_load(oj + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != bookmark) ++k;
                        for (int q = k; q + 1 < n; ++q) {
                            oj[q] = __hip_atomic_load(oj + q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ox[q] = __hip_atomic_load(ox + q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (k < n) --n;
                    }
                    if (p.normalize == GE_NORM_UNITY) {
                        float sum = 0.0f;
                        for (int k = 0; k < n; ++k) {
                            const float v = __hip_atomic_load(ox + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            sum = k == 0 ? v : sum + v;                // reduce(Float::sum)
                        }
                        for (int k = 0; k < n; ++k) ox[k] = __hip_atomic_load(ox + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / sum - 1e-6f;
                    }
                    float mx = 1.0f;                                   // max().orElse(1f)
                    for (int k = 0; k < n; ++k) {
                        const float v = __hip_atomic_load(ox + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        mx = k == 0 ? v : (float_compare(mx, v) >= 0 ? mx : v);
                    }
                    row_max = mx;
                    n_out = n;
                }
                n_out = rfl(n_out);
            }
        }
        if (lane == 0) {
            p.row_n[r] = ok ? n_out : 0;
            p.row_off[r] = (LDS && !ok) ? -2 : off;                   // (the LDS kernel's tables do not grow: the other kernel takes the row)
            p.row_max[r] = row_max;
            if (!ok) atomicMax(p.status, LDS ? 4 : (status ? status : 1));
        }
        if (p.prof && lane == 0) {
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            atomicAdd(p.prof, t_passes - t_start); atomicAdd(p.prof + 1, t_end - t_passes);
        }
        // reset the table for the next bookmark
        if constexpr (LDS) { for (int i = lane; i < LDS_HC; i += 64) s_hkey[i] = KEY_EMPTY; }       // (the emission has used the LDS)
        else {
            for (int e = lane; e < n_touched && e < p.hc / 2; e += 64) w.hkey[w.touched[e]] = KEY_EMPTY;
            if (!ok && status == 1) for (int i = lane; i < p.hc; i += 64) w.hkey[i] = KEY_EMPTY;   // touched[] was truncated
        }
        wave_sync();
    }
}

// totalWeight per vertex, summed sequentially in neighbour order in fp64 exactly as the Java loops do
// (DirectedWeighted.java:69-75; UndirectedWeighted.java:61-73: out-neighbours then in-neighbours into ONE accumulator).
__global__ void k_totals(BcaGraph g, double *tot_out, double *tot_in, double *tot_und) {
    const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= g.V) return;
    double a = 0;
    for (int64_t k = g.out_ptr[v]; k < g.out_ptr[v + 1]; ++k) a += g.out_w[k];
    tot_out[v] = a;
    double u = a;
    double b = 0;
    for (int64_t k = g.in_ptr[v]; k < g.in_ptr[v + 1]; ++k) { b += g.in_w[k]; u += g.in_w[k]; }
    tot_in[v] = b;
    tot_und[v] = u;
}

// rows from the pool -> bookmark order
__global__ void k_gather_rows(const int32_t *poolJ, const float *poolX, const int32_t *pool2J, const float *pool2X, int64_t pool_cap,
                              const int64_t *row_off, const int32_t *row_n,
                              const int64_t *dst_off, int32_t n_rows, int32_t row_begin,
                              int32_t *I, int32_t *J, float *X) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x / 64;
    for (int32_t r = blockIdx.x * wpb + (threadIdx.x >> 6); r < n_rows; r += gridDim.x * wpb) {
        int64_t so = row_off[r];
        const int64_t d = dst_off[r];
        const int32_t n = row_n[r];
        const int32_t *sJ = poolJ; const float *sX = poolX;
        if (so >= pool_cap) { so -= pool_cap; sJ = pool2J; sX = pool2X; }     // row lives in the second pool
        for (int k = lane; k < n; k += 64) { I[d + k] = row_begin + r; J[d + k] = sJ[so + k]; X[d + k] = sX[so + k]; }
    }
}

}  // namespace

// GE_BCA_TIMING=1 prints the phases of ge_bca_build to stderr
struct PhaseClock {
    bool on = std::getenv("GE_BCA_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[ge_bca_build] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

struct ge_coo {
    int64_t nnz = 0;
    int32_t V = 0;
    struct Free { void operator()(void *q) const { std::free(q); } };
    std::unique_ptr<int32_t[], Free> I, J;      // nnz entries each; left uninitialised until the device copy fills them
    std::unique_ptr<float[], Free> X;           //   (a value-initialising container would touch 12 bytes per entry once more)
    int64_t capacity = 0;                       // entries each array has room for (>= nnz)
    // Fresh host memory costs more than the copy into it: 520 MB of J and X arrive in 9 ms once their pages exist and in 35 - 60 ms when
    // every page is met for the first time (tools/r03/pinned_probe.py).  The arrays are therefore allocated BEFORE the main launch, from
    // the sample's estimate of the total, and their pages are touched by host threads while the device works.
    bool reserve(int64_t entries) {
        const size_t bytes = ((size_t)std::max<int64_t>(entries, 1) * 4 + 4095) / 4096 * 4096;
        void *q[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < 3; ++k)
            if (posix_memalign(&q[k], (size_t)1 << 21, bytes) != 0) { for (int j = 0; j < k; ++j) std::free(q[j]); return false; }
        for (int k = 0; k < 3; ++k) (void)madvise(q[k], bytes, MADV_HUGEPAGE);      // where the system allows it: 2 MB pages, 512 times fewer faults
        I.reset(static_cast<int32_t *>(q[0])); J.reset(static_cast<int32_t *>(q[1])); X.reset(static_cast<float *>(q[2]));
        capacity = (int64_t)(bytes / 4);
        return true;
    }
    // one write per 4 KB page of the first `entries` entries of the three arrays, slice t of n
    void touch(int64_t entries, int t, int n) {
        const int64_t pages = (std::min(entries, capacity) * 4 + 4095) / 4096;
        for (int a = 0; a < 3; ++a) {
            volatile char *base = a == 0 ? reinterpret_cast<char *>(I.get()) : a == 1 ? reinterpret_cast<char *>(J.get()) : reinterpret_cast<char *>(X.get());
            for (int64_t pg = pages * t / n, p1 = pages * (t + 1) / n; pg < p1; ++pg) base[pg * 4096] = 0;
        }
    }
    std::vector<int64_t> row_ptr;
    double max = 0;
};

namespace {

template <typename T>
ge_status upload(const T *src, size_t n, T **dst) {
    GE_HIP(hipMalloc((void **)dst, sizeof(T) * std::max<size_t>(n, 1)));
    if (n) GE_HIP(hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice));
    return GE_OK;
}

struct DevKeeper {      // frees every device buffer of one ge_bca_build call
    std::vector<void *> ptrs;
    ~DevKeeper() { for (void *q : ptrs) (void)hipFree(q); }
    template <typename T> T *keep(T *q) { ptrs.push_back((void *)q); return q; }
};

// Math.max(double, double)
double java_math_max(double a, double b) {
    if (a != a) return a;
    if (a == 0.0 && b == 0.0) return std::signbit(a) ? b : a;
    return a >= b ? a : b;
}

}  // namespace

extern "C" {

static ge_status ge_bca_build_impl(const ge_csr *out_nbrs, const ge_csr *in_nbrs, const ge_bca_cfg *cfg, ge_coo **result) {
    if (!result) return ge::fail(GE_ERR_ARG, "result is null");
    *result = nullptr;
    if (!out_nbrs || !in_nbrs || !cfg) return ge::fail(GE_ERR_ARG, "null argument");
    const int32_t V = out_nbrs->num_vertices;
    if (V <= 0 || in_nbrs->num_vertices != V) return ge::fail(GE_ERR_ARG, "out/in neighbourhoods must describe the same V > 0 vertices");
    if (!(cfg->alpha > 0) || !(cfg->epsilon > 0))
        return ge::fail(GE_ERR_ARG, "Invalid BCA parameters, alpha and epsilon are mandatory");       // Configuration.check
    if (cfg->normalize < GE_NORM_NONE || cfg->normalize > GE_NORM_COUNTS) return ge::fail(GE_ERR_ARG, "invalid normalize %d", cfg->normalize);
    if (cfg->table_slots < 0 || cfg->pool_entries < 0) return ge::fail(GE_ERR_ARG, "table_slots / pool_entries must be >= 0");
    if (!out_nbrs->ptr || !in_nbrs->ptr) return ge::fail(GE_ERR_ARG, "null CSR pointer array");
    int32_t rb = cfg->row_begin, re = cfg->row_end;
    if (rb == 0 && re == 0) re = V;
    if (rb < 0 || re > V || rb >= re) return ge::fail(GE_ERR_ARG, "invalid bookmark range [%d,%d)", rb, re);
    PhaseClock clk;
    const int64_t Eo = out_nbrs->ptr[V], Ei = in_nbrs->ptr[V];
    if (out_nbrs->ptr[0] != 0 || in_nbrs->ptr[0] != 0 || Eo < 0 || Ei < 0) return ge::fail(GE_ERR_ARG, "CSR offsets must start at 0");
    for (int32_t v = 0; v < V; ++v)
        if (out_nbrs->ptr[v + 1] < out_nbrs->ptr[v] || in_nbrs->ptr[v + 1] < in_nbrs->ptr[v]) return ge::fail(GE_ERR_ARG, "CSR offsets must be non-decreasing (vertex %d)", v);
    if ((Eo && (!out_nbrs->idx || !out_nbrs->weight)) || (Ei && (!in_nbrs->idx || !in_nbrs->weight))) return ge::fail(GE_ERR_ARG, "null CSR index/weight array");
    for (int64_t k = 0; k < Eo; ++k) if (out_nbrs->idx[k] < 0 || out_nbrs->idx[k] >= V) return ge::fail(GE_ERR_ARG, "out neighbour %lld = %d outside [0,%d)", (long long)k, out_nbrs->idx[k], V);
    for (int64_t k = 0; k < Ei; ++k) if (in_nbrs->idx[k] < 0 || in_nbrs->idx[k] >= V) return ge::fail(GE_ERR_ARG, "in neighbour %lld = %d outside [0,%d)", (long long)k, in_nbrs->idx[k], V);
    ge_status st = ge::select_device(cfg->device);
    if (st != GE_OK) return st;

    DevKeeper dev;

    BcaParams p{};
    int64_t *d_optr = nullptr, *d_iptr = nullptr; int32_t *d_oidx = nullptr, *d_iidx = nullptr; float *d_ow = nullptr, *d_iw = nullptr;
    if ((st = upload(out_nbrs->ptr, (size_t)V + 1, &d_optr)) != GE_OK) return st; dev.keep(d_optr);
    if ((st = upload(in_nbrs->ptr, (size_t)V + 1, &d_iptr)) != GE_OK) return st;  dev.keep(d_iptr);
    if ((st = upload(out_nbrs->idx, (size_t)Eo, &d_oidx)) != GE_OK) return st;    dev.keep(d_oidx);
    if ((st = upload(in_nbrs->idx, (size_t)Ei, &d_iidx)) != GE_OK) return st;     dev.keep(d_iidx);
    if ((st = upload(out_nbrs->weight, (size_t)Eo, &d_ow)) != GE_OK) return st;   dev.keep(d_ow);
    if ((st = upload(in_nbrs->weight, (size_t)Ei, &d_iw)) != GE_OK) return st;    dev.keep(d_iw);
    double *d_tot = nullptr;
    GE_HIP(hipMalloc((void **)&d_tot, sizeof(double) * 3 * (size_t)V)); dev.keep(d_tot);
    p.g = BcaGraph{d_optr, d_iptr, d_oidx, d_iidx, d_ow, d_iw, d_tot, d_tot + V, d_tot + 2 * (size_t)V, V};
    hipLaunchKernelGGL(k_totals, dim3((V + 255) / 256), dim3(256), 0, 0, p.g, d_tot, d_tot + V, d_tot + 2 * (size_t)V);
    GE_HIP(hipGetLastError());

    p.alpha = cfg->alpha; p.epsilon = cfg->epsilon; p.directed = cfg->directed ? 1 : 0; p.normalize = cfg->normalize;
    p.row_begin = rb; p.row_end = re;
    const int32_t n_rows = re - rb;

    hipDeviceProp_t prop;
    GE_HIP(hipGetDeviceProperties(&prop, cfg->device));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    int32_t *d_row_n = nullptr; int64_t *d_row_off = nullptr; float *d_row_max = nullptr;
    GE_HIP(hipMalloc((void **)&d_row_n, sizeof(int32_t) * (size_t)n_rows)); dev.keep(d_row_n);
    GE_HIP(hipMalloc((void **)&d_row_off, sizeof(int64_t) * (size_t)n_rows)); dev.keep(d_row_off);
    GE_HIP(hipMalloc((void **)&d_row_max, sizeof(float) * (size_t)n_rows)); dev.keep(d_row_max);
    unsigned long long *d_ctr = nullptr;    // [0] pool_used, [1] queue, then status word
    GE_HIP(hipMalloc((void **)&d_ctr, 32)); dev.keep(d_ctr);
    p.row_n = d_row_n; p.row_off = d_row_off; p.row_max = d_row_max;
    p.pool_used = d_ctr; p.queue = d_ctr + 1; p.status = reinterpret_cast<int32_t *>(d_ctr + 2);
    unsigned long long *d_prof = nullptr;
    if (clk.on) { GE_HIP(hipMalloc((void **)&d_prof, 16)); dev.keep(d_prof); GE_HIP(hipMemset(d_prof, 0, 16)); }
    p.prof = d_prof;

    // ---- the passes ---------------------------------------------------------------------------------------------------------
    // Main launch: the LDS kernel (k_bca<true>: 1 024-slot tables per wavefront in LDS, seven wavefronts per CU) over every
    // bookmark into a pool whose size comes from a sample.  What it leaves: rows that did not fit the pool (size known) and
    // bookmarks that outgrew the LDS tables (status 4, size unknown).  Those are run by the global-memory kernel (k_bca<false>,
    // tables that grow on overflow): first without a pool, to learn their sizes, then -- together with the rows the pool had no
    // room for -- into a second pool of exactly the missing size.  cfg.table_slots > 0 (tests) or GE_BCA_TABLES=global: the
    // global-memory kernel does everything, as before round 3.
    const char *tables_env = std::getenv("GE_BCA_TABLES");
    const bool use_lds = cfg->table_slots == 0 && !(tables_env && std::strcmp(tables_env, "global") == 0);
    // global-memory tables: the TreeMap never holds more than 1/epsilon nodes (every node in it carries >= epsilon of at most 1.0
    // paint); the table holds every node touched by the forward+reverse passes.  Both grow on overflow.
    int64_t ac = (int64_t)std::min<double>((double)V, std::ceil(1.0 / cfg->epsilon) + 2.0) + 64;
    int64_t hc = 2048;
    while (hc < 4 * std::min<int64_t>(V, 256)) hc <<= 1;
    if (cfg->table_slots > 0) hc = std::max<int64_t>(64, cfg->table_slots);
    int64_t pool_cap = std::max<int64_t>((int64_t)n_rows * 128, 1 << 16);
    bool sampled = n_rows <= 16384;
    if (cfg->pool_entries > 0) { pool_cap = std::max<int64_t>(64, cfg->pool_entries); sampled = true; }
    int32_t *d_pJ = nullptr, *d_pJ2 = nullptr; float *d_pX = nullptr, *d_pX2 = nullptr;
    std::vector<int32_t> h_n((size_t)n_rows);
    std::vector<int64_t> h_off((size_t)n_rows);

    struct Work { char *mem = nullptr; int64_t hc = 0, ac = 0, stride = 0, n_waves = 0; bool lds = false; };
    auto free_work = [&](Work &w) { if (w.mem) { (void)hipFree(w.mem); w.mem = nullptr; } };
    auto make_work = [&](Work &w, bool lds) -> ge_status {
        free_work(w);
        w.lds = lds;
        if (lds) { w.hc = 2 * LDS_MAX; w.ac = LDS_AC; }                  // the workspace of the emission: 2 x 512 slots (a power of two; LDS slots index it)
        else { int hl = 0; while ((1ll << hl) < hc) ++hl; hc = 1ll << hl; w.hc = hc; w.ac = ac; }
        w.stride = (work_bytes(w.hc, w.ac) + 255) / 256 * 256;
        w.n_waves = std::min<int64_t>((int64_t)cus * (lds ? 10 : 16), n_rows);
        while (w.n_waves > 1 && w.n_waves * w.stride > (int64_t)6 << 30) w.n_waves /= 2;
        if (hipMalloc((void **)&w.mem, (size_t)(w.n_waves * w.stride)) != hipSuccess) {
            (void)hipGetLastError(); w.mem = nullptr;
            return ge::fail(GE_ERR_OOM, "device allocation failed for BCA work buffers (table %lld slots x %lld waves)", (long long)w.hc, (long long)w.n_waves);
        }
        return GE_OK;
    };
    // one launch over `n_jobs` bookmarks (all rows, or the list `jobs`); returns the status word, *used = entries taken from the pool
    auto run = [&](const Work &w, const int32_t *jobs, int32_t n_jobs, int32_t *oJ, float *oX, int64_t out_cap, unsigned long long used0,
                   int32_t *status, unsigned long long *used) -> ge_status {
        int hl = 0; while ((1ll << hl) < w.hc) ++hl;
        p.hc = (int32_t)w.hc; p.hc_log2 = hl; p.ac = (int32_t)w.ac; p.work = w.mem; p.work_stride = w.stride;
        p.outJ = oJ; p.outX = oX; p.out_cap = out_cap; p.redo = jobs; p.n_jobs = n_jobs;
        unsigned long long h_ctr[4] = {used0, 0, 0, 0};
        hipError_t e = hipMemcpy(d_ctr, h_ctr, 32, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            const dim3 g((unsigned)std::max<int64_t>(1, std::min<int64_t>(w.n_waves, n_jobs))), b(64);
            if (w.lds) hipLaunchKernelGGL(k_bca<true>, g, b, 0, 0, p); else hipLaunchKernelGGL(k_bca<false>, g, b, 0, 0, p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(h_ctr, d_ctr, 32, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return ge::fail(GE_ERR_HIP, "BCA kernel failed: %s", hipGetErrorString(e));
        *status = (int32_t)(h_ctr[2] & 0xFFFFFFFFull); *used = h_ctr[0];
        return GE_OK;
    };
    struct Guard { std::function<void()> f; ~Guard() { f(); } };
    Work wmain, wglob;
    Guard work_guard{[&] { free_work(wmain); free_work(wglob); }};
    auto grow = [&](int32_t status) { if (status == 1) hc *= 4; else ac = std::min<int64_t>(ac * 4, (int64_t)V + 64); };

    // the result object exists from here on: its host arrays are sized from the sample and their pages touched while the main launch runs
    std::unique_ptr<ge_coo> cown(new (std::nothrow) ge_coo());
    if (!cown) return ge::fail(GE_ERR_OOM, "host allocation failed");
    std::vector<std::thread> prefault;
    Guard prefault_guard{[&] { for (auto &th : prefault) if (th.joinable()) th.join(); }};      // (declared after cown: joins before the arrays go)
    int64_t est_total = 0;

    int32_t status = 0; unsigned long long used = 0;
    for (int attempt = 0;; ++attempt) {
        if (attempt > 12) return ge::fail(GE_ERR_OVERFLOW, "BCA work buffers kept overflowing (table %lld, active list %lld, pool %lld)", (long long)hc, (long long)ac, (long long)pool_cap);
        if ((st = make_work(wmain, use_lds)) != GE_OK) return st;
        if (!sampled) {
            // Large builds size the pool from a sample first: 2048 evenly spaced bookmarks are run without a pool (every row reports
            // its size, none is stored), the mean row size + 25 % decides.  < 1 % extra work instead of re-running what a wrong guess leaves out.
            const int32_t n_sample = 2048;
            std::vector<int32_t> rows((size_t)n_sample);
            for (int32_t k = 0; k < n_sample; ++k) rows[(size_t)k] = (int32_t)((int64_t)k * n_rows / n_sample);
            int32_t *d_sample = nullptr;
            if (hipMalloc((void **)&d_sample, sizeof(int32_t) * (size_t)n_sample) != hipSuccess) return ge::fail(GE_ERR_OOM, "device allocation failed for the BCA sample");
            dev.keep(d_sample);
            GE_HIP(hipMemcpy(d_sample, rows.data(), sizeof(int32_t) * (size_t)n_sample, hipMemcpyHostToDevice));
            if ((st = run(wmain, d_sample, n_sample, nullptr, nullptr, 0, 0, &status, &used)) != GE_OK) return st;
            if (!use_lds && (status == 1 || status == 2)) { grow(status); continue; }
            const double mean = (double)used / (double)n_sample;            // pool_used counted every sampled row that ran through
            pool_cap = std::max<int64_t>(1 << 16, (int64_t)(mean * 1.25 * (double)n_rows) + 65536);
            est_total = (int64_t)(mean * (double)n_rows);
            sampled = true;
        }
        if (d_pJ) { (void)hipFree(d_pJ); d_pJ = nullptr; }
        if (d_pX) { (void)hipFree(d_pX); d_pX = nullptr; }
        if (hipMalloc((void **)&d_pJ, sizeof(int32_t) * (size_t)pool_cap) != hipSuccess || hipMalloc((void **)&d_pX, sizeof(float) * (size_t)pool_cap) != hipSuccess) {
            (void)hipGetLastError();
            if (d_pJ) (void)hipFree(d_pJ);
            if (d_pX) (void)hipFree(d_pX);
            return ge::fail(GE_ERR_OOM, "device allocation failed for the BCA pool (%lld entries)", (long long)pool_cap);
        }
        if (est_total >= ((int64_t)1 << 22) && prefault.empty() && cown->capacity == 0 && !std::getenv("GE_BCA_NO_PREFAULT")
            && cown->reserve(est_total + est_total / 8 + 65536)) {
            const int n_touch = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
            ge_coo *const cc = cown.get();
            const int64_t upto = est_total + est_total / 50;
            for (int t = 0; t < n_touch; ++t) prefault.emplace_back([cc, upto, t, n_touch] { cc->touch(upto, t, n_touch); });
        }
        if ((st = run(wmain, nullptr, n_rows, d_pJ, d_pX, pool_cap, 0, &status, &used)) != GE_OK) { (void)hipFree(d_pJ); (void)hipFree(d_pX); return st; }
        if (!use_lds && (status == 1 || status == 2)) { grow(status); continue; }
        break;
    }
    dev.keep(d_pJ); dev.keep(d_pX);
    if (d_prof) {
        unsigned long long hp[2] = {0, 0};
        GE_HIP(hipMemcpy(hp, d_prof, 16, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[ge_bca_build] wavefront time in the passes %.1f %%, in the emission %.1f %% (%.3g / %.3g ticks)\n",
                     100.0 * (double)hp[0] / (double)std::max<unsigned long long>(hp[0] + hp[1], 1), 100.0 * (double)hp[1] / (double)std::max<unsigned long long>(hp[0] + hp[1], 1), (double)hp[0], (double)hp[1]);
    }
    if (status != 0) {
        // rows left over: no room in the pool (row_off -1, size in row_n) or too large for the LDS tables (row_off -2, size unknown)
        GE_HIP(hipMemcpy(h_n.data(), d_row_n, sizeof(int32_t) * (size_t)n_rows, hipMemcpyDeviceToHost));
        GE_HIP(hipMemcpy(h_off.data(), d_row_off, sizeof(int64_t) * (size_t)n_rows, hipMemcpyDeviceToHost));
        std::vector<int32_t> redo, big;
        for (int32_t r = 0; r < n_rows; ++r) if (h_off[(size_t)r] < 0) { redo.push_back(r); if (h_off[(size_t)r] == -2) big.push_back(r); }
        int32_t *d_redo = nullptr;
        GE_HIP(hipMalloc((void **)&d_redo, sizeof(int32_t) * std::max<size_t>(redo.size(), 1))); dev.keep(d_redo);
        const Work *wredo = &wmain;
        if (!big.empty()) {
            // a bookmark never yields more entries than the global-memory kernel's table holds nodes (hc / 2): that bounds the big rows,
            // whose sizes are unknown (should the tables have to grow, the launch below is repeated with the larger bound)
            if ((st = make_work(wglob, false)) != GE_OK) return st;
            wredo = &wglob;
        }
        auto row_bound = [&](int32_t r) -> int64_t { return h_off[(size_t)r] == -2 ? std::min<int64_t>(wglob.hc / 2, V) : (int64_t)h_n[(size_t)r]; };
        GE_HIP(hipMemcpy(d_redo, redo.data(), sizeof(int32_t) * redo.size(), hipMemcpyHostToDevice));
        // second-pool offsets are stored shifted by pool_cap so that one gather kernel can tell the pools apart
        int64_t cap2 = 0;
        for (int attempt = 0;; ++attempt) {
            if (attempt > 12) return ge::fail(GE_ERR_OVERFLOW, "BCA work buffers kept overflowing (table %lld, active list %lld)", (long long)hc, (long long)ac);
            int64_t need = 0;
            for (int32_t r : redo) need += row_bound(r);
            cap2 = need + 64;
            d_pJ2 = nullptr; d_pX2 = nullptr;
            bool good = hipMalloc((void **)&d_pJ2, sizeof(int32_t) * (size_t)cap2) == hipSuccess;
            if (d_pJ2) dev.keep(d_pJ2);
            good = good && hipMalloc((void **)&d_pX2, sizeof(float) * (size_t)cap2) == hipSuccess;
            if (d_pX2) dev.keep(d_pX2);
            if (!good) return ge::fail(GE_ERR_OOM, "device allocation failed for the second BCA pool (%lld entries)", (long long)cap2);
            if ((st = run(*wredo, d_redo, (int32_t)redo.size(), d_pJ2 - pool_cap, d_pX2 - pool_cap, pool_cap + cap2, (unsigned long long)pool_cap, &status, &used)) != GE_OK) return st;
            if (!wredo->lds && (status == 1 || status == 2)) { grow(status); if ((st = make_work(wglob, false)) != GE_OK) return st; wredo = &wglob; continue; }
            break;
        }
        if (status != 0) return ge::fail(GE_ERR_OVERFLOW, "BCA second pool overflowed (internal sizing error, status %d)", status);
    }
    clk.lap("upload + k_bca passes");

    for (auto &th : prefault) th.join();
    prefault.clear();
    ge_coo *c = cown.release();
    c->V = V;
    std::vector<float> h_max((size_t)n_rows);
    hipError_t e = hipMemcpy(h_n.data(), d_row_n, sizeof(int32_t) * (size_t)n_rows, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_max.data(), d_row_max, sizeof(float) * (size_t)n_rows, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { delete c; return ge::fail(GE_ERR_HIP, "copy back failed: %s", hipGetErrorString(e)); }
    c->row_ptr.assign((size_t)V + 1, 0);
    std::vector<int64_t> dst((size_t)n_rows);
    int64_t total = 0;
    for (int32_t v = 0; v < V; ++v) {
        c->row_ptr[(size_t)v] = total;
        if (v >= rb && v < re) { dst[(size_t)(v - rb)] = total; total += h_n[(size_t)(v - rb)]; }
    }
    c->row_ptr[(size_t)V] = total;
    c->nnz = total;
    // BookmarkColoring.setMax over bookmarks in completion (= ascending) order, J/bca/BookmarkColoring.java:97,162-164
    double mx = 0;
    for (int32_t r = 0; r < n_rows; ++r) mx = java_math_max(mx, (double)h_max[(size_t)r]);
    c->max = mx;
    clk.lap("row sizes, offsets, max");
    if (c->capacity < std::max<int64_t>(total, 1) && !c->reserve(total)) { delete c; return ge::fail(GE_ERR_OOM, "host allocation failed for the COO (%lld entries)", (long long)total); }
    clk.lap("host result arrays");
    if (total > 0) {
        int64_t *d_dst = nullptr; int32_t *d_I = nullptr, *d_J = nullptr; float *d_X = nullptr;
        bool good = hipMalloc((void **)&d_dst, sizeof(int64_t) * (size_t)n_rows) == hipSuccess; if (good) dev.keep(d_dst);
        good = good && hipMalloc((void **)&d_I, sizeof(int32_t) * (size_t)total) == hipSuccess; if (d_I) dev.keep(d_I);
        good = good && hipMalloc((void **)&d_J, sizeof(int32_t) * (size_t)total) == hipSuccess; if (d_J) dev.keep(d_J);
        good = good && hipMalloc((void **)&d_X, sizeof(float) * (size_t)total) == hipSuccess;   if (d_X) dev.keep(d_X);
        if (!good) { delete c; return ge::fail(GE_ERR_OOM, "device allocation failed for the COO (%lld entries)", (long long)total); }
        e = hipMemcpy(d_dst, dst.data(), sizeof(int64_t) * (size_t)n_rows, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)std::min<int64_t>((n_rows + 3) / 4, 65535)), dim3(256), 0, 0,
                               d_pJ, d_pX, d_pJ2, d_pX2, pool_cap, d_row_off, d_row_n, d_dst, n_rows, rb, d_I, d_J, d_X);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        clk.lap("k_gather_rows");
        // I is the row index repeated: host threads write it from the row sizes while J and X come over the bus (a third less to copy)
        int32_t *const hI = c->I.get();
        const int n_fill = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        std::vector<std::thread> fill;
        for (int t = 0; t < n_fill; ++t)
            fill.emplace_back([&, t] {
                for (int32_t r = (int32_t)((int64_t)n_rows * t / n_fill), r1 = (int32_t)((int64_t)n_rows * (t + 1) / n_fill); r < r1; ++r)
                    std::fill(hI + dst[(size_t)r], hI + dst[(size_t)r] + h_n[(size_t)r], rb + r);
            });
        // J and X into fresh pageable memory: a copy of that kind is staged through the runtime's pinned buffers by the calling thread
        // (and meets every page of the destination for the first time), so it is one host thread's work -- slices on several threads
        // run side by side (GE_BCA_COPY_THREADS, default 4; 1 = the two plain copies)
        int n_copy = 4;
        if (const char *ev = std::getenv("GE_BCA_COPY_THREADS")) n_copy = std::max(1, std::min(16, std::atoi(ev)));
        if (total < ((int64_t)1 << 22)) n_copy = 1;
        if (e == hipSuccess && n_copy == 1) {
            e = hipMemcpy(c->J.get(), d_J, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(c->X.get(), d_X, sizeof(float) * (size_t)total, hipMemcpyDeviceToHost);
        } else if (e == hipSuccess) {
            std::vector<hipError_t> err((size_t)n_copy, hipSuccess);
            std::vector<std::thread> copy;
            const int device = cfg->device;
            char *const hJ = reinterpret_cast<char *>(c->J.get()), *const hX = reinterpret_cast<char *>(c->X.get());
            const char *const dJb = reinterpret_cast<const char *>(d_J), *const dXb = reinterpret_cast<const char *>(d_X);
            for (int t = 0; t < n_copy; ++t)
                copy.emplace_back([&, t] {
                    hipError_t q = hipSetDevice(device);
                    // slice t of the 2 * total words [J | X]
                    const int64_t w0 = 2 * total * t / n_copy, w1 = 2 * total * (t + 1) / n_copy;
                    if (q == hipSuccess && w0 < total) q = hipMemcpy(hJ + 4 * w0, dJb + 4 * w0, (size_t)(4 * (std::min(w1, total) - w0)), hipMemcpyDeviceToHost);
                    if (q == hipSuccess && w1 > total) { const int64_t x0 = std::max(w0, total) - total; q = hipMemcpy(hX + 4 * x0, dXb + 4 * x0, (size_t)(4 * (w1 - total - x0)), hipMemcpyDeviceToHost); }
                    err[(size_t)t] = q;
                });
            for (auto &th : copy) th.join();
            for (hipError_t q : err) if (q != hipSuccess) e = q;
        }
        for (auto &th : fill) th.join();
        if (e != hipSuccess) { delete c; return ge::fail(GE_ERR_HIP, "COO gather failed: %s", hipGetErrorString(e)); }
        clk.lap("copy out");
    }
    *result = c;
    return GE_OK;
}

ge_status ge_coo_get(const ge_coo *c, int64_t *nnz, const int32_t **I, const int32_t **J, const float **X,
                     const int64_t **row_ptr, double *max) {
    if (!c) return ge::fail(GE_ERR_ARG, "null ge_coo handle");
    if (nnz) *nnz = c->nnz;
    if (I) *I = c->I.get();
    if (J) *J = c->J.get();
    if (X) *X = c->X.get();
    if (row_ptr) *row_ptr = c->row_ptr.data();
    if (max) *max = c->max;
    return GE_OK;
}

void ge_coo_destroy(ge_coo *c) { delete c; }

ge_status ge_bca_build(const ge_csr *out_nbrs, const ge_csr *in_nbrs, const ge_bca_cfg *cfg, ge_coo **result) {
    GE_GUARD(ge_bca_build_impl(out_nbrs, in_nbrs, cfg, result));
}

}  // extern "C"
