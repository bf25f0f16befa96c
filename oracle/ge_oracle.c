/*
 * ge_oracle.c -- CPU restatement of the Phaken/graph-embeddings hot path.
 * TEST INFRASTRUCTURE ONLY; see ge_oracle.h for the rules and for the
 * "parity unpinned" statement.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -pthread (oracle/Makefile).
 * float/double mixing follows the Java expressions token by token; every
 * narrowing `(float)` below corresponds to a Java implicit compound-assignment
 * narrowing or an explicit cast in the cited line.
 */
#include "ge_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* java.util.Random (JDK 8 source semantics; public algorithm)               */
/* ------------------------------------------------------------------------- */
#define JR_MULT 0x5DEECE66DULL
#define JR_ADD  0xBULL
#define JR_MASK ((1ULL << 48) - 1)

void geo_jrand_init(geo_jrand *r, int64_t seed) {
    r->seed = ((uint64_t)seed ^ JR_MULT) & JR_MASK;       /* initialScramble */
}

int32_t geo_jrand_next(geo_jrand *r, int bits) {
    r->seed = (r->seed * JR_MULT + JR_ADD) & JR_MASK;
    return (int32_t)(uint32_t)(r->seed >> (48 - bits));   /* (int)(seed >>> (48-bits)) */
}

int32_t geo_jrand_next_int(geo_jrand *r) { return geo_jrand_next(r, 32); }

int32_t geo_jrand_next_int_bound(geo_jrand *r, int32_t bound) {
    /* Random.nextInt(int bound); used via ExtendedRandom.uniform(int),
     * J/util/rnd/ExtendedRandom.java:53-56 */
    int32_t rr = geo_jrand_next(r, 31);
    int32_t m = bound - 1;
    if ((bound & m) == 0) {
        rr = (int32_t)(((int64_t)bound * (int64_t)rr) >> 31);
    } else {
        int32_t u = rr;
        for (;;) {
            rr = u % bound;
            /* u - r + m < 0 with Java int wrap-around */
            int32_t t = (int32_t)((uint32_t)u - (uint32_t)rr + (uint32_t)m);
            if (t >= 0) break;
            u = geo_jrand_next(r, 31);
        }
    }
    return rr;
}

float geo_jrand_next_float(geo_jrand *r) {
    return (float)geo_jrand_next(r, 24) / (float)(1 << 24);
}

void geo_jrand_shuffle(geo_jrand *r, int32_t *a, int32_t n) {
    /* J/util/rnd/ExtendedRandom.java:398-407 */
    for (int32_t i = 0; i < n; i++) {
        int32_t k = i + geo_jrand_next_int_bound(r, n - i);
        int32_t t = a[i]; a[i] = a[k]; a[k] = t;
    }
}

/* ------------------------------------------------------------------------- */
/* java.util.HashMap<Integer,Float> iteration-order emulation                 */
/* (BCV extends HashMap, J/bca/util/BCV.java:14).  Linked bins only:          */
/* treeified bins (>= 8 keys in one bin at capacity >= 64) are NOT emulated;  */
/* such bins keep list order here (documented deviation).          */
/* ------------------------------------------------------------------------- */
typedef struct { int32_t key; float val; int32_t next; } jnode;
typedef struct {
    jnode   *nodes;
    int32_t  n_nodes, cap_nodes;
    int32_t *head;
    int32_t  tabcap;     /* 0 = table == null */
    int32_t  threshold;
    int32_t  size;
} jhm;

static inline uint32_t jhm_hash(int32_t key) {
    uint32_t h = (uint32_t)key;            /* Integer.hashCode() == value */
    return h ^ (h >> 16);                  /* HashMap.hash() */
}

static void jhm_init(jhm *m) { memset(m, 0, sizeof(*m)); }
static void jhm_free(jhm *m) { free(m->nodes); free(m->head); jhm_init(m); }
static void jhm_clear(jhm *m) {
    m->n_nodes = 0; m->size = 0; m->tabcap = 0; m->threshold = 0;
}

static void jhm_resize(jhm *m) {
    int32_t oldcap = m->tabcap;
    int32_t newcap = oldcap ? oldcap * 2 : 16;
    int32_t *nh = (int32_t *)malloc(sizeof(int32_t) * (size_t)newcap);
    int32_t *nt = (int32_t *)malloc(sizeof(int32_t) * (size_t)newcap);
    for (int32_t b = 0; b < newcap; b++) { nh[b] = -1; nt[b] = -1; }
    /* HashMap.resize(): each old bin splits into lo/hi lists, relative order kept */
    for (int32_t b = 0; b < oldcap; b++) {
        int32_t e = m->head[b];
        while (e >= 0) {
            int32_t nx = m->nodes[e].next;
            int32_t nb = (int32_t)(jhm_hash(m->nodes[e].key) & (uint32_t)(newcap - 1));
            m->nodes[e].next = -1;
            if (nt[nb] < 0) nh[nb] = e; else m->nodes[nt[nb]].next = e;
            nt[nb] = e;
            e = nx;
        }
    }
    free(nt);
    free(m->head);
    m->head = nh;
    m->tabcap = newcap;
    m->threshold = (newcap / 4) * 3;       /* 0.75 * cap, exact for cap >= 16 */
}

static int32_t jhm_new_node(jhm *m, int32_t key, float val) {
    if (m->n_nodes == m->cap_nodes) {
        m->cap_nodes = m->cap_nodes ? m->cap_nodes * 2 : 64;
        m->nodes = (jnode *)realloc(m->nodes, sizeof(jnode) * (size_t)m->cap_nodes);
    }
    int32_t id = m->n_nodes++;
    m->nodes[id].key = key; m->nodes[id].val = val; m->nodes[id].next = -1;
    return id;
}

static int32_t jhm_find(const jhm *m, int32_t key) {
    if (!m->tabcap) return -1;
    int32_t e = m->head[jhm_hash(key) & (uint32_t)(m->tabcap - 1)];
    while (e >= 0) { if (m->nodes[e].key == key) return e; e = m->nodes[e].next; }
    return -1;
}

/* BCV.add(int,float): super.put(key, getOrDefault(key, 0f) + value)  J/bca/util/BCV.java:35-37.
 * HashMap.putVal: new keys appended at the bin TAIL; resize after ++size > threshold. */
static void jhm_bcv_add(jhm *m, int32_t key, float value) {
    int32_t e = jhm_find(m, key);
    if (e >= 0) { m->nodes[e].val = m->nodes[e].val + value; return; }
    if (!m->tabcap) jhm_resize(m);
    int32_t id = jhm_new_node(m, key, 0.0f + value);
    int32_t b = (int32_t)(jhm_hash(key) & (uint32_t)(m->tabcap - 1));
    if (m->head[b] < 0) m->head[b] = id;
    else { int32_t t = m->head[b]; while (m->nodes[t].next >= 0) t = m->nodes[t].next; m->nodes[t].next = id; }
    if (++m->size > m->threshold) jhm_resize(m);
}

/* HashMap.merge(key, value, Float::sum) (JDK 8): resize BEFORE the lookup when
 * size > threshold; a new key is linked at the bin HEAD; no resize afterwards.
 * Called from BCV.merge, J/bca/util/BCV.java:105-107. */
static void jhm_merge_sum(jhm *m, int32_t key, float value) {
    if (m->size > m->threshold || !m->tabcap) jhm_resize(m);
    int32_t e = jhm_find(m, key);
    if (e >= 0) { m->nodes[e].val = m->nodes[e].val + value; return; }  /* Float.sum(old, value) */
    int32_t id = jhm_new_node(m, key, value);
    int32_t b = (int32_t)(jhm_hash(key) & (uint32_t)(m->tabcap - 1));
    m->nodes[id].next = m->head[b];
    m->head[b] = id;
    ++m->size;
}

static void jhm_remove(jhm *m, int32_t key) {
    if (!m->tabcap) return;
    int32_t b = (int32_t)(jhm_hash(key) & (uint32_t)(m->tabcap - 1));
    int32_t e = m->head[b], prev = -1;
    while (e >= 0) {
        if (m->nodes[e].key == key) {
            if (prev < 0) m->head[b] = m->nodes[e].next; else m->nodes[prev].next = m->nodes[e].next;
            --m->size;
            return;
        }
        prev = e; e = m->nodes[e].next;
    }
}

/* iteration: bins ascending, list order */
#define JHM_FOREACH(m, e) \
    for (int32_t _b = 0; _b < (m)->tabcap; _b++) \
        for (int32_t e = (m)->head[_b]; e >= 0; e = (m)->nodes[e].next)

/* Float.compare */
static int jfloat_compare(float a, float b) {
    if (a < b) return -1;
    if (a > b) return 1;
    int32_t ia, ib;
    if (a != a) ia = 0x7fc00000; else memcpy(&ia, &a, 4);   /* floatToIntBits canonical NaN */
    if (b != b) ib = 0x7fc00000; else memcpy(&ib, &b, 4);
    return ia == ib ? 0 : (ia < ib ? -1 : 1);
}

/* BCV.max(): values().stream().max(Float::compareTo).orElse(1f)  J/bca/util/BCV.java:82-84 */
static float bcv_max(const jhm *m) {
    int have = 0; float best = 1.0f;
    JHM_FOREACH(m, e) {
        float v = m->nodes[e].val;
        if (!have) { best = v; have = 1; }
        else best = (jfloat_compare(best, v) >= 0) ? best : v;   /* BinaryOperator.maxBy */
    }
    return best;
}
/* BCV.min(): orElse(0f)  J/bca/util/BCV.java:75-77 */
static float bcv_min(const jhm *m) {
    int have = 0; float best = 0.0f;
    JHM_FOREACH(m, e) {
        float v = m->nodes[e].val;
        if (!have) { best = v; have = 1; }
        else best = (jfloat_compare(best, v) <= 0) ? best : v;   /* BinaryOperator.minBy */
    }
    return best;
}
/* BCV.sum(): reduce(Float::sum).orElse(0f), sequential stream = left fold in iteration order */
static float bcv_sum(const jhm *m) {
    int have = 0; float s = 0.0f;
    JHM_FOREACH(m, e) {
        if (!have) { s = m->nodes[e].val; have = 1; } else s = s + m->nodes[e].val;
    }
    return s;
}
/* BCV.toUnity  J/bca/util/BCV.java:64-70 */
static void bcv_to_unity(jhm *m, int32_t root) {
    jhm_remove(m, root);
    const float sum = bcv_sum(m);
    JHM_FOREACH(m, e) m->nodes[e].val = m->nodes[e].val / sum - 1e-6f;
}
/* BCV.toCounts + scale  J/bca/util/BCV.java:52-59,89-91 */
static void bcv_to_counts(jhm *m, int32_t root) {
    const float aMax = bcv_max(m), aMin = bcv_min(m);
    const float mx = 1000.0f, mn = 1.0f;
    JHM_FOREACH(m, e) m->nodes[e].val = (m->nodes[e].val / ((aMax - aMin) / (mx - mn))) + mn;
    jhm_remove(m, root);
}

/* ------------------------------------------------------------------------- */
/* BCA jobs                                                                   */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t V;
    const int64_t *out_ptr; const int32_t *out_idx; const float *out_w;
    const int64_t *in_ptr;  const int32_t *in_idx;  const float *in_w;
    double alpha, epsilon;
    /* TreeMap<Integer,PaintedNode> stand-in: dense paint + membership + min-heap of ids */
    double  *paint;
    uint8_t *in_tree;
    int32_t *heap; int32_t heap_n, heap_cap;
} bca_ctx;

static void heap_push(bca_ctx *c, int32_t id) {
    if (c->heap_n == c->heap_cap) {
        c->heap_cap = c->heap_cap ? c->heap_cap * 2 : 256;
        c->heap = (int32_t *)realloc(c->heap, sizeof(int32_t) * (size_t)c->heap_cap);
    }
    int32_t i = c->heap_n++;
    while (i > 0) {
        int32_t p = (i - 1) / 2;
        if (c->heap[p] <= id) break;
        c->heap[i] = c->heap[p]; i = p;
    }
    c->heap[i] = id;
}
static int32_t heap_pop(bca_ctx *c) {
    int32_t top = c->heap[0];
    int32_t last = c->heap[--c->heap_n];
    int32_t i = 0, n = c->heap_n;
    for (;;) {
        int32_t l = 2 * i + 1, r = l + 1, s = i; int32_t sv = last;
        if (l < n && c->heap[l] < sv) { s = l; sv = c->heap[l]; }
        if (r < n && c->heap[r] < sv) { s = r; sv = c->heap[r]; }
        if (s == i) break;
        c->heap[i] = c->heap[s]; i = s;
    }
    if (n > 0) c->heap[i] = last;
    return top;
}

/* nodeTree.containsKey / get().addPaint / put(new PaintedNode)
 * J/bca/jobs/DirectedWeighted.java:89-96 */
static inline void tree_add(bca_ctx *c, int32_t node, double p) {
    if (c->in_tree[node]) c->paint[node] += p;
    else { c->paint[node] = p; c->in_tree[node] = 1; heap_push(c, node); }
}

/* DirectedWeighted.doWork  J/bca/jobs/DirectedWeighted.java:31-101 */
static void dowork_directed(bca_ctx *c, int32_t bookmark, int reverse, jhm *bcv) {
    const double alpha = c->alpha, epsilon = c->epsilon;
    const int64_t *ptr = reverse ? c->in_ptr : c->out_ptr;
    const int32_t *idx = reverse ? c->in_idx : c->out_idx;
    const float   *w   = reverse ? c->in_w   : c->out_w;
    c->heap_n = 0;
    tree_add(c, bookmark, 1.0);                               /* :39 */
    while (c->heap_n > 0) {
        const int32_t focus = heap_pop(c);                    /* pollFirstEntry :48 */
        c->in_tree[focus] = 0;
        const double wet = c->paint[focus];
        jhm_bcv_add(bcv, focus, (float)(alpha * wet));        /* :53 */
        if (wet < epsilon) continue;                          /* :56 */
        const int64_t b = ptr[focus], e = ptr[focus + 1];
        if (e == b) continue;                                 /* :66 */
        double total = 0;
        for (int64_t k = b; k < e; k++) total += w[k];        /* :69-75 */
        if (total == 0) continue;                             /* :77 */
        for (int64_t k = b; k < e; k++) {
            const float weight = w[k];
            const double p = (1 - alpha) * wet * (weight / total);   /* :82 */
            if (p < epsilon) continue;                        /* :85 */
            tree_add(c, idx[k], p);
        }
    }
}

/* UndirectedWeighted.doWork  J/bca/jobs/UndirectedWeighted.java:31-114 */
static void dowork_undirected(bca_ctx *c, int32_t bookmark, jhm *bcv) {
    const double alpha = c->alpha, epsilon = c->epsilon;
    c->heap_n = 0;
    tree_add(c, bookmark, 1.0);
    while (c->heap_n > 0) {
        const int32_t focus = heap_pop(c);
        c->in_tree[focus] = 0;
        const double wet = c->paint[focus];
        jhm_bcv_add(bcv, focus, (float)(alpha * wet));        /* :55 */
        if (wet < epsilon) continue;                          /* :58 */
        double total = 0;
        for (int64_t k = c->out_ptr[focus]; k < c->out_ptr[focus + 1]; k++) total += c->out_w[k];  /* :63-67 */
        for (int64_t k = c->in_ptr[focus];  k < c->in_ptr[focus + 1];  k++) total += c->in_w[k];   /* :69-73 */
        /* no total == 0 guard in the undirected version */
        for (int64_t k = c->out_ptr[focus]; k < c->out_ptr[focus + 1]; k++) {       /* :75-93 */
            const float weight = c->out_w[k];
            const double p = (1 - alpha) * wet * (weight / total);
            if (p < epsilon) continue;
            tree_add(c, c->out_idx[k], p);
        }
        for (int64_t k = c->in_ptr[focus]; k < c->in_ptr[focus + 1]; k++) {         /* :95-112 */
            const float weight = c->in_w[k];
            const double p = (1 - alpha) * wet * (weight / total);
            if (p < epsilon) continue;
            tree_add(c, c->in_idx[k], p);
        }
    }
}

/* BCAJob.call (J/bca/util/BCAJob.java:31-36) + normalisation
 * (J/bca/BookmarkColoring.java:81-91) */
static void bca_job(bca_ctx *c, int32_t bookmark, int directed, int normalize, jhm *bcv, jhm *rev) {
    jhm_clear(bcv);
    if (directed) {
        dowork_directed(c, bookmark, 0, bcv);
        jhm_clear(rev);
        dowork_directed(c, bookmark, 1, rev);     /* DirectedWeighted passes reverse=true, :23 */
        JHM_FOREACH(rev, e) jhm_merge_sum(bcv, rev->nodes[e].key, rev->nodes[e].val);
    } else {
        dowork_undirected(c, bookmark, bcv);
    }
    if (normalize == GEO_NORM_UNITY) bcv_to_unity(bcv, bookmark);
    else if (normalize == GEO_NORM_COUNTS) bcv_to_counts(bcv, bookmark);
}

static int bca_ctx_init(bca_ctx *c, int32_t V,
                        const int64_t *out_ptr, const int32_t *out_idx, const float *out_w,
                        const int64_t *in_ptr, const int32_t *in_idx, const float *in_w,
                        double alpha, double epsilon) {
    memset(c, 0, sizeof(*c));
    c->V = V; c->out_ptr = out_ptr; c->out_idx = out_idx; c->out_w = out_w;
    c->in_ptr = in_ptr; c->in_idx = in_idx; c->in_w = in_w;
    c->alpha = alpha; c->epsilon = epsilon;
    c->paint = (double *)calloc((size_t)(V > 0 ? V : 1), sizeof(double));
    c->in_tree = (uint8_t *)calloc((size_t)(V > 0 ? V : 1), 1);
    return (c->paint && c->in_tree) ? 0 : -1;
}
static void bca_ctx_free(bca_ctx *c) { free(c->paint); free(c->in_tree); free(c->heap); }

/* Math.max(double,double) */
static double jmath_max(double a, double b) {
    if (a != a) return a;
    if (a == 0.0 && b == 0.0) { return signbit(a) ? b : a; }
    return (a >= b) ? a : b;
}

int geo_bca_build(int32_t V,
                  const int64_t *out_ptr, const int32_t *out_idx, const float *out_w,
                  const int64_t *in_ptr, const int32_t *in_idx, const float *in_w,
                  double alpha, double epsilon, int directed, int normalize,
                  geo_coo *res) {
    bca_ctx c;
    if (bca_ctx_init(&c, V, out_ptr, out_idx, out_w, in_ptr, in_idx, in_w, alpha, epsilon)) return -1;
    jhm bcv, rev; jhm_init(&bcv); jhm_init(&rev);
    memset(res, 0, sizeof(*res));
    res->V = V;
    int64_t cap = (int64_t)V * 8 + 16;
    res->I = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    res->J = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    res->X = (float *)malloc(sizeof(float) * (size_t)cap);
    res->row_ptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)V + 1));
    res->max = 0;
    /* bookmarks in ascending id = completion order with threads: 1 */
    for (int32_t b = 0; b < V; b++) {
        res->row_ptr[b] = res->nnz;
        bca_job(&c, b, directed, normalize, &bcv, &rev);
        res->max = jmath_max(res->max, (double)bcv_max(&bcv));          /* :97 */
        if (res->nnz + bcv.size > cap) {
            while (res->nnz + bcv.size > cap) cap *= 2;
            res->I = (int32_t *)realloc(res->I, sizeof(int32_t) * (size_t)cap);
            res->J = (int32_t *)realloc(res->J, sizeof(int32_t) * (size_t)cap);
            res->X = (float *)realloc(res->X, sizeof(float) * (size_t)cap);
        }
        JHM_FOREACH(&bcv, e) {                                          /* :99-103 */
            res->I[res->nnz] = b; res->J[res->nnz] = bcv.nodes[e].key; res->X[res->nnz] = bcv.nodes[e].val;
            res->nnz++;
        }
    }
    res->row_ptr[V] = res->nnz;
    jhm_free(&bcv); jhm_free(&rev); bca_ctx_free(&c);
    return 0;
}

void geo_coo_free(geo_coo *c) {
    free(c->I); free(c->J); free(c->X); free(c->row_ptr);
    memset(c, 0, sizeof(*c));
}

int64_t geo_bca_single(int32_t V,
                  const int64_t *out_ptr, const int32_t *out_idx, const float *out_w,
                  const int64_t *in_ptr, const int32_t *in_idx, const float *in_w,
                  double alpha, double epsilon, int directed, int normalize,
                  int32_t bookmark, int32_t *keys, float *vals, int64_t cap) {
    bca_ctx c;
    if (bca_ctx_init(&c, V, out_ptr, out_idx, out_w, in_ptr, in_idx, in_w, alpha, epsilon)) return -1;
    jhm bcv, rev; jhm_init(&bcv); jhm_init(&rev);
    bca_job(&c, bookmark, directed, normalize, &bcv, &rev);
    int64_t n = 0;
    JHM_FOREACH(&bcv, e) { if (n < cap) { keys[n] = bcv.nodes[e].key; vals[n] = bcv.nodes[e].val; } n++; }
    jhm_free(&bcv); jhm_free(&rev); bca_ctx_free(&c);
    return n;
}

/* ------------------------------------------------------------------------- */
/* GloVe / pGloVe cost + AdaGrad                                              */
/* ------------------------------------------------------------------------- */
struct geo_glove {
    int32_t V, D; int64_t N; int threads; int cost_kind; double xmax;
    int32_t *I, *J; float *X;
    float *focus, *context, *fbias, *cbias;
    float *gsf, *gsc, *gsfb, *gscb;      /* Adagrad.gradSq*  or  Adam/AMSGrad.M1* */
    float *m2f, *m2c, *m2fb, *m2cb;      /* Adam/AMSGrad.M2* (unused by Adagrad)  */
    int32_t *perm;
    geo_jrand rng;
    int opt_kind;
    int iteration;                       /* argument of createJob(id, iteration) for the next epoch */
};

static const float LEARNING_RATE = 0.05f;     /* J/opt/Optimizer.java:26 */

/* GloveCost / PGloveCost  J/opt/GloveCost.java:7-20, J/opt/PGloveCost.java:7-20 */
static inline float inner_cost(int kind, int32_t D, const float *foc, const float *ctx,
                               float fb, float cb, float Xij) {
    float ic = 0;
    for (int32_t d = 0; d < D; d++) ic += foc[d] * ctx[d];
    if (kind == GEO_COST_GLOVE)
        ic = (float)((double)ic + ((double)(fb + cb) - log((double)Xij)));
    else
        ic = (float)((double)ic + ((double)(fb + cb) - log((double)(Xij / (1 - Xij)))));
    return ic;
}
static inline float weighted_cost(int kind, double xmax, float ic, float Xij) {
    if (kind == GEO_COST_GLOVE)
        return ((double)Xij > xmax) ? ic : (float)pow((double)Xij / xmax, 0.75) * ic;
    return Xij * ic;
}

/* Adagrad.createJob body for one nonzero  J/opt/grad/Adagrad.java:51-93 */
static inline void adagrad_update(int kind, double xmax, int32_t D, int32_t bu, int32_t bv, float Xij,
                                  float *focus, float *context, float *fbias, float *cbias,
                                  float *gsf, float *gsc, float *gsfb, float *gscb, float *cost) {
    float *foc = focus + (int64_t)bu * D, *ctx = context + (int64_t)bv * D;
    float *g1s = gsf + (int64_t)bu * D,  *g2s = gsc + (int64_t)bv * D;
    const float ic = inner_cost(kind, D, foc, ctx, fbias[bu], cbias[bv], Xij);
    float wc = weighted_cost(kind, xmax, ic, Xij);
    *cost = (float)((double)*cost + 0.5 * wc * ic);                      /* :60 */
    for (int32_t d = 0; d < D; d++) {
        const float grad1 = wc * ctx[d];                                  /* :73 */
        const float grad2 = wc * foc[d];                                  /* :74 */
        foc[d] = (float)((double)foc[d] - grad1 / sqrt((double)g1s[d]) * LEARNING_RATE);  /* :76 */
        ctx[d] = (float)((double)ctx[d] - grad2 / sqrt((double)g2s[d]) * LEARNING_RATE);  /* :77 */
        g1s[d] += grad1 * grad1;                                          /* :79 */
        g2s[d] += grad2 * grad2;                                          /* :80 */
    }
    fbias[bu] = (float)((double)fbias[bu] - wc / sqrt((double)gsfb[bu]));  /* :88 (no lr) */
    cbias[bv] = (float)((double)cbias[bv] - wc / sqrt((double)gscb[bv]));  /* :89 */
    wc *= wc;                                                              /* :90 */
    gsfb[bu] += wc;                                                        /* :92 */
    gscb[bv] += wc;                                                        /* :93 */
}

/* FastMath.max(float,float) of commons-math 2.x */
static inline float fastmath_maxf(float a, float b) { return (a <= b) ? b : ((a + b) != (a + b) ? NAN : a); }

static const float ADAM_BETA1 = 0.9f, ADAM_BETA2 = 0.999f, ADAM_EPS = 1e-7f;    /* J/opt/grad/Adam.java:45-53 */

/* Adam.createJob's per-job constant  J/opt/grad/Adam.java:84 */
static double adam_correction(int iteration) {
    return LEARNING_RATE * sqrt(1 - pow(ADAM_BETA2, iteration + 1)) / (1 - pow(ADAM_BETA1, iteration + 1));
}

/* Adam.createJob body (J/opt/grad/Adam.java:86-146) and AMSGrad.createJob body (J/opt/grad/AMSGrad.java:100-160)
 * for one nonzero.  m1* live in the gs* arrays of the shared state, m2* in the m2* arrays. */
static inline void adam_update(int ams, double correction, int kind, double xmax, int32_t D, int32_t bu, int32_t bv, float Xij,
                               float *focus, float *context, float *fbias, float *cbias,
                               float *M1f, float *M1c, float *M1fb, float *M1cb,
                               float *M2f, float *M2c, float *M2fb, float *M2cb, float *cost) {
    float *foc = focus + (int64_t)bu * D, *ctx = context + (int64_t)bv * D;
    float *m1f = M1f + (int64_t)bu * D, *m1c = M1c + (int64_t)bv * D;
    float *m2f = M2f + (int64_t)bu * D, *m2c = M2c + (int64_t)bv * D;
    const float beta1 = ADAM_BETA1, beta2 = ADAM_BETA2, epsilon = ADAM_EPS;
    const float ic = inner_cost(kind, D, foc, ctx, fbias[bu], cbias[bv], Xij);
    const float wc = weighted_cost(kind, xmax, ic, Xij);
    *cost = (float)((double)*cost + 0.5 * wc * ic);
    for (int32_t d = 0; d < D; d++) {
        const float grad_u = wc * ctx[d];
        const float grad_v = wc * foc[d];
        const float m1 = beta1 * m1f[d] + (1 - beta1) * grad_u;
        const float m2 = beta1 * m1c[d] + (1 - beta1) * grad_v;
        float v1, v2;
        if (!ams) {
            v1 = beta2 * m2f[d] + (1 - beta2) * (grad_u * grad_u);
            v2 = beta2 * m2c[d] + (1 - beta2) * (grad_v * grad_v);
            foc[d] = (float)((double)foc[d] - correction * m1 / (sqrt((double)v1) + epsilon));      /* Adam.java:118 */
            ctx[d] = (float)((double)ctx[d] - correction * m2 / (sqrt((double)v2) + epsilon));
        } else {
            v1 = fastmath_maxf(m2f[d], beta2 * m2f[d] + (1 - beta2) * (grad_u * grad_u));          /* AMSGrad.java:129-130 */
            v2 = fastmath_maxf(m2c[d], beta2 * m2c[d] + (1 - beta2) * (grad_v * grad_v));
            foc[d] = (float)((double)foc[d] - LEARNING_RATE / (sqrt((double)v1) + epsilon) * m1);   /* AMSGrad.java:133 */
            ctx[d] = (float)((double)ctx[d] - LEARNING_RATE / (sqrt((double)v2) + epsilon) * m2);
        }
        m1f[d] = m1; m1c[d] = m2; m2f[d] = v1; m2c[d] = v2;
    }
    const float m1 = beta1 * M1fb[bu] + (1 - beta1) * wc;
    const float m2 = beta1 * M1cb[bv] + (1 - beta1) * wc;
    float v1, v2;
    if (!ams) {
        v1 = beta2 * M2fb[bu] + (1 - beta2) * (wc * wc);
        v2 = beta2 * M2cb[bv] + (1 - beta2) * (wc * wc);
        fbias[bu] = (float)((double)fbias[bu] - correction * m1 / (sqrt((double)v1) + epsilon));
        cbias[bv] = (float)((double)cbias[bv] - correction * m2 / (sqrt((double)v2) + epsilon));
    } else {
        v1 = fastmath_maxf(M2fb[bu], beta2 * M2fb[bu] + (1 - beta2) * (wc * wc));
        v2 = fastmath_maxf(M2cb[bv], beta2 * M2cb[bv] + (1 - beta2) * (wc * wc));
        fbias[bu] = (float)((double)fbias[bu] - LEARNING_RATE / (sqrt((double)v1) + epsilon) * m1);
        cbias[bv] = (float)((double)cbias[bv] - LEARNING_RATE / (sqrt((double)v2) + epsilon) * m2);
    }
    M1fb[bu] = m1; M1cb[bv] = m2; M2fb[bu] = v1; M2cb[bv] = v2;
}

float geo_opt_job(int opt_kind, int iteration, int32_t D, int64_t n, const int32_t *I, const int32_t *J, const float *X,
                  double xmax, int cost_kind,
                  float *focus, float *context, float *fbias, float *cbias,
                  float *s1f, float *s1c, float *s1fb, float *s1cb,
                  float *s2f, float *s2c, float *s2fb, float *s2cb) {
    float cost = 0;
    if (opt_kind == GEO_OPT_ADAGRAD) {
        for (int64_t k = 0; k < n; k++)
            adagrad_update(cost_kind, xmax, D, I[k], J[k], X[k], focus, context, fbias, cbias, s1f, s1c, s1fb, s1cb, &cost);
    } else {
        const double correction = adam_correction(iteration);
        for (int64_t k = 0; k < n; k++)
            adam_update(opt_kind == GEO_OPT_AMSGRAD, correction, cost_kind, xmax, D, I[k], J[k], X[k], focus, context, fbias, cbias,
                        s1f, s1c, s1fb, s1cb, s2f, s2c, s2fb, s2cb, &cost);
    }
    return cost;
}

float geo_adagrad_job(int32_t D, int64_t n, const int32_t *I, const int32_t *J, const float *X,
                      double xmax, int cost_kind,
                      float *focus, float *context, float *fbias, float *cbias,
                      float *gsf, float *gsc, float *gsfb, float *gscb) {
    float cost = 0;
    for (int64_t k = 0; k < n; k++)
        adagrad_update(cost_kind, xmax, D, I[k], J[k], X[k], focus, context, fbias, cbias,
                       gsf, gsc, gsfb, gscb, &cost);
    return cost;
}

geo_glove *geo_glove_create(int32_t V, int32_t D, int64_t N,
                            const int32_t *I, const int32_t *J, const float *X,
                            double xmax, int cost_kind, int64_t seed, int threads) {
    return geo_glove_create_opt(V, D, N, I, J, X, xmax, cost_kind, seed, threads, GEO_OPT_ADAGRAD);
}

geo_glove *geo_glove_create_opt(int32_t V, int32_t D, int64_t N,
                            const int32_t *I, const int32_t *J, const float *X,
                            double xmax, int cost_kind, int64_t seed, int threads, int opt_kind) {
    geo_glove *g = (geo_glove *)calloc(1, sizeof(*g));
    if (!g) return NULL;
    g->opt_kind = opt_kind;
    g->V = V; g->D = D; g->N = N; g->threads = threads < 1 ? 1 : threads;
    g->cost_kind = cost_kind; g->xmax = xmax;
    size_t vd = (size_t)V * (size_t)D;
    g->I = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
    g->J = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
    g->X = (float *)malloc(sizeof(float) * (size_t)(N ? N : 1));
    g->perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N ? N : 1));
    g->focus = (float *)malloc(sizeof(float) * (vd ? vd : 1));
    g->context = (float *)malloc(sizeof(float) * (vd ? vd : 1));
    g->gsf = (float *)malloc(sizeof(float) * (vd ? vd : 1));
    g->gsc = (float *)malloc(sizeof(float) * (vd ? vd : 1));
    g->fbias = (float *)malloc(sizeof(float) * (size_t)(V ? V : 1));
    g->cbias = (float *)malloc(sizeof(float) * (size_t)(V ? V : 1));
    g->gsfb = (float *)malloc(sizeof(float) * (size_t)(V ? V : 1));
    g->gscb = (float *)malloc(sizeof(float) * (size_t)(V ? V : 1));
    g->m2f = (float *)calloc((vd ? vd : 1), sizeof(float));
    g->m2c = (float *)calloc((vd ? vd : 1), sizeof(float));
    g->m2fb = (float *)calloc((size_t)(V ? V : 1), sizeof(float));
    g->m2cb = (float *)calloc((size_t)(V ? V : 1), sizeof(float));
    memcpy(g->I, I, sizeof(int32_t) * (size_t)N);
    memcpy(g->J, J, sizeof(int32_t) * (size_t)N);
    memcpy(g->X, X, sizeof(float) * (size_t)N);
    geo_jrand_init(&g->rng, seed);
    /* Optimizer ctor  J/opt/Optimizer.java:50-57 */
    for (int32_t i = 0; i < V; i++) {
        g->fbias[i] = (float)(geo_jrand_next_float(&g->rng) - 0.5) / D;
        g->cbias[i] = (float)(geo_jrand_next_float(&g->rng) - 0.5) / D;
        for (int32_t d = 0; d < D; d++) {
            g->focus[(size_t)i * D + d]   = (float)(geo_jrand_next_float(&g->rng) - 0.5) / D;
            g->context[(size_t)i * D + d] = (float)(geo_jrand_next_float(&g->rng) - 0.5) / D;
        }
    }
    /* Adagrad ctor  J/opt/grad/Adagrad.java:27-33: gradSq = 1;  Adam / AMSGrad ctors: moments = 0 (new float[]) */
    const float init1 = opt_kind == GEO_OPT_ADAGRAD ? 1.0f : 0.0f;
    for (size_t k = 0; k < vd; k++) g->gsf[k] = g->gsc[k] = init1;
    for (int32_t i = 0; i < V; i++) g->gsfb[i] = g->gscb[i] = init1;
    /* Permutation ctor  J/util/rnd/Permutation.java:11-15 */
    for (int64_t k = 0; k < N; k++) g->perm[k] = (int32_t)k;
    return g;
}

void geo_glove_destroy(geo_glove *g) {
    if (!g) return;
    free(g->I); free(g->J); free(g->X); free(g->perm);
    free(g->focus); free(g->context); free(g->gsf); free(g->gsc);
    free(g->fbias); free(g->cbias); free(g->gsfb); free(g->gscb);
    free(g->m2f); free(g->m2c); free(g->m2fb); free(g->m2cb);
    free(g);
}

typedef struct { geo_glove *g; int id; float cost; } job_arg;

/* Adagrad.createJob(id, iteration)  J/opt/grad/Adagrad.java:42-98 */
static void *run_job(void *p) {
    job_arg *a = (job_arg *)p;
    geo_glove *g = a->g;
    const int T = g->threads;
    const int64_t per = g->N / T;
    const int64_t offset = per * a->id;                                   /* :47 */
    const int64_t lines = (a->id == T - 1) ? per + g->N % T : per;        /* Optimizer.java:59-63 */
    float cost = 0;
    const double correction = adam_correction(g->iteration);
    for (int64_t i = 0; i < lines; i++) {
        const int32_t p2 = g->perm[i + offset];                           /* BookmarkColoring.java:127-137 */
        if (g->opt_kind == GEO_OPT_ADAGRAD)
            adagrad_update(g->cost_kind, g->xmax, g->D, g->I[p2], g->J[p2], g->X[p2],
                           g->focus, g->context, g->fbias, g->cbias, g->gsf, g->gsc, g->gsfb, g->gscb, &cost);
        else
            adam_update(g->opt_kind == GEO_OPT_AMSGRAD, correction, g->cost_kind, g->xmax, g->D, g->I[p2], g->J[p2], g->X[p2],
                        g->focus, g->context, g->fbias, g->cbias, g->gsf, g->gsc, g->gsfb, g->gscb,
                        g->m2f, g->m2c, g->m2fb, g->m2cb, &cost);
    }
    a->cost = cost;
    return NULL;
}

double geo_glove_epoch_noshuffle(geo_glove *g, int race) {
    const int T = g->threads;
    job_arg *args = (job_arg *)calloc((size_t)T, sizeof(job_arg));
    double local = 0;
    if (race && T > 1) {
        pthread_t *th = (pthread_t *)calloc((size_t)T, sizeof(pthread_t));
        for (int t = 0; t < T; t++) { args[t].g = g; args[t].id = t; pthread_create(&th[t], NULL, run_job, &args[t]); }
        for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); local += args[t].cost; }
        free(th);
    } else {
        for (int t = 0; t < T; t++) { args[t].g = g; args[t].id = t; run_job(&args[t]); local += args[t].cost; }
    }
    free(args);
    g->iteration++;
    return g->N ? local / (double)g->N : local / 0.0;                     /* Optimizer.java:96 */
}

double geo_glove_epoch(geo_glove *g, int race) {
    geo_jrand_shuffle(&g->rng, g->perm, (int32_t)g->N);                   /* Optimizer.java:79 */
    return geo_glove_epoch_noshuffle(g, race);
}

int geo_glove_optimize(geo_glove *g, int maxiter, double tolerance,
                       double *history, double *final_cost, int race) {
    /* Optimizer.optimize  J/opt/Optimizer.java:66-120 */
    double prev = 0, fin = 0; int it;
    for (it = 0; it < maxiter; it++) {
        double local = geo_glove_epoch(g, race);
        if (history) history[it] = local;
        double diff = fabs(prev - local);
        prev = local;
        if (diff <= tolerance) { fin = local; it++; break; }
    }
    if (final_cost) *final_cost = fin;
    return it;
}

void geo_glove_extract(const geo_glove *g, double *out) {
    /* J/opt/Optimizer.java:129-140: float add, float /2, widened on store */
    size_t vd = (size_t)g->V * (size_t)g->D;
    for (size_t k = 0; k < vd; k++) out[k] = (g->focus[k] + g->context[k]) / 2;
}

float   *geo_glove_focus(geo_glove *g)       { return g->focus; }
float   *geo_glove_context(geo_glove *g)     { return g->context; }
float   *geo_glove_fbias(geo_glove *g)       { return g->fbias; }
float   *geo_glove_cbias(geo_glove *g)       { return g->cbias; }
float   *geo_glove_gsq_focus(geo_glove *g)   { return g->gsf; }
float   *geo_glove_gsq_context(geo_glove *g) { return g->gsc; }
float   *geo_glove_gsq_fbias(geo_glove *g)   { return g->gsfb; }
float   *geo_glove_gsq_cbias(geo_glove *g)   { return g->gscb; }
float   *geo_glove_m2_focus(geo_glove *g)    { return g->m2f; }
float   *geo_glove_m2_context(geo_glove *g)  { return g->m2c; }
float   *geo_glove_m2_fbias(geo_glove *g)    { return g->m2fb; }
float   *geo_glove_m2_cbias(geo_glove *g)    { return g->m2cb; }
void     geo_glove_set_iteration(geo_glove *g, int it) { g->iteration = it; }
int32_t *geo_glove_perm(geo_glove *g)        { return g->perm; }
uint64_t geo_glove_rng_state(const geo_glove *g) { return g->rng.seed; }

/* ------------------------------------------------------------------------- */
/* String.format("%11.6E")                                                    */
/* ------------------------------------------------------------------------- */
int geo_format_11_6E(double v, char *buf, int buflen) {
    if (v != v) return snprintf(buf, (size_t)buflen, "%11s", "NaN");
    if (isinf(v)) return snprintf(buf, (size_t)buflen, "%11s", v > 0 ? "Infinity" : "-Infinity");
    char tmp[64];
    int neg = signbit(v) ? 1 : 0;
    double a = fabs(v);
    /* shortest decimal that round-trips (what FloatingDecimal hands the Formatter) */
    int prec;
    for (prec = 0; prec <= 17; prec++) {
        snprintf(tmp, sizeof tmp, "%.*e", prec, a);
        if (strtod(tmp, NULL) == a) break;
    }
    /* tmp = d.ddddde[+-]XX ; collect digits + exponent */
    char digits[32]; int nd = 0; int ex = 0;
    char *ep = strchr(tmp, 'e');
    ex = atoi(ep + 1);
    for (char *p = tmp; p < ep; p++) if (*p >= '0' && *p <= '9') digits[nd++] = *p;
    /* round HALF_UP to 7 significant digits */
    int want = 7;
    int d7[8];
    for (int k = 0; k < want; k++) d7[k] = (k < nd) ? digits[k] - '0' : 0;
    if (nd > want && digits[want] >= '5') {
        int k = want - 1;
        while (k >= 0) { if (++d7[k] < 10) break; d7[k] = 0; k--; }
        if (k < 0) { for (int q = want - 1; q > 0; q--) d7[q] = d7[q - 1]; d7[0] = 1; ex += 1; }
    }
    if (a == 0.0) ex = 0;
    char body[40];
    int n = snprintf(body, sizeof body, "%s%d.%d%d%d%d%d%dE%c%02d", neg ? "-" : "",
                     d7[0], d7[1], d7[2], d7[3], d7[4], d7[5], d7[6],
                     ex < 0 ? '-' : '+', ex < 0 ? -ex : ex);
    (void)n;
    return snprintf(buf, (size_t)buflen, "%11s", body);
}
