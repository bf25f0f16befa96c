/*
 * ge_oracle.h -- CPU restatement of the Phaken/graph-embeddings hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference is Java 8 (no JDK in the build image) and
 * ships no tests, golden vectors or fixtures.
 * This restatement is pinned only by (1) public java.util.Random known
 * answers and (2) known-answer tests derived by hand from the Java source
 * (tests/test_oracle_kat.py).  Golden files under tests/golden/ are generated
 * by THIS restatement and are labelled as such.
 *
 * Citation convention: J/ = src/main/java/org/uu/nl/embedding/ of the reference.
 * Compile with -O2 -ffp-contract=off (Java has no FMA contraction).
 */
#ifndef GE_ORACLE_H
#define GE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- java.util.Random (J/util/rnd/ExtendedRandom.java:27-35 extends it) ---- */
typedef struct { uint64_t seed; } geo_jrand;
void    geo_jrand_init(geo_jrand *r, int64_t seed);
int32_t geo_jrand_next(geo_jrand *r, int bits);
int32_t geo_jrand_next_int(geo_jrand *r);                 /* nextInt()        */
int32_t geo_jrand_next_int_bound(geo_jrand *r, int32_t bound); /* nextInt(n)  */
float   geo_jrand_next_float(geo_jrand *r);               /* nextFloat()      */
/* ExtendedRandom.shuffle(int[]) J/util/rnd/ExtendedRandom.java:398-407 */
void    geo_jrand_shuffle(geo_jrand *r, int32_t *a, int32_t n);

/* ---- BCA co-occurrence builder (J/bca/BookmarkColoring.java:32-120) ---- */
enum { GEO_NORM_NONE = 0, GEO_NORM_UNITY = 1, GEO_NORM_COUNTS = 2 };

typedef struct {
    int64_t  nnz;
    int32_t *I;       /* root (bookmark) of each entry           */
    int32_t *J;       /* painted node                             */
    float   *X;       /* paint value                              */
    int64_t *row_ptr; /* V+1 offsets: entries of bookmark b       */
    double   max;     /* BookmarkColoring.max (J/bca/BookmarkColoring.java:162-164) */
    int32_t  V;
} geo_coo;

/* out_*: CSR of out-neighbours (unique per row, order = given order),
 * in_*:  CSC (in-neighbours).  Weights are the fp32 edge weights read by
 * NumericalProperty.getValueAsFloat (J/bca/jobs/DirectedWeighted.java:73,81). */
int  geo_bca_build(int32_t V,
                   const int64_t *out_ptr, const int32_t *out_idx, const float *out_w,
                   const int64_t *in_ptr,  const int32_t *in_idx,  const float *in_w,
                   double alpha, double epsilon, int directed, int normalize,
                   geo_coo *result);
void geo_coo_free(geo_coo *c);

/* One bookmark, one direction (DirectedWeighted.doWork / UndirectedWeighted.doWork).
 * Returns number of entries written to keys/vals in HashMap iteration order
 * (cap = capacity of keys/vals). For tests. */
int64_t geo_bca_single(int32_t V,
                   const int64_t *out_ptr, const int32_t *out_idx, const float *out_w,
                   const int64_t *in_ptr,  const int32_t *in_idx,  const float *in_w,
                   double alpha, double epsilon, int directed, int normalize,
                   int32_t bookmark, int32_t *keys, float *vals, int64_t cap);

/* java.util.HashMap<Integer,Float> order emulation on its own (KATs): op 0 = BCV.add (put), 1 = merge, 2 = remove. */
int64_t geo_hashmap_replay(int64_t n_ops, const int32_t *ops, const int32_t *keys,
                           int32_t *out_keys, int64_t cap, int32_t *table_len, int32_t *tree_bins);

/* ---- GloVe / pGloVe AdaGrad trainer (J/opt/Optimizer.java, J/opt/grad/Adagrad.java) ---- */
enum { GEO_COST_GLOVE = 0, GEO_COST_PGLOVE = 1 };
/* Configuration.OptimizationMethod: Adagrad (J/opt/grad/Adagrad.java), Adam (Adam.java), AMSGrad (AMSGrad.java) */
enum { GEO_OPT_ADAGRAD = 0, GEO_OPT_ADAM = 1, GEO_OPT_AMSGRAD = 2 };

typedef struct geo_glove geo_glove;

/* Mirrors `new Adagrad(coMatrix, config, costFunction)` (J/Main.java:125):
 * Optimizer ctor seeded init (J/opt/Optimizer.java:50-57) drawing from
 * java.util.Random(seed), gradSq = 1 (J/opt/grad/Adagrad.java:27-33),
 * identity permutation (J/util/rnd/Permutation.java:11-15).
 * I/J/X are in pre-shuffle (matrix) order; threads = Configuration.getThreads(). */
geo_glove *geo_glove_create(int32_t V, int32_t D, int64_t N,
                            const int32_t *I, const int32_t *J, const float *X,
                            double xmax, int cost_kind, int64_t seed, int threads);
/* Same for `new Adam(...)` / `new AMSGrad(...)` (J/Main.java:126-129): moments start at 0.  In the shared
 * state the first-moment tables M1* take the place of gradSq*, the second moments are the m2_* tables. */
geo_glove *geo_glove_create_opt(int32_t V, int32_t D, int64_t N,
                            const int32_t *I, const int32_t *J, const float *X,
                            double xmax, int cost_kind, int64_t seed, int threads, int opt_kind);
void   geo_glove_destroy(geo_glove *g);
/* One iteration of Optimizer.optimize()'s loop body (J/opt/Optimizer.java:79-96):
 * shuffle (cumulative Fisher-Yates), T jobs, returns localCost = sum/N.
 * race=0: jobs run one after another in id order (bit-reproducible; equals Java for T=1).
 * race=1: jobs run as T racing pthreads (Hogwild, as Java does for T>1; CPU baseline). */
double geo_glove_epoch(geo_glove *g, int race);
/* Same without the shuffle (used to time the pure AdaGrad loop / custom orders). */
double geo_glove_epoch_noshuffle(geo_glove *g, int race);
/* Optimizer.optimize() whole loop; returns number of epochs run, fills history. */
int    geo_glove_optimize(geo_glove *g, int maxiter, double tolerance,
                          double *history, double *final_cost, int race);
/* Optimizer.extractResult (J/opt/Optimizer.java:129-140): (focus+context)/2 widened. */
void   geo_glove_extract(const geo_glove *g, double *out);
/* raw state access for tests */
float   *geo_glove_focus(geo_glove *g);
float   *geo_glove_context(geo_glove *g);
float   *geo_glove_fbias(geo_glove *g);
float   *geo_glove_cbias(geo_glove *g);
float   *geo_glove_gsq_focus(geo_glove *g);
float   *geo_glove_gsq_context(geo_glove *g);
float   *geo_glove_gsq_fbias(geo_glove *g);
float   *geo_glove_gsq_cbias(geo_glove *g);
float   *geo_glove_m2_focus(geo_glove *g);
float   *geo_glove_m2_context(geo_glove *g);
float   *geo_glove_m2_fbias(geo_glove *g);
float   *geo_glove_m2_cbias(geo_glove *g);
void     geo_glove_set_iteration(geo_glove *g, int iteration);
int32_t *geo_glove_perm(geo_glove *g);
uint64_t geo_glove_rng_state(const geo_glove *g);

/* Applies ONE batch of updates in the given order on caller-owned state
 * (no RNG, no permutation) -- the bare Adagrad.createJob loop body
 * (J/opt/grad/Adagrad.java:49-95).  Returns the job's fp32 cost. */
float geo_adagrad_job(int32_t D, int64_t n,
                      const int32_t *I, const int32_t *J, const float *X,
                      double xmax, int cost_kind,
                      float *focus, float *context, float *fbias, float *cbias,
                      float *gsf, float *gsc, float *gsfb, float *gscb);

/* One job of any optimiser over (I,J,X) in the given order on caller-owned state; `iteration` is createJob's
 * argument (Adam's bias correction).  s1* = gradSq (Adagrad) or M1 (Adam/AMSGrad); s2* = M2 (ignored by Adagrad). */
float geo_opt_job(int opt_kind, int iteration, int32_t D, int64_t n,
                  const int32_t *I, const int32_t *J, const float *X, double xmax, int cost_kind,
                  float *focus, float *context, float *fbias, float *cbias,
                  float *s1f, float *s1c, float *s1fb, float *s1cb,
                  float *s2f, float *s2c, float *s2fb, float *s2cb);

/* String.format("%11.6E", v) (J/util/write/EmbeddingTextWriter.java:134):
 * Java rounds HALF_UP on the shortest-repr decimal of the double. */
int geo_format_11_6E(double v, char *buf, int buflen);

/* ---- literal-similarity edges (oracle/ge_oracle_sim.c) ----
 * Configuration.SimilarityMethod ordinals (J/util/config/Configuration.java:27-29) and SimilarityGroup.Time (:184-186). */
enum { GEO_SIM_NGRAM_COSINE = 0, GEO_SIM_NGRAM_JACCARD = 1, GEO_SIM_TOKEN_COSINE = 2, GEO_SIM_TOKEN_JACCARD = 3,
       GEO_SIM_JAROWINKLER = 4, GEO_SIM_LEVENSHTEIN = 5, GEO_SIM_NUMERIC = 6,
       GEO_SIM_DATE_DAYS = 7, GEO_SIM_DATE_MONTHS = 8, GEO_SIM_DATE_YEARS = 9 };
enum { GEO_TIME_BACKWARDS = 0, GEO_TIME_FORWARDS = 1, GEO_TIME_BIDIRECTIONAL = 2 };
typedef struct {
    int32_t method;        /* GEO_SIM_*                                               */
    double  threshold;     /* SimilarityGroup.getThreshold                            */
    int32_t ngram;         /* getNgram(): 0 in the YAML means 3                       */
    double  smooth;        /* getSmooth(): 0 means 1 (Numeric's `alpha`)              */
    double  distance;      /* getDistance()                                           */
    int32_t time;          /* GEO_TIME_*                                              */
    const char *pattern;   /* getPattern(): NULL / "iso" = BASIC_ISO_DATE             */
} geo_sim_cfg;
/* metric.similarity(s1, s2) on UTF-16 code units; *threw = 1 where the Java call throws (Numeric.java:35-36) */
double  geo_sim_pair(const geo_sim_cfg *c, const uint16_t *s1, int32_t n1, const uint16_t *s2, int32_t n2, int *threw);
double  geo_sim_jarowinkler(const uint16_t *s1, int32_t n1, const uint16_t *s2, int32_t n2);
int32_t geo_sim_levenshtein_distance(const uint16_t *s1, int32_t n1, const uint16_t *s2, int32_t n2);
int     geo_sim_pattern_supported(const char *pattern);
/* CompareJob loop of one CompareGroup, jobs in order (threads: 1); returns the number of (source i, target j, sim) */
int64_t geo_compare_group(const geo_sim_cfg *c, const int64_t *offset, const uint16_t *units,
                          const int32_t *src, const int32_t *src_vert, int32_t n_src,
                          const int32_t *tgt, const int32_t *tgt_vert, int32_t n_tgt, int upper_triangle,
                          int32_t *out_src, int32_t *out_tgt, float *out_sim, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
