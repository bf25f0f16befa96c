/*
 * geglove.h -- C ABI of libgeglove.so, the MI355X (gfx950) implementation of the
 * Phaken/graph-embeddings hot path:  BCA co-occurrence builder  ->  GloVe/pGloVe
 * AdaGrad loop over the nonzero (i, j, X_ij) entries.
 *
 * The reference is pure Java with no native boundary; this header IS the
 * boundary a maintainer would bind over JNI (INTEGRATION.md shows the Java side).  Each
 * entry point cites the reference interface it replaces (J/ = src/main/java/org/uu/nl/embedding/).
 *
 * Conventions: plain C, no C++ types, no exceptions; every call returns ge_status
 * (0 = ok, <0 = error) and leaves a message readable through ge_last_error() (thread local).
 * The caller owns every host buffer passed in or filled; the library never keeps a host
 * pointer after the call returns.  Handles are opaque and freed only by *_destroy.
 * Calls on one handle are blocking and not re-entrant; distinct handles are independent.
 * The library never calls exit()/abort().  There is NO CPU fallback: every entry point
 * that computes needs a gfx950 device and fails with GE_ERR_HIP otherwise.
 */
#ifndef GEGLOVE_H
#define GEGLOVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t ge_status;
enum {
    GE_OK           =  0,
    GE_ERR_ARG      = -1,   /* invalid argument (the Java path throws IllegalArgumentException / InvalidConfigurationException) */
    GE_ERR_OOM      = -2,   /* host or device allocation failed */
    GE_ERR_HIP      = -3,   /* HIP runtime error / no device */
    GE_ERR_STATE    = -4,   /* call not valid in the handle's current state */
    GE_ERR_OVERFLOW = -5    /* a per-bookmark work buffer overflowed (BCA frontier / row capacity) */
};

/* Configuration.EmbeddingMethod (J/util/config/Configuration.java:19-21): picks
 * GloveCost (J/opt/GloveCost.java:5-21) or PGloveCost (J/opt/PGloveCost.java:5-21). */
enum { GE_COST_GLOVE = 0, GE_COST_PGLOVE = 1 };
/* Configuration.OptimizationMethod (J/util/config/Configuration.java:23-25); Main.createOptimizer
 * (J/Main.java:121-130): Adagrad (J/opt/grad/Adagrad.java), Adam (J/opt/grad/Adam.java:75-147),
 * AMSGrad (J/opt/grad/AMSGrad.java:91-162).  beta1 = 0.9f, beta2 = 0.999f, epsilon = 1e-7f as hard coded there. */
enum { GE_OPT_ADAGRAD = 0, GE_OPT_ADAM = 1, GE_OPT_AMSGRAD = 2 };
/* Configuration.BCANormalization (J/util/config/Configuration.java:31-33). */
enum { GE_NORM_NONE = 0, GE_NORM_UNITY = 1, GE_NORM_COUNTS = 2 };

/* How the update loop is scheduled on the device.
 * DETERMINISTIC: the exact semantics of the Java loop with `threads: T` where the T jobs of
 *   Adagrad.createJob (J/opt/grad/Adagrad.java:42-98) run one after another in id order;
 *   for T = 1 this IS the Java result, bit for bit (sequential fp32 dot, fp64 sqrt/div as in
 *   :76-77, :88-89).  One wavefront; a parity / reproducibility mode, not a fast one.
 * HOGWILD: the production mode.  Thousands of lane groups update the shared tables without
 *   locks, exactly as the Java worker threads do (J/opt/Optimizer.java:27-28 are plain arrays),
 *   fp32 arithmetic, wave-reduced dot. */
enum { GE_MODE_HOGWILD = 0, GE_MODE_DETERMINISTIC = 1 };

/* Order in which an epoch visits the nonzeros.
 * JAVA:   CoOccurrenceMatrix.shuffle() = cumulative forward Fisher-Yates on one permutation,
 *         drawn from the SAME java.util.Random stream that initialised the parameters
 *         (J/opt/Optimizer.java:79; J/util/rnd/Permutation.java:21; ExtendedRandom.java:398-407).
 * DEVICE: (HOGWILD) the matrix is cut once into chunks of 128 nonzeros -- hub columns column-major, the
 *         rest row-major as BookmarkColoring emits it -- and every epoch visits the chunks in a fresh
 *         keyed-bijection order evaluated inside the kernel (no permutation array, no host work).  One
 *         side of each update then stays in registers for a whole run, which halves the HBM traffic.
 *         Not the Java order; measured statistically equivalent.  (DETERMINISTIC: a keyed
 *         bijection of the nonzeros themselves.)
 * NONE:   matrix order (no shuffle). */
enum { GE_SHUFFLE_JAVA = 0, GE_SHUFFLE_DEVICE = 1, GE_SHUFFLE_NONE = 2 };

/* Lock-free does not mean lossy on a GPU: ~10^4 lane groups are in flight (the JVM has <= #cores
 * threads), so a hub column j that appears in a large share of the nonzeros would lose most of its
 * concurrent plain read-modify-write updates.  Nonzeros of such "hot" columns therefore update the
 * context side with float atomic adds (no update is lost; gradients may be a few microseconds
 * stale, as under Hogwild) and read it with agent-coherent loads.
 * AUTO: column j is hot when  count(j) * groups_in_flight >= 0.25 * N  (library default).
 * NONE: no column is treated as a hub: context rows are always read, updated and stored plainly (the literal Java race on that
 *   side; focus rows are still never overwritten, see GE_LAYOUT_*).  ALL: every column (tests). */
enum { GE_HOT_AUTO = 0, GE_HOT_NONE = 1, GE_HOT_ALL = 2 };

/* Storage type of the embedding rows (focus, context).  BF16 is BASELINE config C5: bf16 rows, fp32 AdaGrad
 * accumulators and biases, fp32 arithmetic; rows are narrowed with stochastic rounding, hub context rows keep an
 * fp32 master copy.  HOGWILD + GE_SHUFFLE_DEVICE + ADAGRAD + dim % 4 == 0 only (the reference is fp32 throughout,
 * so there is no bit-exact mode for it); every API still speaks fp32 (get/set/extract convert). */
enum { GE_DTYPE_F32 = 0, GE_DTYPE_BF16 = 1 };

/* Parameter tables a caller can read or write (tests, checkpointing, multi-GPU sync). */
enum {
    GE_STATE_FOCUS = 0,        /* float[V*D]  Optimizer.focus         (J/opt/Optimizer.java:27) */
    GE_STATE_CONTEXT = 1,      /* float[V*D]  Optimizer.context                                  */
    GE_STATE_FBIAS = 2,        /* float[V]    Optimizer.fBias         (J/opt/Optimizer.java:28) */
    GE_STATE_CBIAS = 3,        /* float[V]    Optimizer.cBias                                    */
    GE_STATE_GSQ_FOCUS = 4,    /* float[V*D]  Adagrad.gradSqFocus     (J/opt/grad/Adagrad.java:16) | Adam/AMSGrad.M1focus   */
    GE_STATE_GSQ_CONTEXT = 5,  /* float[V*D]  Adagrad.gradSqContext                                | Adam/AMSGrad.M1context */
    GE_STATE_GSQ_FBIAS = 6,    /* float[V]    Adagrad.gradSqFBias     (J/opt/grad/Adagrad.java:17) | Adam/AMSGrad.M1fBias   */
    GE_STATE_GSQ_CBIAS = 7,    /* float[V]    Adagrad.gradSqCBias                                  | Adam/AMSGrad.M1cBias   */
    GE_STATE_M2_FOCUS = 8,     /* float[V*D]  Adam/AMSGrad.M2focus    (J/opt/grad/Adam.java:40-41); empty for Adagrad */
    GE_STATE_M2_CONTEXT = 9,   /* float[V*D]  Adam/AMSGrad.M2context */
    GE_STATE_M2_FBIAS = 10,    /* float[V]    Adam/AMSGrad.M2fBias   */
    GE_STATE_M2_CBIAS = 11,    /* float[V]    Adam/AMSGrad.M2cBias   */
    GE_STATE_COUNT = 12
};

/* ------------------------------------------------------------------------------------------
 * Trainer.  Replaces `new Adagrad(coMatrix, config, costFunction)` + IOptimizer
 * (J/opt/IOptimizer.java:6-11; J/opt/Optimizer.java:34-64; J/opt/grad/Adagrad.java:19-34).
 * ---------------------------------------------------------------------------------------- */
typedef struct ge_glove ge_glove;

typedef struct {
    int32_t vocab_size;     /* V = coMatrix.vocabSize()          (J/opt/Optimizer.java:40)            */
    int32_t dim;            /* D = config.getDim()               (J/opt/Optimizer.java:43); V*D < 2^31 as in Java */
    int64_t nnz;            /* N = coMatrix.coOccurrenceCount()  (J/opt/Optimizer.java:42); N < 2^31 as in Java  */
    int32_t cost;           /* GE_COST_*                                                               */
    int32_t opt;            /* GE_OPT_*                                                                */
    float   learning_rate;  /* 0.05f in the reference, hard coded (J/opt/Optimizer.java:26)            */
    double  xmax;           /* coMatrix.max()                    (J/opt/GloveCost.java:19)             */
    int64_t seed;           /* Configuration.setThreadLocalRandom(seed) (J/util/config/Configuration.java:161-163) */
    int32_t threads;        /* config.getThreads(): job slicing N/T (+N%T on the last), J/opt/Optimizer.java:59-63.
                               Used by DETERMINISTIC mode only; HOGWILD ignores it.  >= 1.             */
    int32_t mode;           /* GE_MODE_*                                                               */
    int32_t shuffle;        /* GE_SHUFFLE_*                                                            */
    int32_t device;         /* HIP device ordinal                                                      */
    void   *stream;         /* hipStream_t to launch on, or NULL for the device's null stream          */
    int32_t row_begin;      /* multi-GPU row sharding: this handle owns focus rows      */
    int32_t row_end;        /*   [row_begin,row_end); 0,0 = all rows.  I[] must lie inside the range.  */
    int32_t hot_columns;    /* GE_HOT_* (HOGWILD mode only)                                            */
    int32_t workers;        /* HOGWILD: number of sequential workers (wavefronts); 0 = fill the device.
                               Workers pull chunks of 128 consecutive nonzeros of the epoch order from a queue
                               and walk each chunk in stable column order; workers = 1 is fully sequential.
                               workers = -k fills the device except for k wavefront slots, which stay free for
                               kernels running beside the epoch (the all-reduce of an overlapped exchange). */
    int32_t emb_dtype;      /* GE_DTYPE_*: storage of the focus/context rows                           */
    /* HOGWILD tuning; 0 = the library default everywhere.  These change results (which columns publish by delta,
     * how stale a hub run may get), so they live here and under the YAML `device:` block, not in the environment. */
    float   hot_theta;      /* GE_HOT_AUTO: column j is a hub when count(j) * workers >= hot_theta * N.  Default 0.25 */
    float   stale_budget;   /* a hub run is cut (delta published, row re-read) every m_j updates with
                               K_j * m_j <= stale_budget, K_j = expected workers inside column j.  Default 2000
                               (measured: 10 000 is stable at the bench scale, 39 000 diverges)                */
    int32_t flush_every;    /* > 0: cut every hub run after this many updates instead (<= 128)              */
    int32_t blocks_per_cu;  /* > 0: workgroups of 4 workers per CU; default = what the occupancy API reports */
    int32_t layout_flags;   /* GE_LAYOUT_* below                                                            */
} ge_glove_cfg;

/* ge_glove_cfg.layout_flags (GE_SHUFFLE_DEVICE handles).  Default 0: focus rows are packed whole into chunks, so a row
 * is resident in one worker per epoch; a row with more than 128 ordinary nonzeros is cut into pieces that publish the row
 * by delta (float atomics; bf16 rows and Adam/AMSGrad: re-read + add + store), so no worker overwrites another one's run.
 * FIXED_CUTS     round-1 layout: chunks cut every 128 positions whatever the rows (a row cut by a chunk boundary can be
 *                resident in two workers; the later store discards the other worker's whole run) -- ablation / tests.
 * PLAIN_LONG_ROWS  pieces of long rows store plainly (same hazard for those rows only) -- ablation / tests.
 * SEPARATE_TABLES  round-1 storage: a table per kind.  Default (fp32 rows): a side's row and its accumulator row (and
 *                second-moment row) are ONE record of 2 (3) x row width floats, so the loads / stores of one update touch
 *                one region per side instead of two -- measured 7 % faster in alternation and free of the slow placements
 *                separate tables fall into -- ablation / tests.
 * PACKED_RECORDS  fat rows exactly dim + 4 floats wide and records back to back, as before the alignment work.  Default:
 *                records start on 64-byte boundaries, and an fp32 row with dim % 4 == 0 is as wide as the whole 64-byte lines
 *                that hold dim + 1 floats (dim 200: 208; skipped where that would add more than 10 %), so every row store writes
 *                whole lines and a row shares no line with its accumulator row -- ablation / tests.
 * FIRST_PLACEMENT  take the record tables where the first allocation puts them.  Default: each side's table is allocated up to six
 *                times (the earlier ones held meanwhile), a millisecond of random record reads and write-backs is timed on each, the
 *                fastest is kept and the others freed -- where the driver places a table decides which of two epoch times a handle
 *                gets (same process, same virtual address, 48 or 54 ms) -- ablation / tests. */
enum { GE_LAYOUT_FIXED_CUTS = 1, GE_LAYOUT_PLAIN_LONG_ROWS = 2, GE_LAYOUT_SEPARATE_TABLES = 4, GE_LAYOUT_PACKED_RECORDS = 8, GE_LAYOUT_FIRST_PLACEMENT = 16 };

/* What the library decided for a handle (reporting / DESIGN.md numbers). */
typedef struct {
    int32_t group_width;      /* lanes that own one nonzero (16, 32 or 64)                 */
    int32_t vector_width;     /* floats per lane access (4, 2 or 1)                        */
    int32_t chunks_per_lane;
    int32_t blocks;           /* workgroups launched per epoch                             */
    int32_t groups_in_flight; /* sequential workers (wavefronts) in flight                 */
    int32_t hot_columns;      /* columns updated with atomics                              */
    int64_t hot_nonzeros;     /* nonzeros whose column is hot                              */
    int64_t hot_threshold;    /* count(j) >= this => hot                                   */
    int64_t chunks;           /* GE_SHUFFLE_DEVICE: chunks per epoch (<= 128 nonzeros each) ...          */
    int64_t hub_chunks;       /*   ... of which hub-column chunks                                         */
    int64_t long_rows;        /*   focus rows cut into pieces (more than 128 ordinary nonzeros)           */
    int64_t shared_chunks;    /*   chunks that are such a piece (they publish the row by delta)           */
    int32_t flush_min;        /* smallest flush limit of a hub run                                        */
    int32_t row_stride;       /* floats between rows of the fp32 row tables                               */
    int64_t runs;             /*   runs per epoch: a run loads its resident row pair once and publishes it once */
    int64_t schedule_bytes;   /* bytes one epoch of this schedule has to move (what bench.py's roofline.achieved divides by the
                                 kernel time): per nonzero 20 B of (bA, bB, L, W) + the streamed row and its accumulator row(s)
                                 loaded and stored; per run the resident row and its accumulator row(s) loaded and published */
    int32_t placements;       /* allocations tried for the record tables, both sides together (2 = no choice was made)              */
    float   placement_best_ms, placement_worst_ms;   /* probe times of the kept and of the slowest candidate, summed over the two sides */
    int32_t reserved_;
} ge_glove_info;


/* Fills *cfg with the reference's defaults (adagrad, lr 0.05f, threads 1, HOGWILD, DEVICE shuffle). */
void ge_glove_cfg_default(ge_glove_cfg *cfg);

/* I, J, X: host arrays of length cfg->nnz in matrix (pre-shuffle) order, i.e. what
 * cIdx_I/J/C(k) return before the first shuffle() (J/util/CoOccurrenceMatrix.java:12-14).
 * Initialises focus/context/biases from java.util.Random(seed) in the reference's draw order
 * (J/opt/Optimizer.java:50-57), gradSq* = 1 (J/opt/grad/Adagrad.java:27-33), identity permutation. */
ge_status ge_glove_create(const ge_glove_cfg *cfg,
                          const int32_t *I, const int32_t *J, const float *X,
                          ge_glove **out);

/* One pass of the body of Optimizer.optimize()'s loop (J/opt/Optimizer.java:79-94):
 * shuffle, run all jobs, *cost_sum = sum over jobs of the job cost (what `localCost` holds at
 * :94, BEFORE the division by coCount at :96).  `iteration` mirrors createJob's argument
 * (unused by AdaGrad; it keys the DEVICE shuffle). */
ge_status ge_glove_epoch(ge_glove *h, int32_t iteration, double *cost_sum);

/* Optimizer.extractResult (J/opt/Optimizer.java:129-140): out[k] = (focus[k]+context[k])/2.
 * _f32 keeps fp32; _f64 widens as the Java double[] does. out has V*D elements (row-major). */
ge_status ge_glove_extract_f32(ge_glove *h, float *out);
ge_status ge_glove_extract_f64(ge_glove *h, double *out);

/* Copy one GE_STATE_* table device->host / host->device (count = number of floats). */
ge_status ge_glove_get_state(ge_glove *h, int32_t which, float *out, int64_t count);
ge_status ge_glove_set_state(ge_glove *h, int32_t which, const float *in, int64_t count);
/* Raw device pointer of a table (for zero-copy wrapping by the host runtime, e.g. the
 * torch.distributed/RCCL all-reduce of the context factors). Valid until destroy.
 * GE_MODE_DETERMINISTIC handles: the table as the API shows it, *count floats.  GE_MODE_HOGWILD handles with fp32 rows keep
 * FAT rows (a row's bias at [dim]; row width dim + 4 or the whole lines that hold it) inside records of ge_context_layout.row_stride floats (row | accumulator
 * row | ...): a row table id returns the address of ITS row in the first record and *count = the floats from there to the end
 * of its row in the last record; a bias table id returns the address of row 0's scalar inside the row that carries it (FBIAS ->
 * column [dim] of FOCUS, GSQ_CBIAS -> column [dim] of GSQ_CONTEXT ...; with bf16 rows both scalars follow the accumulator row:
 * GSQ_*BIAS at its column [dim], *BIAS at [dim + 1]), the scalar of row r being element r * row_stride from there (row_stride:
 * ge_glove_info.row_stride floats).  bf16 row tables are refused (ge_glove_context_layout). */
ge_status ge_glove_device_ptr(ge_glove *h, int32_t which, void **dptr, int64_t *count);

/* The order in which ONE worker (cfg.workers = 1) walks the nonzeros in epoch `iteration` of a HOGWILD handle:
 * out[k] = index into the caller's I/J/X of the k-th update.  With more workers the same chunks of 128 are
 * handed out in this order but run concurrently.  (GE_SHUFFLE_JAVA: the order of the most recent epoch.)
 * Lets a sequential implementation replay the device pass exactly (parity tests). */
ge_status ge_glove_epoch_order(ge_glove *h, int32_t iteration, int32_t *out, int64_t count);

/* Current permutation (GE_SHUFFLE_JAVA only) and java.util.Random state, for parity tests. */
ge_status ge_glove_get_perm(ge_glove *h, int32_t *out, int64_t count);
ge_status ge_glove_rng_state(ge_glove *h, uint64_t *state);

/* Device time of the last epoch's update kernel(s), measured with hipEvents on the launch
 * stream, in milliseconds; *launches = number of kernel launches it covered. */
ge_status ge_glove_last_kernel_ms(ge_glove *h, float *ms, int32_t *launches);

ge_status ge_glove_get_info(ge_glove *h, ge_glove_info *info);

void ge_glove_destroy(ge_glove *h);

/* ------------------------------------------------------------------------------------------
 * Co-occurrence builder.  Replaces `new BookmarkColoring(graph, config)`
 * (J/bca/BookmarkColoring.java:32-120) once the host has flattened the grph graph into
 * weighted CSR (out-neighbours) + CSC (in-neighbours), which is what
 * In/OutEdgeNeighborhoodAlgorithm.compute + getIn/OutNeighborhoods produce
 * (J/bca/BookmarkColoring.java:49-52; J/convert/util/EdgeNeighborhoodAlgorithm.java:22-33).
 * ---------------------------------------------------------------------------------------- */
typedef struct ge_coo ge_coo;

typedef struct {
    int32_t        num_vertices;   /* V */
    const int64_t *ptr;            /* V+1 offsets */
    const int32_t *idx;            /* neighbour ids, unique within a row, in grph neighbour order */
    const float   *weight;         /* edge weight per neighbour (NumericalProperty.getValueAsFloat) */
} ge_csr;

typedef struct {
    double  alpha;       /* bca.alpha   (J/util/config/Configuration.java:322) */
    double  epsilon;     /* bca.epsilon */
    int32_t directed;    /* bca.directed: 1 = DirectedWeighted (forward + forced reverse pass), 0 = UndirectedWeighted */
    int32_t normalize;   /* GE_NORM_* */
    int32_t device;      /* HIP device ordinal */
    int32_t row_begin;   /* bookmarks [row_begin,row_end) only; 0,0 = all (multi-GPU sharding) */
    int32_t row_end;
    int64_t table_slots; /* sizing only, results do not depend on it: slots of a wavefront's work table (0 = from V; it
                            grows by itself on overflow) ...                                                       */
    int64_t pool_entries;/*   ... and entries of the first row pool (0 = from a sample of 2048 bookmarks)            */
} ge_bca_cfg;
/* Diagnostics read from the environment (never results): GE_BCA_TIMING=1 prints the phases of ge_bca_build to stderr. */

/* Runs one BCA job per bookmark (BCAJob.call, J/bca/util/BCAJob.java:31-36) on the device and
 * assembles the COO in ascending bookmark order with each row in java.util.HashMap iteration
 * order -- the order BookmarkColoring produces with `threads: 1`. */
ge_status ge_bca_build(const ge_csr *out_nbrs, const ge_csr *in_nbrs, const ge_bca_cfg *cfg, ge_coo **result);

/* Host views into the result (owned by the handle, valid until ge_coo_destroy):
 * I/J/X = coOccurrenceIdx_I/_J/Values (J/bca/BookmarkColoring.java:23-25), *max = max() (:153),
 * row_ptr[V+1] = first entry of each bookmark.  Any out pointer may be NULL. */
ge_status ge_coo_get(const ge_coo *c, int64_t *nnz, const int32_t **I, const int32_t **J,
                     const float **X, const int64_t **row_ptr, double *max);
void ge_coo_destroy(ge_coo *c);

/* ------------------------------------------------------------------------------------------ */
/* Literal-similarity edges.  Replaces the compare loop of Rdf2GrphConverter.convert
 * (J/convert/Rdf2GrphConverter.java:127-186): for one CompareGroup, CompareJob i (J/compare/CompareJob.java:33-51)
 * walks target[startIndex..] and keeps the pairs with metric.similarity(s1, s2) >= threshold; the caller then adds the
 * two directed edges per pair (:163-173).  Methods = Configuration.SimilarityMethod ordinals
 * (J/util/config/Configuration.java:27-29), metrics = SimilarityGroup.toFunction (:201-227). */
enum { GE_SIM_NGRAM_COSINE = 0, GE_SIM_NGRAM_JACCARD = 1, GE_SIM_TOKEN_COSINE = 2, GE_SIM_TOKEN_JACCARD = 3,
       GE_SIM_JAROWINKLER = 4, GE_SIM_LEVENSHTEIN = 5, GE_SIM_NUMERIC = 6,
       GE_SIM_DATE_DAYS = 7, GE_SIM_DATE_MONTHS = 8, GE_SIM_DATE_YEARS = 9 };
enum { GE_TIME_BACKWARDS = 0, GE_TIME_FORWARDS = 1, GE_TIME_BIDIRECTIONAL = 2 };   /* SimilarityGroup.Time (:184-186) */

typedef struct {
    int32_t method;          /* GE_SIM_*                                                                    */
    double  threshold;       /* SimilarityGroup.getThreshold                                                */
    int32_t ngram;           /* getNgram (0 = 3)                                                            */
    double  smooth;          /* getSmooth (0 = 1); Numeric's alpha                                          */
    double  distance;        /* getDistance                                                                 */
    int32_t time;            /* GE_TIME_* (Date* only)                                                      */
    const char *pattern;     /* getPattern: NULL or "iso" = BASIC_ISO_DATE; else the ofPattern subset
                                yyyy|uuuu, MM|M, dd|d, quoted text and literal characters                   */
    int32_t upper_triangle;  /* CompareGroup.upperTriangle: source predicate == target predicate, job i starts
                                at target i+1 (source and target must then be the same list)               */
    int32_t device;          /* HIP device ordinal                                                          */
    int32_t job_begin;       /* multi-GPU sharding: only CompareJobs (source positions) [job_begin, job_end) run;    */
    int32_t job_end;         /*   0,0 = all.  Positions in the result stay global; concatenating the shards' results
                                in job order gives the whole group's result (no collective involved).             */
} ge_sim_cfg;

/* A table of java.lang.String values: string s = UTF-16 code units [offset[s], offset[s+1]) of `units`
 * (what JNI's GetStringChars returns).  At most 1024 units per string that takes part in a comparison. */
typedef struct {
    int32_t         count;
    const int64_t  *offset;  /* count + 1 ascending entries */
    const uint16_t *units;
} ge_strings;

typedef struct ge_sim_pairs ge_sim_pairs;

void ge_sim_cfg_default(ge_sim_cfg *cfg);
int32_t ge_sim_cfg_size(void);      /* sizeof(ge_sim_cfg) as the library was compiled (same purpose as ge_glove_cfg_size) */
/* 1 when Date* can parse with this pattern (NULL/"iso" included), 0 when it is outside the supported subset. */
int32_t ge_sim_pattern_supported(const char *pattern);
/* source[i] / target[j] are positions in `strings` (vertexLabels.getValueAsString of sourceNodes[i] / targetNodes[j]),
 * source_vertex / target_vertex the vertex ids (a job skips its own vertex, CompareJob.java:38).  The result lists
 * (i, j, (float) similarity) in job order -- i ascending, j ascending inside a job -- which is the order results
 * arrive in with `threads: 1`.  A Numeric job that dies in String.substring (Numeric.java:36) yields nothing, as its
 * ExecutionException does in the reference (Rdf2GrphConverter.java:176-178). */
ge_status ge_similarity_pairs(const ge_strings *strings,
                              const int32_t *source, const int32_t *source_vertex, int32_t n_source,
                              const int32_t *target, const int32_t *target_vertex, int32_t n_target,
                              const ge_sim_cfg *cfg, ge_sim_pairs **result);
/* Host views, valid until ge_sim_pairs_destroy.  Any out pointer may be NULL. */
ge_status ge_sim_pairs_get(const ge_sim_pairs *r, int64_t *count, const int32_t **source_pos, const int32_t **target_pos,
                           const float **similarity);
void ge_sim_pairs_destroy(ge_sim_pairs *r);

/* ------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------ */
/* Where the context rows of a handle live on the device (ge_sync works on these; tests and tools wrap them zero-copy).  dtype GE_DTYPE_F32:
 * `table` is float[vocab_size*row_stride] and the hub fields are NULL/0.  GE_DTYPE_BF16: `table` is bf16[vocab_size*dim];
 * a column v with hub_index[v] >= 0 keeps its current value in hub_rows[hub_index[v]*dim ..] (fp32 master row; the
 * bf16 copy of such a row is stale until extraction).  Which columns are hubs is decided per handle from ITS nonzeros. */
typedef struct {
    void          *table;
    int32_t        dtype;        /* GE_DTYPE_* */
    float         *hub_rows;     /* [n_hub x dim] */
    const int32_t *hub_index;    /* [vocab_size], device memory */
    int32_t        n_hub, vocab_size, dim;
    int32_t        row_stride;   /* elements of `table`'s dtype between consecutive rows of `table`.  fp32 GE_MODE_HOGWILD handles keep
                                    FAT rows (a row's bias at [dim]; width dim + 4, widened to whole 64-byte lines where that costs under
                                    10 %: dim 200 -> 208 floats) inside records [row | accumulator row] that start on 64-byte boundaries:
                                    row_stride = the record, 2 row widths rounded up to 16 floats (dim 200: 416; ge_glove_info.row_stride
                                    says what a handle uses); bf16 rows lead records [bf16 row padded to 16 B | fp32 accumulator row, fat:
                                    gradSq (dim) | its bias accumulator | the bias | 2 x 0]: row_stride = that record in bf16
                                    elements; GE_LAYOUT_SEPARATE_TABLES: the row width itself */
    float         *accum;        /* gradSqContext (Adam/AMSGrad: M1context); fp32 always; fat like `table` when that is fp32 */
    int32_t        accum_stride; /* floats between consecutive rows of `accum` (its bias accumulator at [dim] when fat)             */
    float         *bias;         /* cBias[v] = bias[v * bias_stride]: a vector of its own (stride 1), column [dim] of the fat fp32 row,
                                    or -- bf16 rows, which cannot carry it -- column [dim + 1] of the accumulator row                */
    int32_t        bias_stride;
    float         *accum_bias;   /* gradSqCBias[v] = accum_bias[v * accum_bias_stride] (fat: column [dim] of the accumulator row)      */
    int32_t        accum_bias_stride;
} ge_context_layout;
ge_status ge_glove_context_layout(ge_glove *h, ge_context_layout *out);

/* The elementwise half of one exchange step for a GE_DTYPE_BF16 context table (what ge_sync runs for bf16 rows; exported for
 * its parity test).  One pass over device memory on `stream`, asynchronous:
 *   land != 0:  row value += wire - own, base += wire - own   (`wire` = the all-reduced sum of every rank's bf16 delta, `own`
 *               this rank's part of it: the difference is what the others sent);
 *   take != 0:  d = bf16(row value - base) (before landing); own = d; base += d.  `wire` is never written here: it is the
 *               receive buffer of the all-reduce that follows (send buffer own).
 * A row's value lives in hub_rows[hub_index[v]] (fp32 master) when the column is a hub on this rank, else in the bf16 table.
 * The table's rows are `row_stride` bf16 elements apart (a multiple of 4; ge_context_layout.row_stride).
 * `base` is float[vocab_size*dim] for EVERY row (start: the row values widened), wire / own are bf16[vocab_size*dim].  Landed
 * ordinary rows are re-narrowed with stochastic rounding drawn from `seed` (a new one every turn); that rounding is part of
 * value - base and is fed back with the next delta. */
ge_status ge_exchange_turn_bf16(uint16_t *table, int32_t row_stride, float *hub_rows, const int32_t *hub_index, int32_t vocab_size, int32_t dim,
                                float *base, uint16_t *wire, uint16_t *own, int32_t land, int32_t take, uint32_t seed, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Multi-GPU trainer behind the C ABI (north_star: "rows shard across the 8 GPUs of one node with a periodic
 * RCCL all-reduce of the context factors").  One ge_glove handle per GPU (cfg.device, cfg.row_begin/row_end = its block of
 * focus rows, its nonzeros), one ge_sync per handle; the host runs the handles in lockstep -- one process per GPU, or one
 * thread per GPU inside one process (the JVM) -- and after every ge_glove_epoch calls ge_sync_turn (or ge_sync_sync).
 * The reference is a single JVM and has no counterpart; the merge rule is product semantics and lives in the library:
 *   context rows and both AdaGrad accumulators: the ranks' deltas ADD; cBias: the MEAN over the ranks that moved the element
 *   (the reference updates biases without a learning rate, J/opt/grad/Adagrad.java:88-89); the accumulators are reconciled
 *   only every accum_every-th exchange.  GE_MODE_HOGWILD + GE_OPT_ADAGRAD handles (fp32 or bf16 rows). */
typedef struct ge_sync ge_sync;
typedef struct ge_local_group ge_local_group;

/* A collective supplied by the host instead of RCCL (tests: torch.distributed / gloo with two ranks on one GPU; the C++ CLI:
 * N ranks inside one process).  buf is DEVICE memory of `count` elements of GE_DTYPE_F32 or GE_DTYPE_BF16; the library has
 * drained its stream before it calls.  start: begin summing buf over all ranks in place, *ticket identifies the operation;
 * wait: return when that sum is in buf; broadcast: every rank's buf becomes rank src's (blocking).  Return GE_OK or GE_ERR_*. */
typedef struct {
    void *user;
    ge_status (*start)(void *user, void *buf, int64_t count, int32_t dtype, void **ticket);
    ge_status (*wait)(void *user, void *ticket);
    ge_status (*broadcast)(void *user, void *buf, int64_t count, int32_t dtype, int32_t src);
} ge_transport;

typedef struct {
    int32_t world, rank;           /* ranks = GPUs; world == 1: every call is a no-op                                        */
    int32_t wire;                  /* GE_DTYPE_BF16: the deltas of the row / accumulator tables travel as bf16 (half the bytes
                                      over xGMI; what the rounding drops is fed back with the next delta); GE_DTYPE_F32        */
    int32_t accum_every;           /* accumulators every Nth exchange (0 = 4)                                                  */
    const ge_transport *transport; /* NULL: RCCL (the library opens librccl.so.1 at run time) ...                              */
    const void *rccl_id;           /* ... with this 128-byte ncclUniqueId: ge_rccl_unique_id on rank 0, handed to every rank   */
    ge_local_group *local_group;   /* or: the ranks are threads of THIS process and meet in host memory (any number of devices;
                                      a rehearsal of an N-rank run on fewer GPUs, blocking, no overlap)                        */
} ge_sync_cfg;

/* The ranks of one process (one host thread per rank): created once, handed to every rank's ge_sync_cfg, destroyed after the
 * last ge_sync_destroy. */
ge_status ge_local_group_create(int32_t world, ge_local_group **out);
void ge_local_group_destroy(ge_local_group *g);
/* A rank thread that fails OUTSIDE the ge_sync calls (its ge_glove_create, its epoch) calls this before it leaves: every peer
 * waiting in -- or arriving at -- an exchange of the group returns GE_ERR_STATE instead of waiting for a rank that will not
 * come.  (A rank that fails INSIDE a ge_sync call aborts its group itself.)  The group stays aborted. */
void ge_local_group_abort(ge_local_group *g);

ge_status ge_rccl_unique_id(void *id128);
/* Opens RCCL, makes a one-rank communicator on `device`, runs a sum and a broadcast through it and checks the data: what a
 * single-GPU machine can verify of the RCCL path. */
ge_status ge_rccl_selftest(int32_t device);
int32_t ge_sync_cfg_size(void);
/* Collective: every rank calls it (RCCL: ncclCommInitRank inside).  The base of every table is its value NOW.  A ge_sync
 * works on its handle's device tables: destroy it before the ge_glove it was created for. */
ge_status ge_sync_create(ge_glove *h, const ge_sync_cfg *cfg, ge_sync **out);
/* begin = take: the deltas since the last take go on the wire (everything != 0: accumulators too), asynchronously on RCCL's
 * own stream.  finish = land: waits for them and adds what the OTHER ranks sent.  turn = finish, then begin: the all-reduce of
 * step k runs under ge_glove_epoch of step k+1 (launch that epoch with cfg.workers = -256 so RCCL's kernels find room), and a
 * rank sees the others' moves one step late.  sync = turn, then finish: exact replicas after every step, nothing left in flight.  All enqueue on the
 * handle's stream and return; none blocks the host with RCCL. */
ge_status ge_sync_begin(ge_sync *s, int32_t everything);
ge_status ge_sync_finish(ge_sync *s);
ge_status ge_sync_turn(ge_sync *s);
ge_status ge_sync_sync(ge_sync *s);
/* One epoch of a sharded run, to be called INSTEAD of ge_glove_epoch by every rank (collective).  On the way the HUB rows of the
 * context side -- the union of the ranks' busy columns (count on a rank >= max(256, N_rank / 20 480); from four ranks on the divisor grows with the ranks), a few thousand rows -- are
 * reconciled `segments` times in small fp32 all-reduces (<= 0: as often as the busiest column asks for -- one exchange per 65 536
 * updates that all ranks together put on it, at least max(8, ranks) --; at most 128 live, 64 in segments).
 * Without it eight ranks that each push a busy row for a whole epoch from the same start overshoot where one GPU settles, and with too few
 * exchanges for a very busy column the run leaves the single-GPU trajectory (measured: DESIGN.md 7); inside a GPU the same rows are held
 * together by publishing deltas every few updates, across GPUs by this.  Two forms:
 *   live      (over RCCL or a local group): the epoch is ONE launch and the exchanges run BESIDE it on a stream of their own --
 *             the epoch kernel moves its hub columns by atomic adds only (a sharded handle counts every column that is busy on the rank
 *             among them), so the other ranks' deltas are added the same way (k_live_take / k_live_land) -- paced by the epoch's
 *             ticket counter; the epoch kernel never waits.  Should the exchanges fall behind the epoch (a quarter of them a whole
 *             interval late on half of the ranks, two epochs running: a slow transport, epochs of a few milliseconds), the ranks agree to
 *             continue in segments.
 *             bf16 rows: on the fp32 master rows of the columns that are hubs on every rank (the few at the threshold that are not wait for
 *             the end of the epoch).
 *   segments  (a host transport, GE_SYNC_EPOCH=segments, a run that fell behind): the epoch runs in `segments` launches and behind each one the hub
 *             rows are reconciled exactly (both accumulators summed, cBias averaged over the ranks that moved it, the parameter rows' summed
 *             deltas scaled per element by sqrt((G0 + E / W) / (G0 + E)) -- G0 the accumulator at the last exchange, E this exchange's summed
 *             accumulator deltas: every rank stepped without the others' gradients in its accumulator, and W such pushes summed as they
 *             are overshoot by up to sqrt(W); GE_SYNC_MERGE=sum turns the scaling off --; the live form merges the same way).
 * Both end the epoch with that exact exchange of all hub rows.  ge_sync_turn / ge_sync_sync follow as before (they find nothing left to
 * do for the hub rows).  *cost_sum as ge_glove_epoch.  A bf16 handle reads and writes a hub row where IT keeps it: the fp32 master row
 * of a column that is a hub on this rank, else the bf16 table entry, stochastically rounded.  A one-rank run and a run without hub
 * columns get one plain ge_glove_epoch. */
ge_status ge_sync_epoch(ge_sync *s, int32_t iteration, int32_t segments, double *cost_sum);
/* The hub rows of this run (*count of them, ascending; out may be NULL or shorter), and ONE small exchange of them now (collective; what
 * ge_sync_epoch does behind every segment) -- for a host that cuts its epochs itself. */
ge_status ge_sync_hub_rows(ge_sync *s, int32_t *out, int32_t capacity, int32_t *count);
ge_status ge_sync_hub_exchange(ge_sync *s);
/* One LIVE exchange now (collective; GE_ERR_STATE when the run has no live rows), and what ge_sync_epoch(s, ., segments, .) will do:
 * *live = 1 beside the running kernel / 0 in segments, *exchanges per epoch, *live_rows = rows exchanged live (0 in segments). */
ge_status ge_sync_hub_exchange_live(ge_sync *s);
ge_status ge_sync_live_rows(ge_sync *s, int32_t *out, int32_t capacity, int32_t *count);      /* the rows a live exchange covers (ascending; as ge_sync_hub_rows) */
ge_status ge_sync_hub_plan(ge_sync *s, int32_t segments, int32_t *live, int32_t *exchanges, int32_t *live_rows);
/* Ends a run: lands what is in flight, exchanges everything not sent yet, then every rank takes rank src's fp32 tables. */
ge_status ge_sync_replicate(ge_sync *s, int32_t src);
/* n host doubles summed (op 0) or maximised (op 1) over the ranks through RCCL: the epoch's cost (Optimizer.java:94-96 needs
 * the sum over all jobs), BookmarkColoring's max over shards.  Blocking, and ordered on the communicator BEHIND whatever
 * ge_sync_turn has put on the wire: reduce an epoch's cost BEFORE that epoch's ge_sync_turn (epoch -> allreduce(cost) ->
 * turn), or the host waits for the whole context all-reduce and the overlap with the next epoch is lost. */
ge_status ge_sync_allreduce_f64(ge_sync *s, double *values, int32_t n, int32_t op);
void ge_sync_destroy(ge_sync *s);

/* ------------------------------------------------------------------------------------------ */
const char *ge_last_error(void);     /* message of the calling thread's last failed call */
const char *ge_version(void);
/* sizeof(ge_glove_cfg) as the LIBRARY was compiled: a host built against another header revision must refuse to
 * run instead of letting ge_glove_cfg_default write past its struct. */
int32_t ge_glove_cfg_size(void);
int32_t ge_bca_cfg_size(void);
/* Diagnostic: the rate (GB/s, bytes read + written) of a plain 16-byte-per-lane device-to-device copy of `bytes` on this
 * device -- the practical ceiling bench.py prints beside a kernel's own rate (the boxes of a pool differ). */
ge_status ge_copy_bandwidth(int32_t device, int64_t bytes, int32_t reps, double *gbps);
/* Number of visible HIP devices that are gfx950; <0 on HIP error. Does not compute. */
int32_t ge_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* GEGLOVE_H */
