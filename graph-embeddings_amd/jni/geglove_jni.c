/*
 * geglove_jni.c -- JNI glue between the reference's Java host and libgeglove.so.
 * The build image has no JDK (no jni.h, no javac), so this file is compiled where one exists:
 *     gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include geglove_jni.c \
 *         -L../lib -lgeglove -o libgeglove_jni.so
 * (tests/test_capi_and_host.py compiles it here against a minimal jni.h stand-in to keep syntax and signatures honest).
 * Java side: org.uu.nl.embedding.hip.Native.  Every entry point maps a non-zero ge_status to an exception
 * carrying ge_last_error(); no global references are kept; arrays are taken with Get<Type>ArrayElements for the duration of
 * the native call (never critical pins: ge_glove_create uploads and sorts, the JVM's collector must stay free to run).
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "geglove.h"

static void throw_ge(JNIEnv *env, ge_status st) {
    jclass ex = (*env)->FindClass(env, st == GE_ERR_ARG ? "java/lang/IllegalArgumentException" : "java/lang/RuntimeException");
    if (ex) (*env)->ThrowNew(env, ex, ge_last_error());
}
static void throw_msg(JNIEnv *env, const char *msg) {
    jclass ex = (*env)->FindClass(env, "java/lang/IllegalStateException");
    if (ex) (*env)->ThrowNew(env, ex, msg);
}

/* Indices of gloveCreate's int options (org.uu.nl.embedding.hip.Native.OPT_*) and float options (TUNE_*): every field of
 * ge_glove_cfg that Main.createOptimizer (J/Main.java:107-131) or the `device:` block can set. */
enum { OPT_COST, OPT_OPT, OPT_THREADS, OPT_MODE, OPT_SHUFFLE, OPT_DEVICE, OPT_ROW_BEGIN, OPT_ROW_END, OPT_HOT_COLUMNS, OPT_WORKERS,
       OPT_EMB_DTYPE, OPT_FLUSH_EVERY, OPT_BLOCKS_PER_CU, OPT_LAYOUT_FLAGS, OPT_COUNT };
enum { TUNE_LEARNING_RATE, TUNE_HOT_THETA, TUNE_STALE_BUDGET, TUNE_COUNT };

/* long gloveCreate(int V, int D, int[] I, int[] J, float[] X, double xmax, long seed, int[] opts, float[] tune)
 * opts / tune: OPT_COUNT ints and TUNE_COUNT floats; a 0 learning rate means the reference's 0.05f. */
JNIEXPORT jlong JNICALL Java_org_uu_nl_embedding_hip_Native_gloveCreate(
        JNIEnv *env, jclass cls, jint V, jint D, jintArray I, jintArray J, jfloatArray X, jdouble xmax, jlong seed,
        jintArray opts, jfloatArray tune) {
    (void)cls;
    if (ge_glove_cfg_size() != (int32_t)sizeof(ge_glove_cfg)) { throw_msg(env, "libgeglove.so and libgeglove_jni.so were built from different revisions of geglove.h"); return 0; }
    if ((*env)->GetArrayLength(env, opts) != OPT_COUNT || (*env)->GetArrayLength(env, tune) != TUNE_COUNT) { throw_msg(env, "gloveCreate: opts / tune have the wrong length for this glue"); return 0; }
    jint o[OPT_COUNT]; jfloat t[TUNE_COUNT];
    {
        jint *po = (*env)->GetIntArrayElements(env, opts, NULL); jfloat *pt = (*env)->GetFloatArrayElements(env, tune, NULL);
        if (po) memcpy(o, po, sizeof o);
        if (pt) memcpy(t, pt, sizeof t);
        if (po) (*env)->ReleaseIntArrayElements(env, opts, po, JNI_ABORT);
        if (pt) (*env)->ReleaseFloatArrayElements(env, tune, pt, JNI_ABORT);
        if (!po || !pt) { throw_ge(env, GE_ERR_OOM); return 0; }
    }
    ge_glove_cfg cfg;
    ge_glove_cfg_default(&cfg);
    cfg.vocab_size = V; cfg.dim = D; cfg.nnz = (*env)->GetArrayLength(env, I);
    cfg.xmax = xmax; cfg.seed = seed;
    cfg.cost = o[OPT_COST]; cfg.opt = o[OPT_OPT]; cfg.threads = o[OPT_THREADS]; cfg.mode = o[OPT_MODE]; cfg.shuffle = o[OPT_SHUFFLE];
    cfg.device = o[OPT_DEVICE]; cfg.row_begin = o[OPT_ROW_BEGIN]; cfg.row_end = o[OPT_ROW_END]; cfg.hot_columns = o[OPT_HOT_COLUMNS];
    cfg.workers = o[OPT_WORKERS]; cfg.emb_dtype = o[OPT_EMB_DTYPE]; cfg.flush_every = o[OPT_FLUSH_EVERY];
    cfg.blocks_per_cu = o[OPT_BLOCKS_PER_CU]; cfg.layout_flags = o[OPT_LAYOUT_FLAGS];
    if (t[TUNE_LEARNING_RATE] > 0) cfg.learning_rate = t[TUNE_LEARNING_RATE];
    cfg.hot_theta = t[TUNE_HOT_THETA]; cfg.stale_budget = t[TUNE_STALE_BUDGET];
    jint *pi = (*env)->GetIntArrayElements(env, I, NULL);
    jint *pj = (*env)->GetIntArrayElements(env, J, NULL);
    jfloat *px = (*env)->GetFloatArrayElements(env, X, NULL);
    ge_glove *h = NULL;
    ge_status st = (pi && pj && px) ? ge_glove_create(&cfg, (const int32_t *)pi, (const int32_t *)pj, px, &h) : GE_ERR_OOM;
    if (px) (*env)->ReleaseFloatArrayElements(env, X, px, JNI_ABORT);
    if (pj) (*env)->ReleaseIntArrayElements(env, J, pj, JNI_ABORT);
    if (pi) (*env)->ReleaseIntArrayElements(env, I, pi, JNI_ABORT);
    if (st != GE_OK) { throw_ge(env, st); return 0; }
    return (jlong)(intptr_t)h;
}

/* double gloveEpoch(long handle, int iteration): summed job cost (Optimizer.java:94's localCost) */
JNIEXPORT jdouble JNICALL Java_org_uu_nl_embedding_hip_Native_gloveEpoch(JNIEnv *env, jclass cls, jlong handle, jint iteration) {
    (void)cls;
    double cost = 0;
    ge_status st = ge_glove_epoch((ge_glove *)(intptr_t)handle, iteration, &cost);
    if (st != GE_OK) throw_ge(env, st);
    return cost;
}

/* void gloveExtract(long handle, double[] out): Optimizer.extractResult */
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_gloveExtract(JNIEnv *env, jclass cls, jlong handle, jdoubleArray out) {
    (void)cls;
    jdouble *p = (*env)->GetDoubleArrayElements(env, out, NULL);
    ge_status st = p ? ge_glove_extract_f64((ge_glove *)(intptr_t)handle, p) : GE_ERR_OOM;
    if (p) (*env)->ReleaseDoubleArrayElements(env, out, p, 0);
    if (st != GE_OK) throw_ge(env, st);
}

JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_gloveDestroy(JNIEnv *env, jclass cls, jlong handle) {
    (void)env; (void)cls;
    ge_glove_destroy((ge_glove *)(intptr_t)handle);
}

/* void gloveGetState(long handle, int which, float[] out): one GE_STATE_* table (a sharded run collects its focus rows with it) */
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_gloveGetState(JNIEnv *env, jclass cls, jlong handle, jint which, jfloatArray out) {
    (void)cls;
    jfloat *p = (*env)->GetFloatArrayElements(env, out, NULL);
    ge_status st = p ? ge_glove_get_state((ge_glove *)(intptr_t)handle, which, p, (*env)->GetArrayLength(env, out)) : GE_ERR_OOM;
    if (p) (*env)->ReleaseFloatArrayElements(env, out, p, 0);
    if (st != GE_OK) throw_ge(env, st);
}

/* ---- multi-GPU: one handle + one sync per GPU, ranks = threads of the JVM or processes ---- */
/* byte[] rcclUniqueId(): 128 bytes, made on rank 0, handed to every rank */
JNIEXPORT jbyteArray JNICALL Java_org_uu_nl_embedding_hip_Native_rcclUniqueId(JNIEnv *env, jclass cls) {
    (void)cls;
    jbyte id[128];
    ge_status st = ge_rccl_unique_id(id);
    if (st != GE_OK) { throw_ge(env, st); return NULL; }
    jbyteArray out = (*env)->NewByteArray(env, 128);
    if (out) (*env)->SetByteArrayRegion(env, out, 0, 128, id);
    return out;
}
/* long localGroupCreate(int world) / void localGroupDestroy(long group): ranks that are threads of this JVM and meet in host memory */
JNIEXPORT jlong JNICALL Java_org_uu_nl_embedding_hip_Native_localGroupCreate(JNIEnv *env, jclass cls, jint world) {
    (void)cls;
    ge_local_group *g = NULL;
    ge_status st = ge_local_group_create(world, &g);
    if (st != GE_OK) { throw_ge(env, st); return 0; }
    return (jlong)(intptr_t)g;
}
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_localGroupDestroy(JNIEnv *env, jclass cls, jlong group) {
    (void)env; (void)cls;
    ge_local_group_destroy((ge_local_group *)(intptr_t)group);
}
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_localGroupAbort(JNIEnv *env, jclass cls, jlong group) {
    (void)env; (void)cls;
    ge_local_group_abort((ge_local_group *)(intptr_t)group);
}
/* double syncEpoch(long sync, int iteration, int segments): the rank's epoch with the hub rows reconciled on the way (ge_sync_epoch) */
JNIEXPORT jdouble JNICALL Java_org_uu_nl_embedding_hip_Native_syncEpoch(JNIEnv *env, jclass cls, jlong sync, jint iteration, jint segments) {
    (void)cls;
    double cost = 0;
    ge_status st = ge_sync_epoch((ge_sync *)(intptr_t)sync, iteration, segments, &cost);
    if (st != GE_OK) { throw_ge(env, st); return 0; }
    return cost;
}
/* long syncCreate(long glove, int world, int rank, int wire, int accumEvery, byte[] rcclId (or null), long localGroup (or 0)) */
JNIEXPORT jlong JNICALL Java_org_uu_nl_embedding_hip_Native_syncCreate(
        JNIEnv *env, jclass cls, jlong glove, jint world, jint rank, jint wire, jint accumEvery, jbyteArray rcclId, jlong localGroup) {
    (void)cls;
    if (ge_sync_cfg_size() != (int32_t)sizeof(ge_sync_cfg)) { throw_msg(env, "libgeglove.so and libgeglove_jni.so were built from different revisions of geglove.h"); return 0; }
    jbyte id[128];
    ge_sync_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.world = world; cfg.rank = rank; cfg.wire = wire; cfg.accum_every = accumEvery;
    if (rcclId) { (*env)->GetByteArrayRegion(env, rcclId, 0, 128, id); cfg.rccl_id = id; }
    cfg.local_group = (ge_local_group *)(intptr_t)localGroup;
    ge_sync *s = NULL;
    ge_status st = ge_sync_create((ge_glove *)(intptr_t)glove, &cfg, &s);
    if (st != GE_OK) { throw_ge(env, st); return 0; }
    return (jlong)(intptr_t)s;
}
/* void syncTurn(long sync) / syncSync / syncReplicate(long sync, int src) / syncDestroy */
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_syncTurn(JNIEnv *env, jclass cls, jlong sync) {
    (void)cls;
    ge_status st = ge_sync_turn((ge_sync *)(intptr_t)sync);
    if (st != GE_OK) throw_ge(env, st);
}
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_syncSync(JNIEnv *env, jclass cls, jlong sync) {
    (void)cls;
    ge_status st = ge_sync_sync((ge_sync *)(intptr_t)sync);
    if (st != GE_OK) throw_ge(env, st);
}
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_syncReplicate(JNIEnv *env, jclass cls, jlong sync, jint src) {
    (void)cls;
    ge_status st = ge_sync_replicate((ge_sync *)(intptr_t)sync, src);
    if (st != GE_OK) throw_ge(env, st);
}
/* void syncAllreduce(long sync, double[] values, int op): sum (0) or max (1) over the ranks, in place (the epoch's cost) */
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_syncAllreduce(JNIEnv *env, jclass cls, jlong sync, jdoubleArray values, jint op) {
    (void)cls;
    jdouble *p = (*env)->GetDoubleArrayElements(env, values, NULL);
    ge_status st = p ? ge_sync_allreduce_f64((ge_sync *)(intptr_t)sync, p, (*env)->GetArrayLength(env, values), op) : GE_ERR_OOM;
    if (p) (*env)->ReleaseDoubleArrayElements(env, values, p, 0);
    if (st != GE_OK) throw_ge(env, st);
}
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_syncDestroy(JNIEnv *env, jclass cls, jlong sync) {
    (void)env; (void)cls;
    ge_sync_destroy((ge_sync *)(intptr_t)sync);
}

/* long bcaBuild(int V, long[] outPtr, int[] outIdx, float[] outW, long[] inPtr, int[] inIdx, float[] inW,
 *               double alpha, double epsilon, boolean directed, int normalize, int device, int rowBegin, int rowEnd) */
JNIEXPORT jlong JNICALL Java_org_uu_nl_embedding_hip_Native_bcaBuild(
        JNIEnv *env, jclass cls, jint V, jlongArray outPtr, jintArray outIdx, jfloatArray outW,
        jlongArray inPtr, jintArray inIdx, jfloatArray inW, jdouble alpha, jdouble epsilon,
        jboolean directed, jint normalize, jint device, jint rowBegin, jint rowEnd) {
    (void)cls;
    /* every check that can fail comes BEFORE an array is pinned: an early return must not leave six arrays pinned */
    if (ge_bca_cfg_size() != (int32_t)sizeof(ge_bca_cfg)) { throw_msg(env, "libgeglove.so and libgeglove_jni.so were built from different revisions of geglove.h"); return 0; }
    if (!outPtr || !outIdx || !outW || !inPtr || !inIdx || !inW) { throw_msg(env, "bcaBuild: null neighbourhood array"); return 0; }
    ge_csr out, in;
    memset(&out, 0, sizeof out); memset(&in, 0, sizeof in);
    out.num_vertices = in.num_vertices = V;
    out.ptr = (const int64_t *)(*env)->GetLongArrayElements(env, outPtr, NULL);
    out.idx = (const int32_t *)(*env)->GetIntArrayElements(env, outIdx, NULL);
    out.weight = (*env)->GetFloatArrayElements(env, outW, NULL);
    in.ptr = (const int64_t *)(*env)->GetLongArrayElements(env, inPtr, NULL);
    in.idx = (const int32_t *)(*env)->GetIntArrayElements(env, inIdx, NULL);
    in.weight = (*env)->GetFloatArrayElements(env, inW, NULL);
    if (!out.ptr || !out.idx || !out.weight || !in.ptr || !in.idx || !in.weight) {      /* the JVM could not pin / copy one of them */
        if (out.ptr) (*env)->ReleaseLongArrayElements(env, outPtr, (jlong *)out.ptr, JNI_ABORT);
        if (out.idx) (*env)->ReleaseIntArrayElements(env, outIdx, (jint *)out.idx, JNI_ABORT);
        if (out.weight) (*env)->ReleaseFloatArrayElements(env, outW, (jfloat *)out.weight, JNI_ABORT);
        if (in.ptr) (*env)->ReleaseLongArrayElements(env, inPtr, (jlong *)in.ptr, JNI_ABORT);
        if (in.idx) (*env)->ReleaseIntArrayElements(env, inIdx, (jint *)in.idx, JNI_ABORT);
        if (in.weight) (*env)->ReleaseFloatArrayElements(env, inW, (jfloat *)in.weight, JNI_ABORT);
        if (!(*env)->ExceptionCheck(env)) throw_msg(env, "bcaBuild: out of memory while pinning the neighbourhood arrays");
        return 0;
    }
    ge_bca_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.alpha = alpha; cfg.epsilon = epsilon; cfg.directed = directed ? 1 : 0; cfg.normalize = normalize; cfg.device = device;
    cfg.row_begin = rowBegin; cfg.row_end = rowEnd;              /* a shard of the bookmarks; 0,0 = all */
    ge_coo *coo = NULL;
    ge_status st = ge_bca_build(&out, &in, &cfg, &coo);
    (*env)->ReleaseLongArrayElements(env, outPtr, (jlong *)out.ptr, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, outIdx, (jint *)out.idx, JNI_ABORT);
    (*env)->ReleaseFloatArrayElements(env, outW, (jfloat *)out.weight, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, inPtr, (jlong *)in.ptr, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, inIdx, (jint *)in.idx, JNI_ABORT);
    (*env)->ReleaseFloatArrayElements(env, inW, (jfloat *)in.weight, JNI_ABORT);
    if (st != GE_OK) { throw_ge(env, st); return 0; }
    return (jlong)(intptr_t)coo;
}

/* long cooCount(long coo);  double cooMax(long coo);  void cooCopy(long coo, int[] I, int[] J, float[] X);  void cooDestroy(long coo) */
JNIEXPORT jlong JNICALL Java_org_uu_nl_embedding_hip_Native_cooCount(JNIEnv *env, jclass cls, jlong coo) {
    (void)env; (void)cls;
    int64_t n = 0;
    ge_coo_get((const ge_coo *)(intptr_t)coo, &n, NULL, NULL, NULL, NULL, NULL);
    return n;
}
JNIEXPORT jdouble JNICALL Java_org_uu_nl_embedding_hip_Native_cooMax(JNIEnv *env, jclass cls, jlong coo) {
    (void)env; (void)cls;
    double m = 0;
    ge_coo_get((const ge_coo *)(intptr_t)coo, NULL, NULL, NULL, NULL, NULL, &m);
    return m;
}
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_cooCopy(JNIEnv *env, jclass cls, jlong coo, jintArray I, jintArray J, jfloatArray X) {
    (void)cls;
    int64_t n = 0; const int32_t *pi, *pj; const float *px;
    ge_coo_get((const ge_coo *)(intptr_t)coo, &n, &pi, &pj, &px, NULL, NULL);
    (*env)->SetIntArrayRegion(env, I, 0, (jsize)n, (const jint *)pi);
    (*env)->SetIntArrayRegion(env, J, 0, (jsize)n, (const jint *)pj);
    (*env)->SetFloatArrayRegion(env, X, 0, (jsize)n, px);
}
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_cooDestroy(JNIEnv *env, jclass cls, jlong coo) {
    (void)env; (void)cls;
    ge_coo_destroy((ge_coo *)(intptr_t)coo);
}

/* long similarityPairs(String[] labels, int[] source, int[] sourceVertex, int[] target, int[] targetVertex,
 *                      int method, double threshold, int ngram, double smooth, double distance, int time, String pattern,
 *                      boolean upperTriangle, int device)
 * labels[k] = vertexLabels.getValueAsString(...) of every vertex in the group; source/target index into labels.
 * Replaces the CompareJob loop of Rdf2GrphConverter.convert (:127-186). */
JNIEXPORT jlong JNICALL Java_org_uu_nl_embedding_hip_Native_similarityPairs(
        JNIEnv *env, jclass cls, jobjectArray labels, jintArray source, jintArray sourceVertex, jintArray target, jintArray targetVertex,
        jint method, jdouble threshold, jint ngram, jdouble smooth, jdouble distance, jint time, jstring pattern,
        jboolean upperTriangle, jint device) {
    (void)cls;
    const jsize count = (*env)->GetArrayLength(env, labels);
    int64_t *offset = (int64_t *)malloc(sizeof(int64_t) * ((size_t)count + 1));
    offset[0] = 0;
    for (jsize k = 0; k < count; ++k) {
        jstring s = (jstring)(*env)->GetObjectArrayElement(env, labels, k);
        offset[k + 1] = offset[k] + (*env)->GetStringLength(env, s);          /* UTF-16 code units, as String.length() */
        (*env)->DeleteLocalRef(env, s);
    }
    uint16_t *units = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(offset[count] + 1));
    for (jsize k = 0; k < count; ++k) {
        jstring s = (jstring)(*env)->GetObjectArrayElement(env, labels, k);
        (*env)->GetStringRegion(env, s, 0, (jsize)(offset[k + 1] - offset[k]), (jchar *)(units + offset[k]));
        (*env)->DeleteLocalRef(env, s);
    }
    ge_strings table = {count, offset, units};
    ge_sim_cfg cfg;
    ge_sim_cfg_default(&cfg);
    cfg.method = method; cfg.threshold = threshold; cfg.ngram = ngram; cfg.smooth = smooth; cfg.distance = distance; cfg.time = time;
    const char *pat = pattern ? (*env)->GetStringUTFChars(env, pattern, NULL) : NULL;
    cfg.pattern = pat; cfg.upper_triangle = upperTriangle ? 1 : 0; cfg.device = device;
    const jsize ns = (*env)->GetArrayLength(env, source), nt = (*env)->GetArrayLength(env, target);
    jint *ps = (*env)->GetIntArrayElements(env, source, NULL), *psv = (*env)->GetIntArrayElements(env, sourceVertex, NULL);
    jint *pt = (*env)->GetIntArrayElements(env, target, NULL), *ptv = (*env)->GetIntArrayElements(env, targetVertex, NULL);
    ge_sim_pairs *res = NULL;
    ge_status st = ge_similarity_pairs(&table, (const int32_t *)ps, (const int32_t *)psv, ns, (const int32_t *)pt, (const int32_t *)ptv, nt, &cfg, &res);
    (*env)->ReleaseIntArrayElements(env, source, ps, JNI_ABORT); (*env)->ReleaseIntArrayElements(env, sourceVertex, psv, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, target, pt, JNI_ABORT); (*env)->ReleaseIntArrayElements(env, targetVertex, ptv, JNI_ABORT);
    if (pat) (*env)->ReleaseStringUTFChars(env, pattern, pat);
    free(offset); free(units);
    if (st != GE_OK) { throw_ge(env, st); return 0; }
    return (jlong)(intptr_t)res;
}

/* long pairsCount(long pairs);  void pairsCopy(long pairs, int[] sourcePos, int[] targetPos, float[] similarity);  void pairsDestroy(long pairs) */
JNIEXPORT jlong JNICALL Java_org_uu_nl_embedding_hip_Native_pairsCount(JNIEnv *env, jclass cls, jlong pairs) {
    (void)env; (void)cls;
    int64_t n = 0;
    ge_sim_pairs_get((const ge_sim_pairs *)(intptr_t)pairs, &n, NULL, NULL, NULL);
    return n;
}
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_pairsCopy(JNIEnv *env, jclass cls, jlong pairs, jintArray sourcePos, jintArray targetPos, jfloatArray similarity) {
    (void)cls;
    int64_t n = 0; const int32_t *pi, *pj; const float *ps;
    ge_sim_pairs_get((const ge_sim_pairs *)(intptr_t)pairs, &n, &pi, &pj, &ps);
    (*env)->SetIntArrayRegion(env, sourcePos, 0, (jsize)n, (const jint *)pi);
    (*env)->SetIntArrayRegion(env, targetPos, 0, (jsize)n, (const jint *)pj);
    (*env)->SetFloatArrayRegion(env, similarity, 0, (jsize)n, ps);
}
JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_pairsDestroy(JNIEnv *env, jclass cls, jlong pairs) {
    (void)env; (void)cls;
    ge_sim_pairs_destroy((ge_sim_pairs *)(intptr_t)pairs);
}
