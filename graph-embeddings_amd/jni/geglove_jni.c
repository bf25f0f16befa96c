/*
 * geglove_jni.c -- JNI glue between the reference's Java host and libgeglove.so.
 * SOURCE ONLY in this repository: the build image has no JDK (no jni.h, no javac), so this file is
 * compiled where one exists:   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux \
 *                                  -I../../include geglove_jni.c -L../lib -lgeglove -o libgeglove_jni.so
 * Java side: org.uu.nl.embedding.hip.Native.  Every entry point maps a non-zero
 * ge_status to a RuntimeException carrying ge_last_error(); no global references are kept; arrays are
 * pinned only for the duration of the native call (Get/ReleasePrimitiveArrayCritical).
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include "geglove.h"

static void throw_ge(JNIEnv *env, ge_status st) {
    jclass ex = (*env)->FindClass(env, st == GE_ERR_ARG ? "java/lang/IllegalArgumentException" : "java/lang/RuntimeException");
    if (ex) (*env)->ThrowNew(env, ex, ge_last_error());
}

/* long gloveCreate(int V, int D, int[] I, int[] J, float[] X, double xmax, int cost, long seed, int threads,
 *                  int mode, int shuffle, int device)                                                         */
JNIEXPORT jlong JNICALL Java_org_uu_nl_embedding_hip_Native_gloveCreate(
        JNIEnv *env, jclass cls, jint V, jint D, jintArray I, jintArray J, jfloatArray X, jdouble xmax,
        jint cost, jlong seed, jint threads, jint mode, jint shuffle, jint device) {
    (void)cls;
    ge_glove_cfg cfg;
    ge_glove_cfg_default(&cfg);
    cfg.vocab_size = V; cfg.dim = D; cfg.nnz = (*env)->GetArrayLength(env, I);
    cfg.cost = cost; cfg.xmax = xmax; cfg.seed = seed; cfg.threads = threads;
    cfg.mode = mode; cfg.shuffle = shuffle; cfg.device = device;
    jint *pi = (*env)->GetPrimitiveArrayCritical(env, I, NULL);
    jint *pj = (*env)->GetPrimitiveArrayCritical(env, J, NULL);
    jfloat *px = (*env)->GetPrimitiveArrayCritical(env, X, NULL);
    ge_glove *h = NULL;
    ge_status st = (pi && pj && px) ? ge_glove_create(&cfg, (const int32_t *)pi, (const int32_t *)pj, px, &h) : GE_ERR_OOM;
    if (px) (*env)->ReleasePrimitiveArrayCritical(env, X, px, JNI_ABORT);
    if (pj) (*env)->ReleasePrimitiveArrayCritical(env, J, pj, JNI_ABORT);
    if (pi) (*env)->ReleasePrimitiveArrayCritical(env, I, pi, JNI_ABORT);
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
    jdouble *p = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
    ge_status st = p ? ge_glove_extract_f64((ge_glove *)(intptr_t)handle, p) : GE_ERR_OOM;
    if (p) (*env)->ReleasePrimitiveArrayCritical(env, out, p, 0);
    if (st != GE_OK) throw_ge(env, st);
}

JNIEXPORT void JNICALL Java_org_uu_nl_embedding_hip_Native_gloveDestroy(JNIEnv *env, jclass cls, jlong handle) {
    (void)env; (void)cls;
    ge_glove_destroy((ge_glove *)(intptr_t)handle);
}

/* long bcaBuild(int V, long[] outPtr, int[] outIdx, float[] outW, long[] inPtr, int[] inIdx, float[] inW,
 *               double alpha, double epsilon, boolean directed, int normalize, int device)                   */
JNIEXPORT jlong JNICALL Java_org_uu_nl_embedding_hip_Native_bcaBuild(
        JNIEnv *env, jclass cls, jint V, jlongArray outPtr, jintArray outIdx, jfloatArray outW,
        jlongArray inPtr, jintArray inIdx, jfloatArray inW, jdouble alpha, jdouble epsilon,
        jboolean directed, jint normalize, jint device) {
    (void)cls;
    ge_csr out, in;
    out.num_vertices = in.num_vertices = V;
    out.ptr = (const int64_t *)(*env)->GetLongArrayElements(env, outPtr, NULL);
    out.idx = (const int32_t *)(*env)->GetIntArrayElements(env, outIdx, NULL);
    out.weight = (*env)->GetFloatArrayElements(env, outW, NULL);
    in.ptr = (const int64_t *)(*env)->GetLongArrayElements(env, inPtr, NULL);
    in.idx = (const int32_t *)(*env)->GetIntArrayElements(env, inIdx, NULL);
    in.weight = (*env)->GetFloatArrayElements(env, inW, NULL);
    ge_bca_cfg cfg = {alpha, epsilon, directed ? 1 : 0, normalize, device, 0, 0};
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
