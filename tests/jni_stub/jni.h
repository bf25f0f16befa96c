/* Minimal stand-in for <jni.h>: the types and the JNIEnv function-table members that graph-embeddings_amd/jni/geglove_jni.c
 * uses, with the signatures of the JNI specification (Java SE 8, chapter 4).  TEST INFRASTRUCTURE: the build image has no JDK;
 * tests/test_capi_and_host.py compiles the glue against this header (-fsyntax-only -Wall -Werror) so that syntax and
 * signature rot is caught.  It is never shipped and nothing links against it. */
#ifndef GE_TEST_JNI_STUB_H
#define GE_TEST_JNI_STUB_H
#include <stdint.h>
typedef int32_t jint; typedef int64_t jlong; typedef int8_t jbyte; typedef uint8_t jboolean; typedef uint16_t jchar;
typedef float jfloat; typedef double jdouble; typedef jint jsize;
typedef struct _jobject *jobject;
typedef jobject jclass, jstring, jarray, jobjectArray, jintArray, jlongArray, jfloatArray, jdoubleArray, jbyteArray, jthrowable;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_FALSE 0
#define JNI_TRUE 1
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *, const char *);
    jint (*ThrowNew)(JNIEnv *, jclass, const char *);
    jboolean (*ExceptionCheck)(JNIEnv *);
    void (*DeleteLocalRef)(JNIEnv *, jobject);
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    jobject (*GetObjectArrayElement)(JNIEnv *, jobjectArray, jsize);
    jsize (*GetStringLength)(JNIEnv *, jstring);
    void (*GetStringRegion)(JNIEnv *, jstring, jsize, jsize, jchar *);
    const char *(*GetStringUTFChars)(JNIEnv *, jstring, jboolean *);
    void (*ReleaseStringUTFChars)(JNIEnv *, jstring, const char *);
    jint *(*GetIntArrayElements)(JNIEnv *, jintArray, jboolean *);
    jlong *(*GetLongArrayElements)(JNIEnv *, jlongArray, jboolean *);
    jfloat *(*GetFloatArrayElements)(JNIEnv *, jfloatArray, jboolean *);
    jdouble *(*GetDoubleArrayElements)(JNIEnv *, jdoubleArray, jboolean *);
    void (*ReleaseIntArrayElements)(JNIEnv *, jintArray, jint *, jint);
    void (*ReleaseLongArrayElements)(JNIEnv *, jlongArray, jlong *, jint);
    void (*ReleaseFloatArrayElements)(JNIEnv *, jfloatArray, jfloat *, jint);
    void (*ReleaseDoubleArrayElements)(JNIEnv *, jdoubleArray, jdouble *, jint);
    void (*SetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, const jint *);
    void (*SetFloatArrayRegion)(JNIEnv *, jfloatArray, jsize, jsize, const jfloat *);
    void (*GetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, jbyte *);
    void (*SetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, const jbyte *);
    jbyteArray (*NewByteArray)(JNIEnv *, jsize);
};
#endif
