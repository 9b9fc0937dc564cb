/* Minimal stand-in for <jni.h>: ONLY for the syntax / ABI-drift check of jni/presto_amd_jni.c in an image without a JDK
 * (tests/test_lib_cpu.py compiles the shim against it with -fsyntax-only).  It declares the JNI types and the members of
 * JNINativeInterface_ the shim uses, with the signatures of the JNI specification; a real build uses the JDK's header. */
#ifndef PRESTO_AMD_STUB_JNI_H
#define PRESTO_AMD_STUB_JNI_H
#include <stdint.h>
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_OK 0
#define JNI_VERSION_1_8 0x00010008
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef double jdouble;
typedef jint jsize;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;
typedef jarray jbyteArray;
typedef jarray jobjectArray;
struct _jmethodID;
typedef struct _jmethodID* jmethodID;
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNIInvokeInterface_;
typedef const struct JNIInvokeInterface_* JavaVM;
struct JNIInvokeInterface_ {
    jint (*GetEnv)(JavaVM*, void**, jint);
};
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv*, const char*);
    jmethodID (*GetMethodID)(JNIEnv*, jclass, const char*, const char*);
    jobject (*NewObject)(JNIEnv*, jclass, jmethodID, ...);
    jint (*Throw)(JNIEnv*, jthrowable);
    jstring (*NewStringUTF)(JNIEnv*, const char*);
    jsize (*GetArrayLength)(JNIEnv*, jarray);
    jobject (*GetObjectArrayElement)(JNIEnv*, jobjectArray, jsize);
    jint* (*GetIntArrayElements)(JNIEnv*, jintArray, jboolean*);
    jlong* (*GetLongArrayElements)(JNIEnv*, jlongArray, jboolean*);
    jdouble* (*GetDoubleArrayElements)(JNIEnv*, jdoubleArray, jboolean*);
    jbyte* (*GetByteArrayElements)(JNIEnv*, jbyteArray, jboolean*);
    void (*ReleaseIntArrayElements)(JNIEnv*, jintArray, jint*, jint);
    void (*ReleaseLongArrayElements)(JNIEnv*, jlongArray, jlong*, jint);
    void (*ReleaseDoubleArrayElements)(JNIEnv*, jdoubleArray, jdouble*, jint);
    void (*ReleaseByteArrayElements)(JNIEnv*, jbyteArray, jbyte*, jint);
    void (*SetLongArrayRegion)(JNIEnv*, jlongArray, jsize, jsize, const jlong*);
    void (*SetIntArrayRegion)(JNIEnv*, jintArray, jsize, jsize, const jint*);
    jlongArray (*NewLongArray)(JNIEnv*, jsize);
    jobject (*NewDirectByteBuffer)(JNIEnv*, void*, jlong);
    void* (*GetDirectBufferAddress)(JNIEnv*, jobject);
    jlong (*GetDirectBufferCapacity)(JNIEnv*, jobject);
    jboolean (*ExceptionCheck)(JNIEnv*);
    jint (*GetJavaVM)(JNIEnv*, JavaVM**);
    jclass (*GetObjectClass)(JNIEnv*, jobject);
    jobject (*NewGlobalRef)(JNIEnv*, jobject);
    void (*DeleteGlobalRef)(JNIEnv*, jobject);
    jobject (*CallObjectMethod)(JNIEnv*, jobject, jmethodID, ...);
    void (*CallVoidMethod)(JNIEnv*, jobject, jmethodID, ...);
};
#endif
