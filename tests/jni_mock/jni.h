/*
 * jni.h — TEST-ONLY stand-in for the JDK header (the build image has no JDK).  It declares just enough of the
 * JNI surface for integration/jni/surge_replay_jni.c to compile unchanged: the scalar typedefs, the export
 * macros and a function table with the four calls the shim makes.  The table layout is NOT the JVM's; the
 * shim is source-compatible with the real header because it only ever calls (*env)->Fn(env, ...).
 * Used by tests/jni_mock/jni_harness.c, which plays the JVM's part with a fake JNIEnv.
 */
#ifndef SURGE_TEST_JNI_MOCK_H
#define SURGE_TEST_JNI_MOCK_H

#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

typedef int32_t jint;
typedef int64_t jlong;
typedef void* jobject;
typedef jobject jclass;
typedef jobject jthrowable;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;

struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv* env, const char* name);
  jint (*ThrowNew)(JNIEnv* env, jclass clazz, const char* msg);
  void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
  jlong (*GetDirectBufferCapacity)(JNIEnv* env, jobject buf);
};

#endif
