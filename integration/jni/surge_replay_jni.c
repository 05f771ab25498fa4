/*
 * surge_replay_jni.c — thin JNI shim over include/surge_replay.h.
 *
 * The build image has no JDK (no jni.h): for a JVM this file is compiled where JAVA_HOME is set (below); in
 * this repository it is compiled unchanged against the stand-in header tests/jni_mock/jni.h and driven by a
 * fake JNIEnv (tests/jni_mock/jni_harness.c, tests/test_abi.py).
 *
 *   gcc -shared -fPIC -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -I../../include \
 *       surge_replay_jni.c -L../../surge_amd -lsurge_replay -o libsurge_replay_jni.so
 *
 * Binds `surge.replay.gpu.NativeReplay` (integration/scala/NativeReplay.scala).  Handles are jlong,
 * bulk data are direct java.nio.ByteBuffers (no array copies), a negative status becomes an
 * IOException carrying surge_replay_last_error().
 */
#include <jni.h>
#include <stddef.h>
#include <stdint.h>
#include "surge_replay.h"

#define H(h) ((surge_replay_handle*)(intptr_t)(h))

static jint check(JNIEnv* env, surge_replay_handle* h, int32_t rc) {
  if (rc != SURGE_OK) {
    jclass ex = (*env)->FindClass(env, "java/io/IOException");
    const char* msg = surge_replay_last_error(NULL); /* the calling thread's own failure (get() runs on a 32-thread pool) */
    (void)h;
    if (ex) (*env)->ThrowNew(env, ex, msg ? msg : "surge_replay call failed");
  }
  return rc;
}

static void* addr(JNIEnv* env, jobject buf) { return buf ? (*env)->GetDirectBufferAddress(env, buf) : NULL; }

JNIEXPORT jlong JNICALL Java_surge_replay_gpu_NativeReplay_create(JNIEnv* env, jclass c, jobject schemaBuf, jint device) {
  surge_replay_handle* h = NULL;
  surge_replay_schema sc;
  (void)c;
  if (schemaBuf) {
    sc = *(const surge_replay_schema*)addr(env, schemaBuf);
  } else {
    surge_replay_default_schema(&sc);
  }
  check(env, NULL, surge_replay_create(&sc, device, &h));
  return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL Java_surge_replay_gpu_NativeReplay_destroy(JNIEnv* env, jclass c, jlong h) {
  (void)env; (void)c;
  surge_replay_destroy(H(h));
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_loadCsr(JNIEnv* env, jclass c, jlong h, jobject segOff,
                                                                    jlong nAgg, jobject events, jlong nEvents,
                                                                    jobject initState) {
  (void)c;
  return check(env, H(h), surge_replay_load_csr(H(h), (const int64_t*)addr(env, segOff), nAgg, addr(env, events),
                                                 nEvents, addr(env, initState)));
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_fold(JNIEnv* env, jclass c, jlong h, jint algo) {
  (void)c;
  return check(env, H(h), surge_replay_fold(H(h), algo));
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_appendFold(JNIEnv* env, jclass c, jlong h, jobject groupAgg,
                                                                       jobject groupOff, jlong nGroups, jobject events,
                                                                       jlong nEvents) {
  (void)c;
  return check(env, H(h), surge_replay_append_fold(H(h), (const int64_t*)addr(env, groupAgg),
                                                    (const int64_t*)addr(env, groupOff), nGroups, addr(env, events),
                                                    nEvents));
}

/* Publishes the host mirror that concurrent get() calls then read under a shared lock; states may be null. */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_snapshot(JNIEnv* env, jclass c, jlong h, jobject states,
                                                                     jobject present) {
  (void)c;
  return check(env, H(h), surge_replay_snapshot(H(h), addr(env, states), (uint8_t*)addr(env, present)));
}

/* Returns 1 when the aggregate is present (state64 filled), 0 for None. */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_get(JNIEnv* env, jclass c, jlong h, jlong aggIdx,
                                                                jobject state64) {
  uint8_t present = 0;
  (void)c;
  if (check(env, H(h), surge_replay_get(H(h), aggIdx, addr(env, state64), &present)) != SURGE_OK) return -1;
  return present;
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_partitionHash(JNIEnv* env, jclass c, jobject utf16,
                                                                          jobject strOff, jlong n, jint nPartitions,
                                                                          jobject partOut) {
  (void)c;
  return check(env, NULL, surge_replay_partition_hash((const uint16_t*)addr(env, utf16), (const int64_t*)addr(env, strOff),
                                                       n, nPartitions, (int32_t*)addr(env, partOut)));
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_partitionHashUpToColon(JNIEnv* env, jclass c, jobject utf16,
                                                                                   jobject strOff, jlong n, jint nPartitions,
                                                                                   jobject partOut) {
  (void)c;
  return check(env, NULL, surge_replay_partition_hash_up_to_colon((const uint16_t*)addr(env, utf16), (const int64_t*)addr(env, strOff),
                                                                   n, nPartitions, (int32_t*)addr(env, partOut)));
}
