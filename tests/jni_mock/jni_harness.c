/*
 * jni_harness.c — TEST-ONLY: drives the JNI shim (integration/jni/surge_replay_jni.c, compiled unchanged against
 * the stand-in jni.h next to this file) the way surge.replay.gpu.NativeReplay would, with a fake JNIEnv:
 * a "direct ByteBuffer" is a struct holding an address, ThrowNew records the pending exception.
 * Known answers: PersistentActorSpec.scala:134-168, 466-493 ((3,3) + two increments = (5,5)),
 * scala MurmurHash3.stringHash("") / ("a") (tests/test_oracle_kat.py).
 * Exit code 0 = all good, 2 = no GPU (create threw IOException "... no CPU fallback").
 */
#include <stdio.h>
#include <string.h>
#include <jni.h>
#include "surge_replay.h"

jlong Java_surge_replay_gpu_NativeReplay_create(JNIEnv*, jclass, jobject, jint);
void Java_surge_replay_gpu_NativeReplay_destroy(JNIEnv*, jclass, jlong);
jint Java_surge_replay_gpu_NativeReplay_loadCsr(JNIEnv*, jclass, jlong, jobject, jlong, jobject, jlong, jobject);
jint Java_surge_replay_gpu_NativeReplay_fold(JNIEnv*, jclass, jlong, jint);
jint Java_surge_replay_gpu_NativeReplay_appendFold(JNIEnv*, jclass, jlong, jobject, jobject, jlong, jobject, jlong);
jint Java_surge_replay_gpu_NativeReplay_snapshot(JNIEnv*, jclass, jlong, jobject, jobject);
jint Java_surge_replay_gpu_NativeReplay_get(JNIEnv*, jclass, jlong, jlong, jobject);
jint Java_surge_replay_gpu_NativeReplay_partitionHash(JNIEnv*, jclass, jobject, jobject, jlong, jint, jobject);
jint Java_surge_replay_gpu_NativeReplay_partitionHashUpToColon(JNIEnv*, jclass, jobject, jobject, jlong, jint, jobject);

typedef struct { void* address; } fake_direct_buffer;

static char pending_class[64];
static char pending_msg[512];
static int n_thrown = 0;

static jclass fake_FindClass(JNIEnv* env, const char* name) {
  (void)env;
  snprintf(pending_class, sizeof(pending_class), "%s", name);
  return (jclass)pending_class;
}
static jint fake_ThrowNew(JNIEnv* env, jclass clazz, const char* msg) {
  (void)env; (void)clazz;
  snprintf(pending_msg, sizeof(pending_msg), "%s", msg);
  ++n_thrown;
  return 0;
}
static void* fake_GetDirectBufferAddress(JNIEnv* env, jobject buf) {
  (void)env;
  return ((fake_direct_buffer*)buf)->address;
}

static int fails = 0;
static void check(int ok, const char* what) {
  printf("%s  %s\n", ok ? "PASS" : "FAIL", what);
  if (!ok) ++fails;
}

int main(void) {
  const struct JNINativeInterface_ table = {fake_FindClass, fake_ThrowNew, fake_GetDirectBufferAddress};
  JNIEnv env_obj = &table;
  JNIEnv* env = &env_obj;

  /* partitionHash runs on the host: works with or without a GPU */
  {
    const uint16_t utf16[] = {'a', 'a', ':', '7'};
    const int64_t off[] = {0, 0, 1, 4};
    int32_t part[3] = {-1, -1, -1};
    fake_direct_buffer b_utf16 = {(void*)utf16}, b_off = {(void*)off}, b_part = {part};
    jint rc = Java_surge_replay_gpu_NativeReplay_partitionHash(env, NULL, &b_utf16, &b_off, 3, 1000003, &b_part);
    check(rc == 0 && part[0] == 926349 && part[1] == 229102 && part[2] == 324673, "partitionHash = partitionForKey of the whole string (\"\", \"a\", \"a:7\")");
    rc = Java_surge_replay_gpu_NativeReplay_partitionHashUpToColon(env, NULL, &b_utf16, &b_off, 3, 1000003, &b_part);
    check(rc == 0 && part[0] == 926349 && part[1] == 229102 && part[2] == 229102, "partitionHashUpToColon: key cut at ':' (PartitionStringUpToColon)");
  }

  const jlong h = Java_surge_replay_gpu_NativeReplay_create(env, NULL, NULL, 0);
  if (h == 0) {
    check(n_thrown == 1 && strcmp(pending_class, "java/io/IOException") == 0, "create without a GPU throws IOException");
    printf("no GPU: %s\n", pending_msg);
    return fails ? 1 : 2;
  }

  /* a status < 0 becomes a pending IOException carrying the engine's message */
  n_thrown = 0;
  check(Java_surge_replay_gpu_NativeReplay_fold(env, NULL, h, 0) == SURGE_E_STATE && n_thrown == 1 &&
            strstr(pending_msg, "fold before") != NULL, "fold before loadCsr -> IOException(\"fold before ...\")");

  surge_state64 init[2], out[2], one;
  surge_event16 ev[3];
  int64_t seg_off[3] = {0, 2, 3};
  uint8_t present[2] = {9, 9};
  memset(init, 0, sizeof(init)); memset(ev, 0, sizeof(ev)); memset(out, 0, sizeof(out));
  init[0].count = 3; init[0].version = 3; init[0].min_arg = 0x7fffffff; init[0].max_arg = (int32_t)0x80000000;
  init[0].flags = SURGE_STATE_PRESENT;
  ev[0].type = SURGE_EVT_INC; ev[0].seq = 4; ev[0].p.i.arg = 1;
  ev[1].type = SURGE_EVT_INC; ev[1].seq = 5; ev[1].p.i.arg = 1;
  ev[2].type = SURGE_EVT_SET_BALANCE; ev[2].p.value = 5.0; /* update before create: stays None */
  fake_direct_buffer b_off = {seg_off}, b_ev = {ev}, b_init = {init}, b_out = {out}, b_present = {present}, b_one = {&one};
  n_thrown = 0;
  check(Java_surge_replay_gpu_NativeReplay_loadCsr(env, NULL, h, &b_off, 2, &b_ev, 3, &b_init) == 0 &&
            Java_surge_replay_gpu_NativeReplay_fold(env, NULL, h, 0) == 0 &&
            Java_surge_replay_gpu_NativeReplay_snapshot(env, NULL, h, &b_out, &b_present) == 0 && n_thrown == 0,
        "loadCsr / fold / snapshot through direct buffers");
  check(out[0].count == 5 && out[0].version == 5 && present[0] == 1 && present[1] == 0, "(3,3) + two increments = (5,5); orphan update stays None");
  check(Java_surge_replay_gpu_NativeReplay_get(env, NULL, h, 0, &b_one) == 1 && one.count == 5 &&
            Java_surge_replay_gpu_NativeReplay_get(env, NULL, h, 1, &b_one) == 0, "get: 1 = Some(state64 filled), 0 = None");
  n_thrown = 0;
  check(Java_surge_replay_gpu_NativeReplay_get(env, NULL, h, 7, &b_one) == -1 && n_thrown == 1, "get out of range -> IOException, -1");

  /* micro-batch: one more increment on aggregate 0 */
  {
    int64_t group_agg[1] = {0}, group_off[2] = {0, 1};
    surge_event16 e;
    memset(&e, 0, sizeof(e));
    e.type = SURGE_EVT_INC; e.seq = 6; e.p.i.arg = 1;
    fake_direct_buffer b_ga = {group_agg}, b_go = {group_off}, b_e = {&e};
    check(Java_surge_replay_gpu_NativeReplay_appendFold(env, NULL, h, &b_ga, &b_go, 1, &b_e, 1) == 0 &&
              Java_surge_replay_gpu_NativeReplay_get(env, NULL, h, 0, &b_one) == 1 && one.count == 6 && one.version == 6,
          "appendFold onto the resident state: (6,6)");
  }
  Java_surge_replay_gpu_NativeReplay_destroy(env, NULL, h);
  printf("%s\n", fails ? "FAILED" : "ALL PASS");
  return fails ? 1 : 0;
}
