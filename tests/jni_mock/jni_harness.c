/*
 * jni_harness.c — TEST-ONLY: drives EVERY Java_… export of the JNI shim (integration/jni/surge_replay_jni.c, compiled
 * unchanged against the stand-in jni.h next to this file) the way surge.replay.gpu.NativeReplay would, with a fake
 * JNIEnv: a "direct ByteBuffer" is a struct holding an address and a capacity (address NULL = a heap buffer), ThrowNew
 * records the pending exception.
 * Known answers: PersistentActorSpec.scala:134-168, 466-493 ((3,3) + two increments = (5,5)), the throwing event of
 * :431-464, scala MurmurHash3.stringHash("") / ("a") (tests/test_oracle_kat.py).
 * Exit code 0 = all good, 2 = no GPU (create threw IOException "... no CPU fallback").
 */
#include <stdio.h>
#include <string.h>
#include <jni.h>
#include "surge_replay.h"
#include "surge_ingest.h"

jlong Java_surge_replay_gpu_NativeReplay_create(JNIEnv*, jclass, jobject, jint);
void Java_surge_replay_gpu_NativeReplay_destroy(JNIEnv*, jclass, jlong);
jint Java_surge_replay_gpu_NativeReplay_loadCsr(JNIEnv*, jclass, jlong, jobject, jlong, jobject, jlong, jobject);
jint Java_surge_replay_gpu_NativeReplay_fold(JNIEnv*, jclass, jlong, jint);
jint Java_surge_replay_gpu_NativeReplay_prepare(JNIEnv*, jclass, jlong, jint, jobject);
jint Java_surge_replay_gpu_NativeReplay_appendFold(JNIEnv*, jclass, jlong, jobject, jobject, jlong, jobject, jlong);
jint Java_surge_replay_gpu_NativeReplay_appendEvents(JNIEnv*, jclass, jlong, jobject, jobject, jlong);
jint Java_surge_replay_gpu_NativeReplay_grow(JNIEnv*, jclass, jlong, jlong);
jint Java_surge_replay_gpu_NativeReplay_snapshot(JNIEnv*, jclass, jlong, jlong, jobject, jobject);
jint Java_surge_replay_gpu_NativeReplay_get(JNIEnv*, jclass, jlong, jlong, jobject);
jint Java_surge_replay_gpu_NativeReplay_gather(JNIEnv*, jclass, jlong, jobject, jlong, jobject);
jint Java_surge_replay_gpu_NativeReplay_partitionHash(JNIEnv*, jclass, jobject, jobject, jlong, jint, jobject);
jint Java_surge_replay_gpu_NativeReplay_partitionHashUpToColon(JNIEnv*, jclass, jobject, jobject, jlong, jint, jobject);
jint Java_surge_replay_gpu_NativeReplay_commUniqueId(JNIEnv*, jclass, jobject);
jint Java_surge_replay_gpu_NativeReplay_commInit(JNIEnv*, jclass, jlong, jint, jint, jobject);
jint Java_surge_replay_gpu_NativeReplay_commDestroy(JNIEnv*, jclass, jlong);
jlong Java_surge_replay_gpu_NativeReplay_commCounts(JNIEnv*, jclass, jlong, jlong, jint, jobject);
jint Java_surge_replay_gpu_NativeReplay_allgatherSnapshot(JNIEnv*, jclass, jlong, jlong, jint, jint);
jint Java_surge_replay_gpu_NativeReplay_allgatherGroup(JNIEnv*, jclass, jobject, jint, jint);
jint Java_surge_replay_gpu_NativeReplay_gatheredRead(JNIEnv*, jclass, jlong, jint, jint, jlong, jlong, jobject);
jlong Java_surge_replay_gpu_NativeReplay_decoderCreate(JNIEnv*, jclass, jobject, jint);
void Java_surge_replay_gpu_NativeReplay_decoderDestroy(JNIEnv*, jclass, jlong);
jint Java_surge_replay_gpu_NativeReplay_decoderPushRecords(JNIEnv*, jclass, jlong, jobject, jobject, jobject, jobject, jobject, jlong);
jint Java_surge_replay_gpu_NativeReplay_appendDecoded(JNIEnv*, jclass, jlong, jlong, jobject);
jint Java_surge_replay_gpu_NativeReplay_stageDecoded(JNIEnv*, jclass, jlong, jlong, jobject);
jint Java_surge_replay_gpu_NativeReplay_packStaged(JNIEnv*, jclass, jlong, jlong);
jint Java_surge_replay_gpu_NativeReplay_decoderKeys(JNIEnv*, jclass, jlong, jobject, jobject, jobject);

typedef struct { void* address; jlong capacity; } fake_direct_buffer;
#define DB(ptr, bytes) { (void*)(ptr), (jlong)(bytes) }

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
static jlong fake_GetDirectBufferCapacity(JNIEnv* env, jobject buf) {
  (void)env;
  return ((fake_direct_buffer*)buf)->address ? ((fake_direct_buffer*)buf)->capacity : -1;
}

static int fails = 0;
static void check(int ok, const char* what) {
  printf("%s  %s\n", ok ? "PASS" : "FAIL", what);
  if (!ok) ++fails;
}

int main(void) {
  const struct JNINativeInterface_ table = {fake_FindClass, fake_ThrowNew, fake_GetDirectBufferAddress, fake_GetDirectBufferCapacity};
  JNIEnv env_obj = &table;
  JNIEnv* env = &env_obj;

  /* partitionHash runs on the host: works with or without a GPU */
  {
    const uint16_t utf16[] = {'a', 'a', ':', '7'};
    const int64_t off[] = {0, 0, 1, 4};
    int32_t part[3] = {-1, -1, -1};
    fake_direct_buffer b_utf16 = DB(utf16, sizeof(utf16)), b_off = DB(off, sizeof(off)), b_part = DB(part, sizeof(part));
    jint rc = Java_surge_replay_gpu_NativeReplay_partitionHash(env, NULL, &b_utf16, &b_off, 3, 1000003, &b_part);
    check(rc == 0 && part[0] == 926349 && part[1] == 229102 && part[2] == 324673, "partitionHash = partitionForKey of the whole string (\"\", \"a\", \"a:7\")");
    rc = Java_surge_replay_gpu_NativeReplay_partitionHashUpToColon(env, NULL, &b_utf16, &b_off, 3, 1000003, &b_part);
    check(rc == 0 && part[0] == 926349 && part[1] == 229102 && part[2] == 229102, "partitionHashUpToColon: key cut at ':' (PartitionStringUpToColon)");
    /* a short or a heap (non-direct) buffer is refused before the C ABI sees it */
    fake_direct_buffer b_short = DB(part, 8), b_heap = DB(NULL, 0);
    n_thrown = 0;
    rc = Java_surge_replay_gpu_NativeReplay_partitionHash(env, NULL, &b_utf16, &b_off, 3, 1000003, &b_short);
    check(rc == SURGE_E_INVALID && n_thrown == 1 && strcmp(pending_class, "java/lang/IllegalArgumentException") == 0,
          "a too-small direct buffer -> IllegalArgumentException, no native access");
    n_thrown = 0;
    rc = Java_surge_replay_gpu_NativeReplay_partitionHash(env, NULL, &b_heap, &b_off, 3, 1000003, &b_part);
    check(rc == SURGE_E_INVALID && n_thrown == 1, "a heap (non-direct) buffer -> IllegalArgumentException");
    /* sizes and offsets the JVM hands over are checked before anything is derived from them */
    n_thrown = 0;
    rc = Java_surge_replay_gpu_NativeReplay_partitionHash(env, NULL, &b_utf16, &b_off, -1, 1000003, &b_part);
    check(rc == SURGE_E_INVALID, "a negative count is refused before any buffer size is computed from it");
    {
      const int64_t bad_off[] = {0, 3, 1, 4};
      fake_direct_buffer b_bad = DB(bad_off, sizeof(bad_off));
      n_thrown = 0;
      rc = Java_surge_replay_gpu_NativeReplay_partitionHashUpToColon(env, NULL, &b_utf16, &b_bad, 3, 1000003, &b_part);
      check(rc == SURGE_E_INVALID && n_thrown == 1, "decreasing string offsets -> IllegalArgumentException");
    }
  }

  n_thrown = 0;
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

  surge_state64 init[3], out[3], one;
  surge_event16 ev[4];
  int64_t seg_off[4] = {0, 2, 3, 4};
  uint8_t present[3] = {9, 9, 9};
  memset(init, 0, sizeof(init)); memset(ev, 0, sizeof(ev)); memset(out, 0, sizeof(out));
  init[0].count = 3; init[0].version = 3; init[0].min_arg = 0x7fffffff; init[0].max_arg = (int32_t)0x80000000;
  init[0].flags = SURGE_STATE_PRESENT;
  ev[0].type = SURGE_EVT_INC; ev[0].seq = 4; ev[0].p.i.arg = 1;
  ev[1].type = SURGE_EVT_INC; ev[1].seq = 5; ev[1].p.i.arg = 1;
  ev[2].type = SURGE_EVT_SET_BALANCE; ev[2].p.value = 5.0; /* update before create: stays None */
  ev[3].type = SURGE_EVT_THROW;                            /* ExceptionThrowingEvent: aggregate 2 is poisoned */
  fake_direct_buffer b_off = DB(seg_off, sizeof(seg_off)), b_ev = DB(ev, sizeof(ev)), b_init = DB(init, sizeof(init)),
                     b_out = DB(out, sizeof(out)), b_present = DB(present, sizeof(present)), b_one = DB(&one, sizeof(one));
  n_thrown = 0;
  check(Java_surge_replay_gpu_NativeReplay_loadCsr(env, NULL, h, &b_off, 3, &b_ev, 4, &b_init) == 0 &&
            Java_surge_replay_gpu_NativeReplay_fold(env, NULL, h, 0) == 0 &&
            Java_surge_replay_gpu_NativeReplay_snapshot(env, NULL, h, 3, &b_out, &b_present) == 0 && n_thrown == 0,
        "loadCsr / fold / snapshot through direct buffers");
  check(out[0].count == 5 && out[0].version == 5 && present[0] == 1 && present[1] == 0, "(3,3) + two increments = (5,5); orphan update stays None");
  {
    /* the tile-major path through JNI: prepare(TILED) reports its one-off cost, fold(TILED) gives the same states */
    unsigned char lay[24];
    fake_direct_buffer b_lay = DB(lay, sizeof lay);
    surge_state64 out2[3];
    uint8_t present2[3];
    fake_direct_buffer b_out2 = DB(out2, sizeof out2), b_present2 = DB(present2, sizeof present2);
    double relayout_ms = -1.0;
    int64_t tiled_bytes = 0;
    n_thrown = 0;
    const int okp = Java_surge_replay_gpu_NativeReplay_prepare(env, NULL, h, SURGE_ALGO_TILED, &b_lay) == 0 &&
                    Java_surge_replay_gpu_NativeReplay_fold(env, NULL, h, SURGE_ALGO_TILED) == 0 &&
                    Java_surge_replay_gpu_NativeReplay_snapshot(env, NULL, h, 3, &b_out2, &b_present2) == 0 && n_thrown == 0;
    memcpy(&relayout_ms, lay + 8, 8);
    memcpy(&tiled_bytes, lay + 16, 8);
    check(okp && memcmp(out, out2, sizeof out) == 0 && relayout_ms > 0.0 && tiled_bytes > 0 && tiled_bytes % 8192 == 0,
          "prepare(TILED) + fold(TILED): same states, layout cost reported");
  }
  check(Java_surge_replay_gpu_NativeReplay_get(env, NULL, h, 0, &b_one) == 1 && one.count == 5 &&
            Java_surge_replay_gpu_NativeReplay_get(env, NULL, h, 1, &b_one) == 0, "get: 1 = Some(state64 filled), 0 = None");
  check(Java_surge_replay_gpu_NativeReplay_get(env, NULL, h, 2, &b_one) == 2, "get: 2 = POISONED (replay hit a throwing event): never served as a state");
  n_thrown = 0;
  check(Java_surge_replay_gpu_NativeReplay_get(env, NULL, h, 7, &b_one) == -1 && n_thrown == 1, "get out of range -> IOException, -1");
  {
    fake_direct_buffer b_small = DB(&one, 32);
    n_thrown = 0;
    check(Java_surge_replay_gpu_NativeReplay_get(env, NULL, h, 0, &b_small) == -1 && n_thrown == 1 &&
              strcmp(pending_class, "java/lang/IllegalArgumentException") == 0, "get into a 32-byte buffer -> IllegalArgumentException");
  }

  /* micro-batch: one more increment on aggregate 0 */
  {
    int64_t group_agg[1] = {0}, group_off[2] = {0, 1};
    surge_event16 e;
    memset(&e, 0, sizeof(e));
    e.type = SURGE_EVT_INC; e.seq = 6; e.p.i.arg = 1;
    fake_direct_buffer b_ga = DB(group_agg, 8), b_go = DB(group_off, 16), b_e = DB(&e, 16);
    check(Java_surge_replay_gpu_NativeReplay_appendFold(env, NULL, h, &b_ga, &b_go, 1, &b_e, 1) == 0 &&
              Java_surge_replay_gpu_NativeReplay_get(env, NULL, h, 0, &b_one) == 1 && one.count == 6 && one.version == 6,
          "appendFold onto the resident state: (6,6)");
  }
  /* a new aggregate appears after recovery: grow, then an ungrouped micro-batch in topic order */
  {
    int64_t agg_idx[3] = {3, 0, 3}, gidx[2] = {3, 0};
    surge_event16 e[3];
    surge_state64 two[2];
    memset(e, 0, sizeof(e));
    e[0].type = SURGE_EVT_INC; e[0].seq = 1; e[0].p.i.arg = 10;
    e[1].type = SURGE_EVT_DEC; e[1].seq = 7; e[1].p.i.arg = 2;
    e[2].type = SURGE_EVT_INC; e[2].seq = 2; e[2].p.i.arg = 5;
    fake_direct_buffer b_ai = DB(agg_idx, sizeof(agg_idx)), b_e = DB(e, sizeof(e)), b_gi = DB(gidx, sizeof(gidx)), b_two = DB(two, sizeof(two));
    check(Java_surge_replay_gpu_NativeReplay_grow(env, NULL, h, 4) == 0 &&
              Java_surge_replay_gpu_NativeReplay_appendEvents(env, NULL, h, &b_ai, &b_e, 3) == 0 &&
              Java_surge_replay_gpu_NativeReplay_gather(env, NULL, h, &b_gi, 2, &b_two) == 0 &&
              two[0].count == 15 && two[0].version == 2 && (two[0].flags & SURGE_STATE_PRESENT) && two[1].count == 4 && two[1].version == 7,
          "grow + appendEvents + gather: new aggregate (15,2), aggregate 0 (4,7)");
  }
  /* the exchange through JNI: one rank, the handle keeps the gathered snapshot */
  {
    uint8_t id[SURGE_COMM_ID_BYTES];
    int64_t counts[1] = {-1};
    surge_state64 g[4];
    fake_direct_buffer b_id = DB(id, sizeof(id)), b_counts = DB(counts, sizeof(counts)), b_g = DB(g, sizeof(g));
    n_thrown = 0;
    const int ok = Java_surge_replay_gpu_NativeReplay_commUniqueId(env, NULL, &b_id) == 0 &&
                   Java_surge_replay_gpu_NativeReplay_commInit(env, NULL, h, 0, 1, &b_id) == 0 &&
                   Java_surge_replay_gpu_NativeReplay_commCounts(env, NULL, h, 4, 1, &b_counts) == 4 && counts[0] == 4 &&
                   Java_surge_replay_gpu_NativeReplay_allgatherSnapshot(env, NULL, h, 4, 0, SURGE_GATHER_P2P) == 0 &&
                   Java_surge_replay_gpu_NativeReplay_gatheredRead(env, NULL, h, 0, 0, 0, 4, &b_g) == 0;
    check(ok && n_thrown == 0 && g[0].count == 4 && g[0].version == 7 && g[3].count == 15 && !(g[1].flags & SURGE_STATE_PRESENT) &&
              (g[2].flags & SURGE_STATE_POISONED),
          "commUniqueId / commInit / commCounts / allgatherSnapshot / gatheredRead: RCCL behind JNI, one rank");
    n_thrown = 0;
    check(Java_surge_replay_gpu_NativeReplay_gatheredRead(env, NULL, h, 0, 1, 0, 1, &b_g) == SURGE_E_RANGE && n_thrown == 1,
          "gatheredRead of a rank outside the communicator -> IOException");
    check(Java_surge_replay_gpu_NativeReplay_commDestroy(env, NULL, h) == 0, "commDestroy");
  }
  /* one JVM, two handles: the in-process group (peer copies; both handles end up with both shards) */
  {
    surge_state64 g[4];
    surge_event16 e2[2];
    int64_t off2[3] = {0, 1, 2}, hs[2];
    fake_direct_buffer b_g = DB(g, sizeof(g)), b_hs = DB(hs, sizeof(hs)), b_off2 = DB(off2, sizeof(off2)), b_e2 = DB(e2, sizeof(e2));
    const jlong h2 = Java_surge_replay_gpu_NativeReplay_create(env, NULL, NULL, 0);
    memset(e2, 0, sizeof(e2));
    e2[0].type = SURGE_EVT_INC; e2[0].seq = 1; e2[0].p.i.arg = 21;
    e2[1].type = SURGE_EVT_INC; e2[1].seq = 1; e2[1].p.i.arg = 22;
    hs[0] = (int64_t)h; hs[1] = (int64_t)h2;
    n_thrown = 0;
    const int ok = h2 != 0 && Java_surge_replay_gpu_NativeReplay_loadCsr(env, NULL, h2, &b_off2, 2, &b_e2, 2, NULL) == 0 &&
                   Java_surge_replay_gpu_NativeReplay_fold(env, NULL, h2, 0) == 0 &&
                   Java_surge_replay_gpu_NativeReplay_allgatherGroup(env, NULL, &b_hs, 2, 1) == 0 &&
                   Java_surge_replay_gpu_NativeReplay_gatheredRead(env, NULL, h2, 1, 0, 0, 4, &b_g) == 0;
    const int first = ok && g[0].count == 4 && g[0].version == 7 && g[3].count == 15;
    const int ok2 = first && Java_surge_replay_gpu_NativeReplay_gatheredRead(env, NULL, h, 1, 1, 0, 4, &b_g) == 0;
    check(ok2 && n_thrown == 0 && g[0].count == 21 && g[1].count == 22 && !(g[2].flags & SURGE_STATE_PRESENT) && !(g[3].flags & SURGE_STATE_PRESENT),
          "allgatherGroup: two handles in one process, each reads the other's shard; short shard padded with None");
    if (h2) Java_surge_replay_gpu_NativeReplay_destroy(env, NULL, h2);
  }
  {
    /* device decode through JNI: a poll of the Counter fixture's play-json events (TestBoundedContext.scala:122-124) as key /
     * value byte arrays -> decoderPushRecords -> appendDecoded onto an empty store -> (count, version) of agg-1 = (2 - 5, 3) */
    surge_event_json_template t;
    const char* keys = "agg-1:1agg-2:1agg-1:2agg-1:3";
    const char* vals = "{\"aggregateId\":\"agg-1\",\"incrementBy\":1,\"sequenceNumber\":1,\"_type\":\"countIncremented\"}"
                       "{\"_type\":\"countIncremented\",\"aggregateId\":\"agg-2\",\"incrementBy\":7,\"sequenceNumber\":1}"
                       "{\"aggregateId\":\"agg-1\",\"incrementBy\":1,\"sequenceNumber\":2,\"_type\":\"countIncremented\"}"
                       "{\"aggregateId\":\"agg-1\",\"decrementBy\":5,\"sequenceNumber\":3,\"_type\":\"countDecremented\"}";
    const char* v0 = vals;
    int64_t key_off[5] = {0, 7, 14, 21, 28}, val_off[5], offs[4] = {10, 11, 12, 13}, out2[2] = {0, 0}, counts[2] = {0, 0}, koff[3] = {0, 0, 0};
    char utf8[16];
    int i;
    surge_state64 got;
    memset(&t, 0, sizeof(t));
    t.n_types = 2;
    strcpy(t.discriminator, "_type");
    strcpy(t.types[0].name, "countIncremented"); t.types[0].event_type = SURGE_EVT_INC; t.types[0].arg_kind = SURGE_EVJ_ARG_I32;
    strcpy(t.types[0].seq_field, "sequenceNumber"); strcpy(t.types[0].arg_field, "incrementBy");
    strcpy(t.types[1].name, "countDecremented"); t.types[1].event_type = SURGE_EVT_DEC; t.types[1].arg_kind = SURGE_EVJ_ARG_I32;
    strcpy(t.types[1].seq_field, "sequenceNumber"); strcpy(t.types[1].arg_field, "decrementBy");
    val_off[0] = 0;
    for (i = 0; i < 4; ++i) { const char* e = strchr(v0 + val_off[i], '}'); val_off[i + 1] = (int64_t)(e - v0) + 1; }
    {
      fake_direct_buffer b_t = DB(&t, sizeof(t)), b_k = DB(keys, 28), b_ko = DB(key_off, sizeof(key_off)), b_v = DB(vals, val_off[4]),
                         b_vo = DB(val_off, sizeof(val_off)), b_of = DB(offs, sizeof(offs)), b_o2 = DB(out2, sizeof(out2)),
                         b_cn = DB(counts, sizeof(counts)), b_u = DB(utf8, sizeof(utf8)), b_kf = DB(koff, sizeof(koff)), b_st = DB(&got, sizeof(got));
      int64_t zero_off[1] = {0};
      fake_direct_buffer b_z = DB(zero_off, sizeof(zero_off));
      jlong hd, dec;
      n_thrown = 0;
      hd = Java_surge_replay_gpu_NativeReplay_create(env, NULL, NULL, 0);
      dec = Java_surge_replay_gpu_NativeReplay_decoderCreate(env, NULL, &b_t, 0);
      check(hd != 0 && dec != 0 && Java_surge_replay_gpu_NativeReplay_loadCsr(env, NULL, hd, &b_z, 0, NULL, 0, NULL) == 0 &&
                Java_surge_replay_gpu_NativeReplay_fold(env, NULL, hd, 0) == 0 &&
                Java_surge_replay_gpu_NativeReplay_decoderPushRecords(env, NULL, dec, &b_k, &b_ko, &b_v, &b_vo, &b_of, 4) == 0 &&
                Java_surge_replay_gpu_NativeReplay_appendDecoded(env, NULL, hd, dec, &b_o2) == 0 && out2[0] == 4 && out2[1] == 2 && n_thrown == 0,
            "decoderCreate / decoderPushRecords / appendDecoded: a poll of play-json Counter events folded on the device");
      check(Java_surge_replay_gpu_NativeReplay_get(env, NULL, hd, 0, &b_st) == 1 && got.count == -3 && got.version == 3 &&
                Java_surge_replay_gpu_NativeReplay_get(env, NULL, hd, 1, &b_st) == 1 && got.count == 7 && got.version == 1,
            "states after the poll: agg-1 = (1 + 1 - 5, 3), agg-2 = (7, 1)");
      check(Java_surge_replay_gpu_NativeReplay_decoderKeys(env, NULL, dec, &b_u, &b_kf, &b_cn) == 0 && counts[0] == 2 && counts[1] == 10 &&
                memcmp(utf8, "agg-1agg-2", 10) == 0 && koff[1] == 5 && koff[2] == 10,
            "decoderKeys: aggregate ids in first-delivered order");
      {
        /* the same poll through the recovery that folds ONCE: stageDecoded (nothing folded yet) -> packStaged -> one fold */
        jlong h3 = Java_surge_replay_gpu_NativeReplay_create(env, NULL, NULL, 0), dec3 = Java_surge_replay_gpu_NativeReplay_decoderCreate(env, NULL, &b_t, 0);
        out2[0] = out2[1] = 0;
        n_thrown = 0;
        check(h3 != 0 && dec3 != 0 && Java_surge_replay_gpu_NativeReplay_decoderPushRecords(env, NULL, dec3, &b_k, &b_ko, &b_v, &b_vo, &b_of, 4) == 0 &&
                  Java_surge_replay_gpu_NativeReplay_stageDecoded(env, NULL, h3, dec3, &b_o2) == 0 && out2[0] == 4 && out2[1] == 2 &&
                  Java_surge_replay_gpu_NativeReplay_packStaged(env, NULL, h3, 2) == 0 && Java_surge_replay_gpu_NativeReplay_fold(env, NULL, h3, 0) == 0 && n_thrown == 0,
              "stageDecoded / packStaged / fold: the poll staged on the device, packed into one bound log, folded once");
        check(Java_surge_replay_gpu_NativeReplay_get(env, NULL, h3, 0, &b_st) == 1 && got.count == -3 && got.version == 3 &&
                  Java_surge_replay_gpu_NativeReplay_get(env, NULL, h3, 1, &b_st) == 1 && got.count == 7 && got.version == 1,
              "states of the packed log: agg-1 = (1 + 1 - 5, 3), agg-2 = (7, 1)");
        Java_surge_replay_gpu_NativeReplay_decoderDestroy(env, NULL, dec3);
        Java_surge_replay_gpu_NativeReplay_destroy(env, NULL, h3);
      }
      n_thrown = 0;
      val_off[2] = val_off[1] - 3; /* offsets that decrease: refused before the C ABI sees them */
      check(Java_surge_replay_gpu_NativeReplay_decoderPushRecords(env, NULL, dec, &b_k, &b_ko, &b_v, &b_vo, &b_of, 4) == SURGE_E_INVALID && n_thrown == 1,
            "decreasing value offsets -> IllegalArgumentException");
      Java_surge_replay_gpu_NativeReplay_decoderDestroy(env, NULL, dec);
      Java_surge_replay_gpu_NativeReplay_destroy(env, NULL, hd);
    }
  }
  Java_surge_replay_gpu_NativeReplay_destroy(env, NULL, h);
  printf("%s\n", fails ? "FAILED" : "ALL PASS");
  return fails ? 1 : 0;
}
