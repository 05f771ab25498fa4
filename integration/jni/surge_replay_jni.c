/*
 * surge_replay_jni.c — thin JNI shim over include/surge_replay.h.
 *
 * The build image has no JDK (no jni.h): for a JVM this file is compiled where JAVA_HOME is set (below); in
 * this repository it is compiled unchanged against the stand-in header tests/jni_mock/jni.h and driven by a
 * fake JNIEnv (tests/jni_mock/jni_harness.c, tests/test_abi.py) through every Java_… export below.
 *
 *   gcc -shared -fPIC -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -I../../include \
 *       surge_replay_jni.c -L../../surge_amd -lsurge_replay -o libsurge_replay_jni.so
 *
 * Binds `surge.replay.gpu.NativeReplay` (integration/scala/NativeReplay.scala).  Handles are jlong, bulk data are
 * DIRECT java.nio.ByteBuffers in native byte order (no array copies); every buffer's address AND capacity are
 * checked against what the call will touch before the C ABI sees it — a heap buffer or a short one becomes an
 * IllegalArgumentException, never an out-of-bounds access.  A negative status becomes an IOException carrying
 * surge_replay_last_error().
 */
#include <jni.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "surge_replay.h"
#include "surge_ingest.h"

#define H(h) ((surge_replay_handle*)(intptr_t)(h))

static jint check(JNIEnv* env, int32_t rc) {
  if (rc != SURGE_OK) {
    jclass ex = (*env)->FindClass(env, "java/io/IOException");
    const char* msg = surge_replay_last_error(NULL); /* the calling thread's own failure (get() runs on a 32-thread pool) */
    if (ex) (*env)->ThrowNew(env, ex, msg && msg[0] ? msg : "surge_replay call failed");
  }
  return rc;
}

/* Address of a direct buffer that must hold at least `need` bytes (need < 0: unchecked).  NULL buffers are allowed
 * only where the C ABI takes a nullable pointer (`nullable`).  On violation: IllegalArgumentException, *bad = 1. */
static void* buf(JNIEnv* env, jobject b, int64_t need, int nullable, int* bad, const char* what) {
  void* p;
  if (!b) {
    if (nullable) return NULL;
  } else {
    p = (*env)->GetDirectBufferAddress(env, b);
    if (p && (need < 0 || (int64_t)(*env)->GetDirectBufferCapacity(env, b) >= need)) return p;
  }
  if (!*bad) {
    jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (ex) (*env)->ThrowNew(env, ex, what);
  }
  *bad = 1;
  return NULL;
}

/* string offsets handed over by the JVM: non-negative and non-decreasing, so strOff[n] bounds every string */
static int offsets_ok(const int64_t* so, int64_t n) {
  int64_t i;
  if (so[0] < 0) return 0;
  for (i = 0; i < n; ++i)
    if (so[i + 1] < so[i]) return 0;
  return 1;
}

JNIEXPORT jlong JNICALL Java_surge_replay_gpu_NativeReplay_create(JNIEnv* env, jclass c, jobject schemaBuf, jint device) {
  surge_replay_handle* h = NULL;
  surge_replay_schema sc;
  int bad = 0;
  (void)c;
  if (schemaBuf) {
    const void* p = buf(env, schemaBuf, (int64_t)sizeof(sc), 0, &bad, "schema: direct buffer of sizeof(surge_replay_schema) bytes expected");
    if (bad) return 0;
    sc = *(const surge_replay_schema*)p;
  } else {
    surge_replay_default_schema(&sc);
  }
  check(env, surge_replay_create(&sc, device, &h));
  return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL Java_surge_replay_gpu_NativeReplay_destroy(JNIEnv* env, jclass c, jlong h) {
  (void)env; (void)c;
  surge_replay_destroy(H(h));
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_loadCsr(JNIEnv* env, jclass c, jlong h, jobject segOff,
                                                                    jlong nAgg, jobject events, jlong nEvents,
                                                                    jobject initState) {
  int bad = 0;
  const int64_t* so;
  const void *ev, *in;
  (void)c;
  if (nAgg < 0 || nEvents < 0) return SURGE_E_INVALID;
  so = (const int64_t*)buf(env, segOff, (nAgg + 1) * 8, 0, &bad, "segOff: direct buffer of (nAgg + 1) longs expected");
  ev = buf(env, events, nEvents * 16, nEvents == 0, &bad, "events: direct buffer of nEvents x 16 bytes expected");
  in = buf(env, initState, nAgg * 64, 1, &bad, "initState: direct buffer of nAgg x 64 bytes expected");
  if (bad) return SURGE_E_INVALID;
  return check(env, surge_replay_load_csr(H(h), so, nAgg, ev, nEvents, in));
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_fold(JNIEnv* env, jclass c, jlong h, jint algo) {
  (void)c;
  return check(env, surge_replay_fold(H(h), algo));
}

/* Builds the per-log index `algo` needs (SURGE_ALGO_TILED = 7: the tile-major copy of the bound log) without folding, so a
 * recovery can pay the one-off layout cost when the log is bound; out (nullable, >= 24 bytes) receives
 * {index_build_ms, relayout_ms (doubles), tiled_bytes (long)} of surge_replay_layout_info. */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_prepare(JNIEnv* env, jclass c, jlong h, jint algo, jobject out) {
  (void)c;
  int bad = 0;
  unsigned char* o = (unsigned char*)buf(env, out, 24, 1, &bad, "out: direct buffer of 24 bytes expected");
  surge_replay_layout_info_t info;
  int32_t rc;
  if (bad) return SURGE_E_INVALID;
  rc = check(env, surge_replay_prepare(H(h), algo));
  if (rc != SURGE_OK || !o) return rc;
  rc = check(env, surge_replay_layout_info(H(h), &info));
  if (rc != SURGE_OK) return rc;
  memcpy(o, &info.index_build_ms, 8);
  memcpy(o + 8, &info.relayout_ms, 8);
  memcpy(o + 16, &info.tiled_bytes, 8);
  return SURGE_OK;
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_appendFold(JNIEnv* env, jclass c, jlong h, jobject groupAgg,
                                                                       jobject groupOff, jlong nGroups, jobject events,
                                                                       jlong nEvents) {
  int bad = 0;
  const int64_t* ga = (const int64_t*)buf(env, groupAgg, nGroups * 8, 0, &bad, "groupAgg: direct buffer of nGroups longs expected");
  const int64_t* go = (const int64_t*)buf(env, groupOff, (nGroups + 1) * 8, 0, &bad, "groupOff: direct buffer of (nGroups + 1) longs expected");
  const void* ev = buf(env, events, nEvents * 16, 0, &bad, "events: direct buffer of nEvents x 16 bytes expected");
  (void)c;
  if (bad || nGroups < 0 || nEvents < 0) return SURGE_E_INVALID;
  return check(env, surge_replay_append_fold(H(h), ga, go, nGroups, ev, nEvents));
}

/* n events in topic order, event i tagged with aggIdx[i]: grouped inside the library (surge_replay_append_events) */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_appendEvents(JNIEnv* env, jclass c, jlong h, jobject aggIdx,
                                                                         jobject events, jlong nEvents) {
  int bad = 0;
  const int64_t* ai = (const int64_t*)buf(env, aggIdx, nEvents * 8, 0, &bad, "aggIdx: direct buffer of nEvents longs expected");
  const void* ev = buf(env, events, nEvents * 16, 0, &bad, "events: direct buffer of nEvents x 16 bytes expected");
  (void)c;
  if (bad || nEvents < 0) return SURGE_E_INVALID;
  return check(env, surge_replay_append_events(H(h), ai, ev, nEvents));
}

/* New aggregates after recovery: extend the resident state (surge_replay_grow) */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_grow(JNIEnv* env, jclass c, jlong h, jlong newNAgg) {
  (void)c;
  return check(env, surge_replay_grow(H(h), newNAgg));
}

/* Publishes the host mirror that concurrent get() calls then read under a shared lock; states / present may be null. */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_snapshot(JNIEnv* env, jclass c, jlong h, jlong nAgg, jobject states,
                                                                     jobject present) {
  int bad = 0;
  void* st = buf(env, states, nAgg * 64, 1, &bad, "states: direct buffer of nAgg x 64 bytes expected");
  uint8_t* pr = (uint8_t*)buf(env, present, nAgg, 1, &bad, "present: direct buffer of nAgg bytes expected");
  (void)c;
  if (bad) return SURGE_E_INVALID;
  return check(env, surge_replay_snapshot(H(h), st, pr));
}

/* 1 = Some (state64 filled), 0 = None, 2 = POISONED (replay of this aggregate hit an event whose handler throws: the
 * caller must fail the actor's initialisation, PersistentActor.scala:328-333 — never serve the frozen state), -1 = error. */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_get(JNIEnv* env, jclass c, jlong h, jlong aggIdx, jobject state64) {
  uint8_t present = 0;
  int bad = 0;
  surge_state64* st = (surge_state64*)buf(env, state64, 64, 0, &bad, "state64: direct buffer of 64 bytes expected");
  (void)c;
  if (bad) return -1;
  if (check(env, surge_replay_get(H(h), aggIdx, st, &present)) != SURGE_OK) return -1;
  if (st->flags & SURGE_STATE_POISONED) return 2;
  return present;
}

/* Bulk point read: states[i] = state of aggIdx[i] (surge_replay_gather); the snapshot writer's read */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_gather(JNIEnv* env, jclass c, jlong h, jobject aggIdx, jlong n,
                                                                   jobject states) {
  int bad = 0;
  const int64_t* ai;
  void* st;
  (void)c;
  if (n < 0) return SURGE_E_INVALID;
  ai = (const int64_t*)buf(env, aggIdx, n * 8, n == 0, &bad, "aggIdx: direct buffer of n longs expected");
  st = buf(env, states, n * 64, n == 0, &bad, "states: direct buffer of n x 64 bytes expected");
  if (bad) return SURGE_E_INVALID;
  return check(env, surge_replay_gather(H(h), ai, n, st));
}

/* partitionForKey of whole strings (KafkaPartitioner.scala:8) */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_partitionHash(JNIEnv* env, jclass c, jobject utf16,
                                                                          jobject strOff, jlong n, jint nPartitions,
                                                                          jobject partOut) {
  int bad = 0;
  const int64_t* so;
  const uint16_t* u;
  int32_t* po;
  (void)c;
  if (n < 0) return SURGE_E_INVALID; /* before any size is derived from it */
  so = (const int64_t*)buf(env, strOff, (n + 1) * 8, 0, &bad, "strOff: direct buffer of (n + 1) longs expected");
  if (!bad && !offsets_ok(so, n)) {
    jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (ex) (*env)->ThrowNew(env, ex, "strOff: offsets must start at >= 0 and never decrease");
    return SURGE_E_INVALID;
  }
  u = (const uint16_t*)buf(env, utf16, bad ? -1 : so[n] * 2, 1, &bad, "utf16: direct buffer of strOff[n] chars expected");
  po = (int32_t*)buf(env, partOut, n * 4, 0, &bad, "partOut: direct buffer of n ints expected");
  if (bad) return SURGE_E_INVALID;
  return check(env, surge_replay_partition_hash(u, so, n, nPartitions, po));
}

/* ... after PartitionStringUpToColon.partitionBy (KafkaPartitioner.scala:38-42) */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_partitionHashUpToColon(JNIEnv* env, jclass c, jobject utf16,
                                                                                   jobject strOff, jlong n, jint nPartitions,
                                                                                   jobject partOut) {
  int bad = 0;
  const int64_t* so;
  const uint16_t* u;
  int32_t* po;
  (void)c;
  if (n < 0) return SURGE_E_INVALID; /* before any size is derived from it */
  so = (const int64_t*)buf(env, strOff, (n + 1) * 8, 0, &bad, "strOff: direct buffer of (n + 1) longs expected");
  if (!bad && !offsets_ok(so, n)) {
    jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (ex) (*env)->ThrowNew(env, ex, "strOff: offsets must start at >= 0 and never decrease");
    return SURGE_E_INVALID;
  }
  u = (const uint16_t*)buf(env, utf16, bad ? -1 : so[n] * 2, 1, &bad, "utf16: direct buffer of strOff[n] chars expected");
  po = (int32_t*)buf(env, partOut, n * 4, 0, &bad, "partOut: direct buffer of n ints expected");
  if (bad) return SURGE_E_INVALID;
  return check(env, surge_replay_partition_hash_up_to_colon(u, so, n, nPartitions, po));
}

/* ---- multi-GPU exchange: RCCL behind the C ABI; the host moves only the 128-byte communicator id ---- */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_commUniqueId(JNIEnv* env, jclass c, jobject idOut) {
  int bad = 0;
  uint8_t* id = (uint8_t*)buf(env, idOut, SURGE_COMM_ID_BYTES, 0, &bad, "idOut: direct buffer of 128 bytes expected");
  (void)c;
  if (bad) return SURGE_E_INVALID;
  return check(env, surge_replay_comm_unique_id(id));
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_commInit(JNIEnv* env, jclass c, jlong h, jint rank, jint world,
                                                                     jobject id) {
  int bad = 0;
  const uint8_t* p = (const uint8_t*)buf(env, id, SURGE_COMM_ID_BYTES, 0, &bad, "id: direct buffer of 128 bytes expected");
  (void)c;
  if (bad) return SURGE_E_INVALID;
  return check(env, surge_replay_comm_init(H(h), rank, world, p));
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_commDestroy(JNIEnv* env, jclass c, jlong h) {
  (void)c;
  return check(env, surge_replay_comm_destroy(H(h)));
}

/* counts[r] = states rank r contributes; returns the largest (rows per rank of the gathered snapshot), -1 on error */
JNIEXPORT jlong JNICALL Java_surge_replay_gpu_NativeReplay_commCounts(JNIEnv* env, jclass c, jlong h, jlong nLocal, jint world,
                                                                        jobject countsOut) {
  int bad = 0;
  int64_t mx = 0;
  int64_t* co = (int64_t*)buf(env, countsOut, (int64_t)world * 8, 1, &bad, "countsOut: direct buffer of `world` longs expected");
  (void)c;
  if (bad) return -1;
  if (check(env, surge_replay_comm_counts(H(h), nLocal, co, &mx)) != SURGE_OK) return -1;
  return (jlong)mx;
}

/* all-gather of the handle's resident state into a buffer the handle owns (a JVM has no device pointers) */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_allgatherSnapshot(JNIEnv* env, jclass c, jlong h, jlong nLocal,
                                                                              jint slot, jint mode) {
  (void)c;
  return check(env, surge_replay_allgather_snapshot(H(h), NULL, nLocal, NULL, 0, slot, mode));
}

/* One JVM drives several GPUs: handles = direct buffer of n native-order longs (handle r = rank r of an in-process
 * group; peer copies, no RCCL, no rendezvous).  Every handle keeps the gathered snapshot (gatheredRead). */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_allgatherGroup(JNIEnv* env, jclass c, jobject handles, jint n, jint slot) {
  int bad = 0;
  surge_replay_handle* hs[64];
  const int64_t* raw = (const int64_t*)buf(env, handles, (int64_t)n * 8, 0, &bad, "handles: direct buffer of n longs expected");
  (void)c;
  if (bad) return SURGE_E_INVALID;
  if (n < 1 || n > 64) {
    jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (ex) (*env)->ThrowNew(env, ex, "allgatherGroup: 1..64 handles");
    return SURGE_E_INVALID;
  }
  for (jint r = 0; r < n; ++r) hs[r] = H(raw[r]);
  return check(env, surge_replay_allgather(hs, n, NULL, NULL, 0, slot));
}

/* rows [firstRow, firstRow + nRows) of rank `rank`'s block of the gathered snapshot (waits for that slot's exchange) */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_gatheredRead(JNIEnv* env, jclass c, jlong h, jint slot, jint rank,
                                                                         jlong firstRow, jlong nRows, jobject states) {
  int bad = 0;
  void* st = buf(env, states, nRows * 64, nRows == 0, &bad, "states: direct buffer of nRows x 64 bytes expected");
  (void)c;
  if (bad || nRows < 0) return SURGE_E_INVALID;
  return check(env, surge_replay_gathered_read(H(h), slot, rank, firstRow, nRows, st));
}

/* ---- device decode (include/surge_ingest.h): ConsumerRecords in bulk -> resident state, no per-record JVM work ----
 * The JVM copies the key / value bytes of a poll into direct buffers (System.arraycopy) and crosses JNI twice per poll:
 * decoderPushRecords (parse keys, intern aggregate ids, decode the event values — 16-byte events or the plugin's
 * play-json text through the template — on the GPU) and appendDecoded (grow + device group-by + fold). */
#define D(d) ((surge_device_decoder*)(intptr_t)(d))

static jint check_dec(JNIEnv* env, int32_t rc) {
  if (rc != SURGE_OK) {
    jclass ex = (*env)->FindClass(env, "java/io/IOException");
    const char* msg = surge_device_decoder_last_error(NULL);
    if (ex) (*env)->ThrowNew(env, ex, msg && msg[0] ? msg : "surge_device_decoder call failed");
  }
  return rc;
}

/* templateBuf: a surge_event_json_template (native layout) or null for topics whose values are 16-byte events */
JNIEXPORT jlong JNICALL Java_surge_replay_gpu_NativeReplay_decoderCreate(JNIEnv* env, jclass c, jobject templateBuf, jint device) {
  surge_device_decoder* d = NULL;
  int bad = 0;
  const void* t = buf(env, templateBuf, (int64_t)sizeof(surge_event_json_template), 1, &bad,
                      "template: direct buffer of sizeof(surge_event_json_template) bytes expected");
  (void)c;
  if (bad) return 0;
  check_dec(env, surge_device_decoder_create(device, NULL, (const surge_event_json_template*)t, &d));
  return (jlong)(intptr_t)d;
}

JNIEXPORT void JNICALL Java_surge_replay_gpu_NativeReplay_decoderDestroy(JNIEnv* env, jclass c, jlong d) {
  (void)env; (void)c;
  surge_device_decoder_destroy(D(d));
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_decoderPushRecords(JNIEnv* env, jclass c, jlong d, jobject keys, jobject keyOff,
                                                                               jobject values, jobject valueOff, jobject offsets, jlong n) {
  int bad = 0;
  const int64_t *ko, *vo, *of;
  const uint8_t *k, *v;
  (void)c;
  if (n < 0) return SURGE_E_INVALID;
  ko = (const int64_t*)buf(env, keyOff, (n + 1) * 8, 0, &bad, "keyOff: direct buffer of (n + 1) longs expected");
  vo = (const int64_t*)buf(env, valueOff, (n + 1) * 8, 0, &bad, "valueOff: direct buffer of (n + 1) longs expected");
  if (!bad && (!offsets_ok(ko, n) || !offsets_ok(vo, n))) {
    jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (ex) (*env)->ThrowNew(env, ex, "keyOff / valueOff: offsets must start at >= 0 and never decrease");
    return SURGE_E_INVALID;
  }
  k = (const uint8_t*)buf(env, keys, bad ? -1 : ko[n], 1, &bad, "keys: direct buffer of keyOff[n] bytes expected");
  v = (const uint8_t*)buf(env, values, bad ? -1 : vo[n], 1, &bad, "values: direct buffer of valueOff[n] bytes expected");
  of = (const int64_t*)buf(env, offsets, n * 8, 1, &bad, "offsets: direct buffer of n longs expected");
  if (bad) return SURGE_E_INVALID;
  return check_dec(env, surge_device_decoder_push_records(D(d), k, ko, v, vo, of, n));
}

/* out (nullable, 16 bytes): {events folded, keys known} */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_appendDecoded(JNIEnv* env, jclass c, jlong h, jlong d, jobject out) {
  int bad = 0;
  int64_t* o = (int64_t*)buf(env, out, 16, 1, &bad, "out: direct buffer of 16 bytes expected");
  int64_t n_events = 0, n_keys = 0;
  int32_t rc;
  (void)c;
  if (bad) return SURGE_E_INVALID;
  rc = check_dec(env, surge_replay_append_decoded(H(h), D(d), &n_events, &n_keys));
  if (o) { o[0] = n_events; o[1] = n_keys; }
  return rc;
}

/* A recovery that folds ONCE (surge_replay.h, "the device packer"): stageDecoded per poll instead of appendDecoded — the decoded
 * events are kept on the device in topic order —, then packStaged(nAgg = the decoder's key count) and ONE fold.
 * out (nullable, 16 bytes): {events staged by this call, keys known} */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_stageDecoded(JNIEnv* env, jclass c, jlong h, jlong d, jobject out) {
  int bad = 0;
  int64_t* o = (int64_t*)buf(env, out, 16, 1, &bad, "out: direct buffer of 16 bytes expected");
  int64_t n_events = 0, n_keys = 0;
  int32_t rc;
  (void)c;
  if (bad) return SURGE_E_INVALID;
  rc = check_dec(env, surge_replay_stage_decoded(H(h), D(d), &n_events, &n_keys));
  if (o) { o[0] = n_events; o[1] = n_keys; }
  return rc;
}

JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_packStaged(JNIEnv* env, jclass c, jlong h, jlong nAgg) {
  (void)c;
  return check(env, surge_replay_pack_staged(H(h), (int64_t)nAgg));
}

/* The key table: utf8Out / keyOffOut nullable (size query); counts (16 bytes) receives {keys, utf8 bytes} */
JNIEXPORT jint JNICALL Java_surge_replay_gpu_NativeReplay_decoderKeys(JNIEnv* env, jclass c, jlong d, jobject utf8Out, jobject keyOffOut,
                                                                        jobject counts) {
  int bad = 0;
  int64_t n_keys = 0, n_bytes = 0;
  int64_t* cn = (int64_t*)buf(env, counts, 16, 0, &bad, "counts: direct buffer of 16 bytes expected");
  uint8_t* u;
  int64_t* ko;
  int32_t rc;
  (void)c;
  if (bad) return SURGE_E_INVALID;
  rc = check_dec(env, surge_device_decoder_keys(D(d), NULL, 0, NULL, &n_keys, &n_bytes));
  if (rc != SURGE_OK) return rc;
  cn[0] = n_keys; cn[1] = n_bytes;
  if (!utf8Out && !keyOffOut) return SURGE_OK;
  u = (uint8_t*)buf(env, utf8Out, n_bytes, n_bytes == 0, &bad, "utf8Out: direct buffer of counts[1] bytes expected");
  ko = (int64_t*)buf(env, keyOffOut, (n_keys + 1) * 8, 0, &bad, "keyOffOut: direct buffer of (counts[0] + 1) longs expected");
  if (bad) return SURGE_E_INVALID;
  return check_dec(env, surge_device_decoder_keys(D(d), u, n_bytes, ko, &n_keys, &n_bytes));
}
