/*
 * surge_replay.h — C ABI of the MI355X-native aggregate-replay engine.
 *
 * This is the drop-in boundary for ONE hot path of UltimateSoftware/surge: the
 * per-aggregate event fold that reconstructs the aggregate state store.  The
 * reference is 100 % JVM and has no FFI; the entry points below are what a JNI
 * shim binds so that the engine can sit behind the
 * reference's own seams.  Citations are into /root/reference (read-only):
 *
 *   R2  the fold                events.foldLeft(state)(handleEvent)
 *       modules/command-engine/scaladsl/src/main/scala/surge/scaladsl/command/CommandModels.scala:20,26
 *   S1  persistence plugin      SurgeKafkaStreamsPersistencePlugin.createSupplier
 *       modules/common/src/main/scala/surge/kafka/streams/SurgeKafkaStreamsPersistencePlugin.scala:12-15
 *   S2  state read seam         AggregateStateStoreKafkaStreams.getAggregateBytes
 *       modules/common/src/main/scala/surge/kafka/streams/AggregateStateStoreKafkaStreams.scala:83-85
 *   R12 store-side recovery     SurgeStateStoreConsumer.ktableIndexingTopology
 *       modules/common/src/main/scala/surge/kafka/streams/SurgeStateStoreConsumer.scala:57-76
 *   R15 shard map               KafkaPartitionProvider.partitionForKey
 *       modules/common/src/main/scala/surge/kafka/KafkaPartitioner.scala:8,38-42
 *
 * Conventions
 *   - plain C, no exceptions across the boundary; every function returns an
 *     int32 status (0 = OK, negative = error class, see SURGE_E_*).
 *   - surge_replay_last_error(h) returns a NUL-terminated message owned by the
 *     handle (or by the calling thread when h == NULL; concurrent readers should
 *     use the NULL form, which always reports the calling thread's own failure).
 *   - all buffers are caller-allocated.  "host" buffers are ordinary process
 *     memory (JNI passes DirectByteBuffer addresses); "_device" entry points
 *     take HIP device pointers (the Python host passes torch tensor pointers).
 *   - one handle per GPU; mutation of a handle is NOT thread-safe, point reads
 *     (surge_replay_get) are: shared (reader) lock against the mirror published by
 *     surge_replay_snapshot, serialized device reads otherwise (the reference reads
 *     S2 from a 32-thread pool, ThreadPools.scala:10-11).
 *   - There is NO CPU fallback inside this library: without a usable HIP device
 *     surge_replay_create fails with SURGE_E_DEVICE.
 */
#ifndef SURGE_REPLAY_H
#define SURGE_REPLAY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SURGE_REPLAY_ABI_VERSION 1u

/* ---- status codes --------------------------------------------------------- */
#define SURGE_OK             0
#define SURGE_E_INVALID     (-1) /* bad argument (NULL, negative size, non-monotone seg_off ...) */
#define SURGE_E_STATE       (-2) /* call order (fold before load, get before fold ...)           */
#define SURGE_E_DEVICE      (-3) /* HIP runtime error / no device                               */
#define SURGE_E_NOMEM       (-4) /* host or device allocation failed                            */
#define SURGE_E_UNSUPPORTED (-5) /* schema / algorithm not supported by this build              */
#define SURGE_E_RANGE       (-6) /* aggregate index out of range                                */
/* (-7 is SURGE_E_CORRUPT of surge_ingest.h)                                                        */
#define SURGE_E_COMM        (-8) /* RCCL not loadable / a collective call failed                    */

/* ---- fixed-width layouts (little-endian) ----------------------------------
 *
 * The reference has no fixed-width format (states are JSON); these
 * layouts are the engine's restatement of the reference fixtures' fields:
 *   count/version  <- Counter  State(aggregateId, count:Int, version:Int)
 *       modules/command-engine/scaladsl/src/test/scala/surge/scaladsl/TestBoundedContext.scala:15
 *   balance        <- BankAccount(..., balance: Double)
 *       modules/surge-docs/src/test/scala/docs/command/BankAccountCommandModel.scala:19
 * Aggregate-id strings never cross this boundary: callers keep a key table and
 * address aggregates by dense index.
 */
typedef struct surge_event16 {
  int32_t type;        /* index into surge_replay_schema.desc; out of range => poison */
  int32_t seq;         /* sequenceNumber of the event (TestBoundedContext.scala:51)  */
  union {
    struct { int32_t arg; int32_t pad; } i; /* integer payload (incrementBy / decrementBy) */
    double value;                           /* f64 payload (balance / newBalance)          */
    uint64_t raw;
  } p;
} surge_event16;

#define SURGE_STATE_PRESENT  1u /* flags bit0: Some(...) vs None                       */
#define SURGE_STATE_POISONED 2u /* flags bit1: an event "threw"; state frozen before it */

typedef struct surge_state64 {
  int32_t  count;       /*  0 */
  int32_t  version;     /*  4 */
  int64_t  sum64;       /*  8 */
  double   balance;     /* 16  (moved bit-exactly, never computed on)                  */
  int32_t  min_arg;     /* 24 */
  int32_t  max_arg;     /* 28 */
  uint32_t event_count; /* 32 */
  uint32_t flags;       /* 36  SURGE_STATE_*                                           */
  uint8_t  reserved[24];/* 40  always zero                                             */
} surge_state64;
/* Canonical encoding of None: all 64 bytes zero except possibly SURGE_STATE_POISONED. */

/* ---- event algebra -------------------------------------
 * handleEvent is arbitrary JVM code (CommandModels.scala:14) and cannot run on
 * a GPU.  A plugin declares, beside its handleEvent, one 32-bit descriptor per
 * event type; the descriptor is the kernel-able restatement of that case of
 * handleEvent.  Presence classes are literal readings of the fixtures:
 *   MATERIALIZE  agg.getOrElse(State(id,0,0)) then update, always Some
 *                (TestBoundedContext.scala:78,88)
 *   REQUIRE      aggregate.map(_.copy(...)): dropped when None
 *                (BankAccountCommandModel.scala:84)
 *   CREATE       Some(<built only from the event>): overwrites
 *                (BankAccountCommandModel.scala:83)
 *   DELETE       handleEvent returns None => tombstone (SurgeModel.scala:62)
 */
#define SURGE_CLS_MATERIALIZE 0u
#define SURGE_CLS_REQUIRE     1u
#define SURGE_CLS_CREATE      2u
#define SURGE_CLS_DELETE      3u
#define SURGE_CLS_MASK        3u
#define SURGE_D_POISON        (1u << 2)  /* handleEvent throws (TestBoundedContext.scala:86) */
#define SURGE_D_COUNT_ADD     (1u << 4)  /* count += arg  (32-bit wrap, JVM Int)             */
#define SURGE_D_COUNT_SUB     (2u << 4)  /* count -= arg                                     */
#define SURGE_D_COUNT_SET     (3u << 4)  /* count  = arg                                     */
#define SURGE_D_COUNT_MASK    (3u << 4)
#define SURGE_D_VERSION_SET   (1u << 6)  /* version = seq                                    */
#define SURGE_D_SUM_ADD       (1u << 8)  /* sum64 += (int64)arg                              */
#define SURGE_D_SUM_SUB       (2u << 8)  /* sum64 -= (int64)arg                              */
#define SURGE_D_SUM_MASK      (3u << 8)
#define SURGE_D_BALANCE_SET   (1u << 10) /* balance = value (bit copy)                       */
#define SURGE_D_MIN_ARG       (1u << 11) /* min_arg = min(min_arg, arg)                      */
#define SURGE_D_MAX_ARG       (1u << 12) /* max_arg = max(max_arg, arg)                      */
#define SURGE_D_EVCOUNT_INC   (1u << 13) /* event_count += 1                                 */

#define SURGE_MAX_EVENT_TYPES 16

typedef struct surge_replay_schema {
  uint32_t abi_version;              /* SURGE_REPLAY_ABI_VERSION                    */
  uint32_t state_size;               /* must be 64                                  */
  uint32_t event_size;               /* must be 16                                  */
  uint32_t n_types;                  /* 1..SURGE_MAX_EVENT_TYPES                    */
  uint32_t desc[SURGE_MAX_EVENT_TYPES];
  surge_state64 default_state;       /* fields an absent aggregate materialises to;
                                        flags/reserved ignored                      */
} surge_replay_schema;

/* ---- ABI v2: field SLOTS (generalised) --------------------------------------------------------
 * handleEvent is arbitrary (CommandModels.scala:14); the v1 descriptor above can only express its seven named
 * fields.  A v2 schema declares up to 7 typed 8-byte SLOTS inside the same 64-byte state and, per event type, one
 * operation per slot.  A model with two counters and a Long version — which v1 cannot express — is three slots.
 *
 *   state (64 B): slot 0..3 at bytes 0, 8, 16, 24 | u32 event_count at 32 | u32 flags at 36 (as in v1) | slot 4..6 at 40, 48, 56
 *                 an I32 slot uses the low 4 bytes of its 8 (the rest are zero — also in a prior snapshot handed to load /
 *                 bind; a fold writes them as zero for every aggregate that has events); None is all-zero, as in v1
 *   slot i:  type  SURGE_SLOT_I32 / _I64 / _F64
 *            source of the operand x of every operation on it:
 *              SURGE_SRC_ARG      (int32) low half of the event payload   (incrementBy)
 *              SURGE_SRC_SEQ      (int32) event.seq                       (sequenceNumber)
 *              SURGE_SRC_PAYLOAD  the 8-byte payload (i64 / f64 bits; an I32 slot takes its low half)
 *              SURGE_SRC_ONE      the constant 1 (1.0 for F64)
 *              (ARG / SEQ are sign-extended for I64 and converted exactly for F64)
 *            default_bits: what an absent aggregate materialises to
 *   event type t:  cls[t] = presence class (SURGE_CLS_*, as v1) | SURGE_D_POISON
 *                  ops[t] = 4 bits per slot, slot i in bits [4i, 4i+4): SURGE_OP_KEEP / ADD / SUB / SET / MIN / MAX
 *   integer ADD / SUB wrap like JVM Int / Long; F64 ADD / SUB are IEEE double operations applied STRICTLY in event
 *   order (the JVM fold's order; the library is built with -ffp-contract=off), so the result is bit-identical to the
 *   sequential fold — no tolerance needed.  F64 MIN / MAX are java.lang.Math.min(cur, x) / Math.max(cur, x) — what a JVM
 *   model's `math.min` / `math.max` call — as the JDK's library source states them: a NaN on either side gives that NaN
 *   (bits preserved), -0.0 orders below +0.0, otherwise `cur <= x ? cur : x` / `cur >= x ? cur : x`.  (HotSpot's
 *   intrinsic may return a different NaN bit pattern than the library source; the value is NaN either way.)
 * Any mix of operations on one slot is allowed: v2 handles always fold with ONE lane per aggregate (or per
 * micro-batch group) walking its events in order (fold_slots.hip), so nothing has to be associative.  The kernels are
 * COMPILED FOR THE SCHEMA when the handle is created (hiprtc, about a second the first time a process sees a schema;
 * surge_replay_kernel_info): the walk then costs the schema's own arithmetic.  Without libhiprtc — or with
 * SURGE_REPLAY_RTC=0 — the same device code runs as a generic interpreter (2-3x slower, same results).  Everything
 * else of the ABI (load / bind, fold, append_*, get, gather, snapshot, grow, encoders, snapshot_delta, the exchange)
 * works on v2 handles; surge_replay_fold accepts SURGE_ALGO_AUTO (= SURGE_ALGO_SLOTS: the bound CSR log through
 * length-sorted row pieces) and SURGE_ALGO_TILED (the tile-major copy, rows never cut). */
#define SURGE_REPLAY_ABI_VERSION_2 2u
#define SURGE_MAX_SLOTS 7
#define SURGE_SLOT_I32 1u
#define SURGE_SLOT_I64 2u
#define SURGE_SLOT_F64 3u
#define SURGE_SRC_ARG     0u
#define SURGE_SRC_SEQ     1u
#define SURGE_SRC_PAYLOAD 2u
#define SURGE_SRC_ONE     3u
#define SURGE_OP_KEEP 0u
#define SURGE_OP_ADD  1u
#define SURGE_OP_SUB  2u
#define SURGE_OP_SET  3u
#define SURGE_OP_MIN  4u
#define SURGE_OP_MAX  5u
#define SURGE_V2_COUNT_EVENTS 1u /* schema flag: event_count += 1 for every event that applies */

typedef struct surge_slot_def {
  uint8_t  type;          /* SURGE_SLOT_*                  */
  uint8_t  source;        /* SURGE_SRC_*                   */
  uint8_t  reserved[6];
  uint64_t default_bits;  /* the slot of a materialised default state (I32: low 4 bytes) */
} surge_slot_def;

typedef struct surge_replay_schema_v2 {
  uint32_t abi_version;   /* SURGE_REPLAY_ABI_VERSION_2 */
  uint32_t state_size;    /* 64 */
  uint32_t event_size;    /* 16 */
  uint32_t n_types;       /* 1..SURGE_MAX_EVENT_TYPES */
  uint32_t n_slots;       /* 1..SURGE_MAX_SLOTS */
  uint32_t flags;         /* SURGE_V2_* */
  surge_slot_def slot[SURGE_MAX_SLOTS];
  uint32_t cls[SURGE_MAX_EVENT_TYPES];
  uint32_t ops[SURGE_MAX_EVENT_TYPES];
} surge_replay_schema_v2;

/* byte offset of slot i in the 64-byte state */
#define SURGE_SLOT_OFFSET(i) ((i) < 4 ? 8 * (i) : 8 * (i) + 8)

/* Built-in event types of the reference fixtures' algebra (default schema). */
#define SURGE_EVT_NOOP        0 /* NoOpEvent            TestBoundedContext.scala:63,85    */
#define SURGE_EVT_INC         1 /* CountIncremented     TestBoundedContext.scala:55,81-82 */
#define SURGE_EVT_DEC         2 /* CountDecremented     TestBoundedContext.scala:59,83-84 */
#define SURGE_EVT_CREATE      3 /* BankAccountCreated   BankAccountCommandModel.scala:39,83 */
#define SURGE_EVT_SET_BALANCE 4 /* BankAccountUpdated   BankAccountCommandModel.scala:46,84 */
#define SURGE_EVT_DELETE      5 /* handleEvent => None  SurgeModel.scala:62 (tombstone)  */
#define SURGE_EVT_THROW       6 /* ExceptionThrowingEvent TestBoundedContext.scala:67,86 */

/* ---- fold algorithms -------------------------------------------------------- */
#define SURGE_ALGO_AUTO  0 /* uniform segment length L (L % 16 == 0): ROWS when there are enough aggregates
                              to fill the chip, else FIXED; otherwise CHUNKED for logs whose aggregates average
                              >= 64 events, else FLAT */
#define SURGE_ALGO_FIXED 1 /* K1b: flat fold with segment heads computed arithmetically (uniform L)   */
#define SURGE_ALGO_FLAT  2 /* K2: load-balanced flat segmented scan over the CSR                      */
#define SURGE_ALGO_ROWS  3 /* K1: uniform L, one lane per aggregate, no cross-lane scan               */
#define SURGE_ALGO_SORTED 4 /* K2b: any CSR; segments counting-sorted by length at load time, persistent
                               waves walk groups of 64 similar-length segments, one lane per aggregate  */
#define SURGE_ALGO_SLOTS 6 /* v2 handles only: sorted-rows transport + slot interpreter, one lane per aggregate */
#define SURGE_ALGO_CHUNKED 5 /* K2c: like SORTED, but an aggregate longer than T events (T ~ log bytes / 4 MB) is cut into
                                independent line-aligned chunks, each a virtual row of its own, stitched left to right
                                by a second, tiny kernel: no wave ever walks more than ~T events alone (mid-size
                                ragged logs) */
#define SURGE_ALGO_TILED 7 /* K2t: CHUNKED's virtual rows folded from a TILE-MAJOR COPY of the log the handle builds once
                              per bound log (groups of 64 similar-length rows, each 8-event x 64-row subtile one
                              contiguous 8 KiB run): every load instruction of the fold covers contiguous memory, the
                              fold streams linearly.  Costs one extra copy of the log in device memory and one re-layout
                              pass (about the time of four folds; surge_replay_layout_info reports it), so it pays when a
                              bound log is folded more than once or is bound ahead of need: never chosen by
                              SURGE_ALGO_AUTO, ask for it (surge_replay_prepare does the copy without folding).  Any
                              CSR; v2 handles fold the same copy with whole aggregates as rows (nothing cut). */
#define SURGE_ALGO_SHORT 8 /* K1s: short rows — one lane per aggregate straight from the CSR arrays, no LDS transport, no index:
                              a wave's 64 consecutive rows are one contiguous stretch of the log, every lane loads its own
                              few events (16 B each, four ahead) and walks a concrete state.  For logs of MANY SHORT
                              aggregates — what a packed events topic looks like when most aggregates published a handful
                              of events — where FLAT spends its time on segment heads (one state store per head) and the
                              lane-per-row kernels' 16-event tiles are mostly padding.  SURGE_ALGO_AUTO picks it when the
                              longest aggregate has at most 64 events and the mean is below 16.  Any CSR.               */

typedef struct surge_replay_stats_t {
  int64_t n_aggregates;
  int64_t n_events;
  int64_t algorithmic_bytes;   /* 16*E + 8*(A+1) + 64*A*(1+r)                  */
  double  last_fold_kernel_ms; /* HIP-event time of the dominant fold kernel, last launch */
  double  last_fold_total_ms;  /* HIP-event time of the whole fold (plan + fold + fill)    */
  double  h2d_ms;              /* last load_csr / append_fold host->device copy           */
  int32_t last_algo;           /* SURGE_ALGO_* actually run                               */
  int32_t n_tasks;             /* wave tasks of the last fold                             */
  int64_t n_folds;             /* folds since create                                      */
  int64_t n_poisoned;          /* aggregates flagged POISONED after the last fold                */
  double  sum_fold_kernel_ms;  /* sum of the dominant kernel's HIP-event times since stats_reset  */
  int64_t timed_folds;         /* number of folds in that sum (at most 256 are kept)              */
} surge_replay_stats_t;

/* The per-log index the handle built for the last SORTED / CHUNKED / TILED fold (or surge_replay_prepare) and what
 * building it cost: device time between HIP events on the handle's stream, paid once per bound log and never part of a
 * fold's kernel time.  Replaces nothing in the reference — its restore consumer (SurgeStateStoreConsumer.scala:57-76)
 * has no index; this is what "recovery happens once" costs here beside the fold itself. */
typedef struct surge_replay_layout_info_t {
  int32_t algo;              /* SURGE_ALGO_SORTED / _CHUNKED / _TILED, 0 = no index built for the bound log   */
  int32_t chunk_events;      /* CHUNKED / TILED: aggregates longer than this were cut into chunks (T)          */
  int64_t virtual_rows;      /* rows the lanes walk: aggregates, or aggregates + the extra chunks of cut ones   */
  int64_t cut_aggregates;    /* aggregates cut into more than one chunk                                        */
  int64_t tiled_bytes;       /* TILED: size of the tile-major copy of the log (0 otherwise)                     */
  int64_t padding_events;    /* TILED: PAD events in that copy (rows rounded up to their group's longest)       */
  double  index_build_ms;    /* length order / chunk table / tile index                                        */
  double  relayout_ms;       /* TILED: the copy into tile-major order                                          */
} surge_replay_layout_info_t;

/* Which build of the fold kernels a handle runs. */
typedef struct surge_replay_kernel_info_t {
  int32_t specialised;   /* 1: kernels compiled at create time for this handle's schema (v2 handles, hiprtc); 0: the
                            ahead-of-time kernels (every v1 handle; v2: the generic slot interpreter)                 */
  int32_t reserved;
  double  compile_ms;    /* wall time of that compilation (shared by every handle of the process with the same schema) */
  char    detail[240];   /* the libhiprtc that compiled them, or why the interpreter runs instead                      */
} surge_replay_kernel_info_t;

typedef struct surge_replay_handle surge_replay_handle;

/* ---- lifecycle -------------------------------------------------------------- */

/* Fills *out with the built-in algebra (Counter + BankAccount fixtures). */
int32_t surge_replay_default_schema(surge_replay_schema* out);

/* Binds a schema to one GPU.  Replaces: constructing the KTable store behind
 * SurgeKafkaStreamsPersistencePlugin.createSupplier (…PersistencePlugin.scala:13). */
int32_t surge_replay_create(const surge_replay_schema* schema, int32_t device_id,
                            surge_replay_handle** out);
/* Same, for a v2 slot schema. */
int32_t surge_replay_create_v2(const surge_replay_schema_v2* schema, int32_t device_id, surge_replay_handle** out);
int32_t surge_replay_kernel_info(surge_replay_handle* h, surge_replay_kernel_info_t* out);
/* The code object surge_replay_create_v2 would compile for `schema` on an `arch` device ("gfx950"): needs libhiprtc, no
 * GPU.  code_out nullable (size query); *code_bytes is set either way.  For build-time checks of a model's schema. */
int32_t surge_replay_compile_schema_v2(const surge_replay_schema_v2* schema, const char* arch, void* code_out, int64_t capacity,
                                       int64_t* code_bytes);
/* The same for a v1 schema: the code object of the flat fold kernel (K3 appends; AUTO on logs of few long rows) compiled
 * for the schema's op table — the one v1 kernel that is bound by its instruction stream, and the one a v1 handle compiles
 * at its first flat fold (surge_replay_kernel_info says which build runs; SURGE_REPLAY_RTC=0 keeps the ahead-of-time one). */
int32_t surge_replay_compile_schema(const surge_replay_schema* schema, const char* arch, void* code_out, int64_t capacity, int64_t* code_bytes);
int32_t surge_replay_destroy(surge_replay_handle* h);
const char* surge_replay_last_error(const surge_replay_handle* h);

/* Launch all work of this handle on the given hipStream_t (NULL = default stream). */
int32_t surge_replay_set_stream(surge_replay_handle* h, void* hip_stream);
/* The hipStream_t the handle's work is enqueued on (what set_stream was given; NULL = default stream): for a host that
 * orders its own device work against the handle's with events instead of surge_replay_synchronize. */
int32_t surge_replay_get_stream(surge_replay_handle* h, void** hip_stream_out);
/* Waits for the handle's stream; also where a skipped device micro-batch (see surge_replay_append_events_device) is
 * reported. */
int32_t surge_replay_synchronize(surge_replay_handle* h);

/* ---- load -------------------------------------------------------------------
 * One shard's CSR-packed event log.  Events of aggregate a are
 * events[seg_off[a] .. seg_off[a+1]) in publish (Kafka offset) order — what one
 * partition of the events topic holds for keys "<id>:<seq>" under
 * PartitionStringUpToColon (KafkaPartitioner.scala:38-42).
 * init_state (nullable, n_agg x 64 B) is a prior snapshot to fold onto.
 * Replaces: the Kafka Streams restore consumer feeding RocksDB
 * (SurgeStateStoreConsumer.scala:57-76).
 */
int32_t surge_replay_load_csr(surge_replay_handle* h, const int64_t* seg_off, int64_t n_agg,
                              const void* events, int64_t n_events, const void* init_state);

/* Zero-copy variant: all pointers are device pointers that stay valid until the
 * next load/bind or destroy.  d_state_out (nullable) receives the n_agg x 64 B
 * result; when NULL the handle allocates it. */
int32_t surge_replay_bind_device_csr(surge_replay_handle* h, const int64_t* d_seg_off,
                                     int64_t n_agg, const void* d_events, int64_t n_events,
                                     const void* d_init_state, void* d_state_out);

/* ---- fold (R2) ----------------------------------------------------------------
 * state[a] = events[seg_off[a]..seg_off[a+1]).foldLeft(init[a])(handleEvent)
 * (CommandModels.scala:26).  Result stays device-resident.  Asynchronous on the
 * handle's stream. */
int32_t surge_replay_fold(surge_replay_handle* h, int32_t algo);
/* Builds whatever per-log index `algo` needs for the bound log (length order, chunk table, tile-major copy) without
 * folding, so a host can pay it when the log is bound rather than inside its first fold.  A later fold with the same
 * algo reuses it.  Asynchronous except for two small device->host size reads. */
int32_t surge_replay_prepare(surge_replay_handle* h, int32_t algo);
int32_t surge_replay_layout_info(surge_replay_handle* h, surge_replay_layout_info_t* out);
/* The row order of the bound log's index, copied to the host (diagnostics and tests: the order is part of no result):
 * SURGE_ALGO_SORTED — the kernel-facing aggregates (the non-empty ones, in aggregate order) by descending event count,
 * equal counts in aggregate order; SURGE_ALGO_CHUNKED — per virtual row of the chunk table, in the order the lanes take
 * them, its first event slot (rows by descending length, equal lengths in emission order).  order_out holds up to `capacity`
 * entries; *n_out = the number of rows.  SURGE_E_STATE when the bound log has no such index yet (surge_replay_prepare). */
int32_t surge_replay_index_order(surge_replay_handle* h, int32_t algo, int64_t* order_out, int64_t capacity, int64_t* n_out);

/* Streaming micro-batch (K3): for every group g,
 *   state[group_agg[g]] = events[group_off[g]..group_off[g+1]).foldLeft(state[group_agg[g]])(handleEvent)
 * Groups are order-preserving and each aggregate appears at most once per batch.
 * Replaces: PersistentActor.doApplyEvent -> callEventHandler for many aggregates
 * at once (PersistentActor.scala:245-272).  Host buffers. */
int32_t surge_replay_append_fold(surge_replay_handle* h, const int64_t* group_agg,
                                 const int64_t* group_off, int64_t n_groups,
                                 const void* events, int64_t n_events);
/* Ungrouped variant: n events in topic (offset) order, event i belongs to aggregate agg_idx[i].  The
 * library groups them by aggregate ON THE DEVICE (stable radix sort of (index, position), head scan, gather;
 * order inside an aggregate is kept), then runs the same micro-batch fold.  This is the shape a consumer of
 * the events topic has after interning record keys "<aggregateId>:<seq>" (TestBoundedContext.scala:122-124).
 * Host buffers are staged through two pinned areas used in turn (the host fills one while the other is still being
 * copied) and their indices are range-checked before anything is enqueued (SURGE_E_RANGE, batch not applied).
 * The _device variant takes device pointers and NEVER waits for the device (v1 handles): group-by, plan and fold are
 * enqueued back to back — the plan reads the group count where the group-by left it.  A device batch with an index out of
 * range is skipped as a whole on the device (it folds nothing) and the skip is reported, once, as SURGE_E_RANGE by the
 * next surge_replay_synchronize. */
int32_t surge_replay_append_events(surge_replay_handle* h, const int64_t* agg_idx, const void* events,
                                   int64_t n_events);
int32_t surge_replay_append_events_device(surge_replay_handle* h, const int64_t* d_agg_idx, const void* d_events,
                                          int64_t n_events);
int32_t surge_replay_append_fold_device(surge_replay_handle* h, const int64_t* d_group_agg,
                                        const int64_t* d_group_off, int64_t n_groups,
                                        const void* d_events, int64_t n_events);

/* ---- the device packer: decoded fetches -> a bound CSR log (SURVEY §8f N1 "ingest -> CSR pack") ---------------------------
 * A recovery that folds the events topic ONCE wants the whole topic as one log the lane-per-row kernels fold (SORTED /
 * CHUNKED / TILED: the kernels of the headline numbers), not a flat append per fetch.  surge_replay_stage_events_device
 * appends n decoded events — topic (offset) order, event i of aggregate d_agg_idx[i], as surge_device_decoder delivers them
 * fetch by fetch — to a staging log in device memory (20 bytes per event; it grows).  surge_replay_pack_staged turns
 * everything staged into a CSR log over n_agg aggregates — a stable radix sort of (aggregate, position) pairs, seg_off by a
 * search of the sorted aggregates, one gather of the 16-byte events: inside an aggregate the topic's order is kept, which is
 * all foldLeft needs (CommandModels.scala:26); aggregates without events get empty segments — releases the staging log and
 * binds the result like surge_replay_bind_device_csr does (handle-owned buffers, no prior state, a state buffer of the
 * handle's own): surge_replay_fold / _prepare with any algorithm follow, and the log stays bound for re-folds.
 * What it replaces: the per-key grouping Kafka Streams' restore gets for free from RocksDB's key order
 * (SurgeStateStoreConsumer.scala:57-76) and the host-side pack of event objects (surge_amd/log.py::pack_events).
 * Limits: fewer than 2^32 staged events and fewer than 2^32 aggregates per pack; an index >= n_agg fails the pack with
 * SURGE_E_RANGE (nothing bound, the staging log kept).  Both enqueue on the handle's stream; the pack waits for it
 * twice (sizes).  surge_replay_stage_reserve: capacity for n_events staged events up front (no growth copies). */
int32_t surge_replay_stage_reserve(surge_replay_handle* h, int64_t n_events);
int32_t surge_replay_stage_events_device(surge_replay_handle* h, const int64_t* d_agg_idx, const void* d_events, int64_t n_events);
int32_t surge_replay_staged(surge_replay_handle* h, int64_t* n_events_out);
int32_t surge_replay_pack_staged(surge_replay_handle* h, int64_t n_agg);
/* The bound log's device arrays (valid until the next load / bind / pack): seg_off[n_agg + 1], events[n_events] x 16 B. */
int32_t surge_replay_bound_log(surge_replay_handle* h, const int64_t** d_seg_off, const void** d_events, int64_t* n_agg, int64_t* n_events);

/* New aggregates after recovery (the normal Surge case: ids that first appear in a later micro-batch): extends the
 * RESIDENT state to new_n_agg aggregates, the new ones None.  Indices below the old n_agg keep their states.  Only
 * for a state buffer the handle owns (SURGE_E_UNSUPPORTED otherwise).  The bound CSR no longer covers the state
 * afterwards: surge_replay_fold returns SURGE_E_STATE until the next load/bind; append_*, get, gather, snapshot and
 * the encoders keep working. */
int32_t surge_replay_grow(surge_replay_handle* h, int64_t new_n_agg);

/* ---- read (S2) ------------------------------------------------------------------
 * Point read of the recovered state; serves
 * AggregateStateStoreKafkaStreams.getAggregateBytes (…KafkaStreams.scala:83-85)
 * after the plugin's writeState re-attaches the aggregate id.  *present_out = 0
 * means None (KTable miss / tombstone).  Thread-safe once surge_replay_snapshot
 * has published a host mirror for the current fold epoch. */
int32_t surge_replay_get(surge_replay_handle* h, int64_t agg_idx, void* state64_out,
                         uint8_t* present_out);

/* Bulk device->host copy of all states (+ presence bytes, nullable); also
 * publishes the host mirror used by surge_replay_get. */
int32_t surge_replay_snapshot(surge_replay_handle* h, void* states_out, uint8_t* present_out);

/* Bulk point read: states_out[i] = state[agg_idx[i]] for n host-side indices (device gather + one D2H).
 * Serves the state-topic snapshot writer (SurgeModel.serializeState, SurgeModel.scala:57-65): after a
 * micro-batch only the touched aggregates need new snapshot records. */
int32_t surge_replay_gather(surge_replay_handle* h, const int64_t* agg_idx, int64_t n, void* states_out);

/* Device pointer of the resident n_agg x 64 B state array (for the host layer's
 * RCCL all-gather of the final snapshot). */
int32_t surge_replay_device_state(surge_replay_handle* h, void** d_states, int64_t* n_agg);

/* ---- serialized state -------------------------------------------------------------
 * GPU-side encoder from the fixed 64-byte state to the plugin's serialized form, for bulk snapshot
 * publishing (10 M aggregates are ~1 GB of JSON).  The shape is declared as a small template, e.g. the
 * Counter fixture's play-json text  {"aggregateId":"<id>","count":N,"version":N}
 * (TestBoundedContext.scala:15-16,127-129) = LITERAL KEY LITERAL I32@0 LITERAL I32@4 LITERAL.
 * KEY is the aggregate id as a JSON string with Jackson's default escaping.  Absent (None) and poisoned
 * aggregates get zero bytes (out_off[a+1] == out_off[a]): the caller writes a tombstone / nothing.
 * A Double field (BankAccount.balance, BankAccountSurgeModel.scala:26-28) is written as play-json 2.9.2 writes a Scala
 * Double: the shortest decimal digits that round back to it (java.lang.Double.toString's contract) formatted by
 * java.math.BigDecimal's rules after stripTrailingZeros — 100.0 -> 100, 0.1 -> 0.1, 1.5E20 -> 1.5E+20, 1.0E-7 -> 1E-7,
 * -0.0 -> 0 (surge_amd/csrc/f64_text.h states the rule and its sources).  NaN / infinities are not JSON numbers — the
 * reference's writeState throws: such an aggregate gets zero bytes and the call returns SURGE_E_UNSUPPORTED after
 * encoding everything else (the output is valid; last_error carries the count).
 * The model's immutable string fields that never enter the fold (BankAccount.accountOwner / securityCode, set once by
 * the creating event) are served from up to SURGE_JSON_STRING_COLUMNS side string tables on the device. */
#define SURGE_JP_LITERAL 0u /* bytes literals[lit_off .. lit_off + lit_len)          */
#define SURGE_JP_KEY     1u /* "<aggregate id>" escaped                               */
#define SURGE_JP_I32     2u /* decimal int32 at state byte offset field_offset        */
#define SURGE_JP_U32     3u
#define SURGE_JP_I64     4u
#define SURGE_JP_F64     5u /* play-json Double text of the f64 at state byte offset field_offset */
#define SURGE_JP_STR     6u /* "<side string column field_offset of this aggregate>" escaped (surge_replay_set_encode_strings) */
#define SURGE_JSON_MAX_PARTS 16
#define SURGE_JSON_STRING_COLUMNS 4
typedef struct surge_json_template {
  uint32_t n_parts;
  struct { uint32_t kind, field_offset, lit_off, lit_len; } part[SURGE_JSON_MAX_PARTS];
  uint8_t literals[256];
} surge_json_template;

/* Side string column `column` (0 .. SURGE_JSON_STRING_COLUMNS-1) for SURGE_JP_STR parts: aggregate a's string is
 * d_utf8[d_off[a] .. d_off[a+1]) (n_agg + 1 offsets; device pointers that stay valid while encoders run; NULL removes it). */
int32_t surge_replay_set_encode_strings(surge_replay_handle* h, int32_t column, const uint8_t* d_utf8, const int64_t* d_off);

/* The Double text above on the host: returns its length (0 for NaN / infinity), at most 26 bytes written to out when
 * capacity allows; _many formats n values back to back (out_off: n + 1 offsets, nullable) and returns the total. */
int32_t surge_format_f64_json(uint64_t f64_bits, uint8_t* out, int32_t capacity);
int64_t surge_format_f64_json_many(const uint64_t* f64_bits, int64_t n, uint8_t* out, int64_t capacity, int64_t* out_off);

/* d_keys_utf8 / d_key_off (n_agg + 1 entries): the key table on the device.  Two passes: lengths ->
 * exclusive scan into d_out_off (n_agg + 1) -> bytes into d_out.  *total_bytes_out (host) receives the
 * total; when it exceeds out_capacity nothing is written and SURGE_E_RANGE is returned (call again with
 * a larger buffer). */
int32_t surge_replay_encode_json(surge_replay_handle* h, const surge_json_template* tmpl, const uint8_t* d_keys_utf8,
                                 const int64_t* d_key_off, uint8_t* d_out, int64_t out_capacity, int64_t* d_out_off,
                                 int64_t* total_bytes_out);

/* Same passes, but every emitted value is wrapped in the multilanguage module's protobuf message
 *   message State { string aggregateId = 1; bytes payload = 2; }   (multilanguage-protocol.proto:7-10)
 * = what GenericSurgeCommandBusinessLogic.aggregateWriteFormatting stores (pbState.toByteArray,
 * GenericSurgeCommandBusinessLogic.scala:36-39): 0x0A varint(len id) id 0x12 varint(len payload) payload, with
 * payload = the template's text (the SDK's serialized state) and the raw UTF-8 aggregate id; proto3 omits an
 * empty field.  None / poisoned aggregates still get zero bytes. */
int32_t surge_replay_encode_protobuf_state(surge_replay_handle* h, const surge_json_template* payload_tmpl,
                                           const uint8_t* d_keys_utf8, const int64_t* d_key_off, uint8_t* d_out,
                                           int64_t out_capacity, int64_t* d_out_off, int64_t* total_bytes_out);

/* ---- snapshot publishing (SURVEY §8f N2 x N3) ---------------------------------------------------------------
 * What the state topic needs after a replay / a run of micro-batches, relative to the last COMMITTED snapshot of this
 * handle: the reference publishes a state record only when the state changed (PersistentActor.scala:212 for commands,
 * :255-257 for ApplyEvents), `null` (a tombstone) when it became None (SurgeModel.scala:62).
 *   d_kind_out[a] = SURGE_SNAP_SKIP       unchanged since the last commit, or POISONED (nothing the JVM fold ever had)
 *                   SURGE_SNAP_VALUE      changed and Some: publish writeState(state)
 *                   SURGE_SNAP_TOMBSTONE  changed and None: publish a null value
 * The first call compares against "nothing published" (all None).  commit != 0 makes the current states the new
 * baseline for the aggregates reported.  surge_replay_set_encode_filter(h, d_kind) then restricts the encoders below
 * to the SURGE_SNAP_VALUE aggregates (NULL removes the filter), so only what is published is encoded and copied. */
#define SURGE_SNAP_SKIP      0
#define SURGE_SNAP_VALUE     1
#define SURGE_SNAP_TOMBSTONE 2
int32_t surge_replay_snapshot_delta(surge_replay_handle* h, uint8_t* d_kind_out, int64_t* n_values_out, int64_t* n_tombstones_out,
                                    int32_t commit);
int32_t surge_replay_set_encode_filter(surge_replay_handle* h, const uint8_t* d_kind);
/* The second half of a two-step publish: surge_replay_snapshot_delta(commit = 0) -> encode -> produce -> and only once the
 * records are safely out (the reference treats a state as published when the producer acknowledged it) make the states
 * of the aggregates d_kind reports (the array that delta call filled) the new baseline.  No fold / append / grow may run
 * on the handle between the two calls: the commit copies the states as they are NOW, so it is refused with SURGE_E_STATE
 * when the handle's fold epoch or aggregate count moved since that delta (take a new delta).  A publisher that wants the
 * store to keep folding while its records travel uses commit = 1 + surge_replay_snapshot_invalidate instead. */
int32_t surge_replay_snapshot_commit(surge_replay_handle* h, const uint8_t* d_kind);
/* The other way round, for a publisher that frames and produces in the background while the store keeps folding: it
 * commits with the delta (commit = 1: the baseline is exactly what was encoded) and, should the records not make it out,
 * calls this with the same d_kind — the reported aggregates then differ from their baseline again and the next delta
 * re-emits them (at-least-once, like a producer retry).  d_kind is read up to the aggregate count of that delta call;
 * folds, appends and grows in between are fine. */
int32_t surge_replay_snapshot_invalidate(surge_replay_handle* h, const uint8_t* d_kind);

/* ---- shard map (R15) --------------------------------------------------------------
 * surge_replay_partition_hash:  part_out[i] = abs(MurmurHash3.stringHash(str_i) % n_partitions)
 *   = KafkaPartitionProvider.partitionForKey(partitionByString, numberOfPartitions), KafkaPartitioner.scala:8 —
 *   the WHOLE string is hashed, exactly as there (StringIdentityPartitioner, :29-31, routes this way).
 * surge_replay_partition_hash_up_to_colon:  the same after PartitionStringUpToColon.partitionBy =
 *   str.takeWhile(_ != ':') (KafkaPartitioner.scala:38-42), i.e. how the default partitioner routes event
 *   keys "<aggregateId>:<seq>" to their aggregate's partition.
 * n strings given as UTF-16 code units utf16[str_off[i] .. str_off[i+1]) (what JVM String.charAt sees).
 * CPU (host buffers) and GPU (K4, device buffers) variants give identical results. */
int32_t surge_replay_partition_hash(const uint16_t* utf16, const int64_t* str_off, int64_t n,
                                    int32_t n_partitions, int32_t* part_out);
int32_t surge_replay_partition_hash_up_to_colon(const uint16_t* utf16, const int64_t* str_off, int64_t n,
                                                int32_t n_partitions, int32_t* part_out);
int32_t surge_replay_partition_hash_device(surge_replay_handle* h, const uint16_t* d_utf16,
                                           const int64_t* d_str_off, int64_t n,
                                           int32_t n_partitions, int32_t* d_part_out);
int32_t surge_replay_partition_hash_up_to_colon_device(surge_replay_handle* h, const uint16_t* d_utf16,
                                                       const int64_t* d_str_off, int64_t n,
                                                       int32_t n_partitions, int32_t* d_part_out);

/* Wire form of a snapshot for the xGMI exchange: only the first 40 bytes of a state carry information (the
 * 24-byte reserved tail is always zero), so shards travel packed (n x 40 B) and are expanded back to the
 * canonical n x 64 B on arrival.  Device pointers; launched on hip_stream (NULL = the handle's stream) so
 * the host can keep them on the collective's side stream. */
#define SURGE_PACKED_STATE_SIZE 40
int32_t surge_replay_pack_states(surge_replay_handle* h, const void* d_states64, int64_t n, void* d_packed40, void* hip_stream);
int32_t surge_replay_unpack_states(surge_replay_handle* h, const void* d_packed40, int64_t n, void* d_states64, void* hip_stream);

/* ---- multi-GPU exchange (SURVEY §8b surge_replay_allgather, §8e) ---------------------------------------
 * The path shards by the reference's own shard map (partition = partitionForKey(aggregateId), gpu = partition %
 * n_gpus; KafkaPartitioner.scala:8,38-42; ownership PartitionAssignments.scala:51-63) and folds with NO
 * communication.  Its one exchange step is the all-gather(v) of the final snapshot: one process per GPU, one
 * communicator rank per handle, RCCL over xGMI.  librccl is dlopen'ed on first use (SURGE_RCCL_LIBRARY overrides
 * the search; a copy already loaded in the process — e.g. PyTorch's — is reused); everything else works without it.
 *
 *   rank 0:  surge_replay_comm_unique_id(id)  -> hand the 128 bytes to every rank over any channel the host has
 *   all:     surge_replay_comm_init(h, rank, world, id)                       (collective, blocking)
 *   all:     surge_replay_comm_counts(h, n_local, counts[world], &max_count)  (collective, every call: all ranks or none)
 *   all:     surge_replay_allgather_snapshot(h, d_states, n_local, d_out, rows_per_rank, slot, mode)
 *              d_out[r * rows_per_rank + i] = state i of rank r (64 B); rows i >= counts[r] are None (zero).
 *              d_states NULL = the handle's resident state.  Asynchronous: runs on the handle's side stream after
 *              everything enqueued so far on its fold stream, so the next fold overlaps it; two slots alternate.
 *   all:     surge_replay_comm_wait(h, slot, host_sync)   make the fold stream (or the host) wait for that slot
 *
 * Transport (mode): SURGE_GATHER_P2P — one ncclSend/ncclRecv pair per peer inside one group: xGMI is a point-to-point
 * full mesh, so every link carries exactly one peer's shard, all links at once; SURGE_GATHER_ALLGATHER — the library's
 * ncclAllGather over max-padded shards.  Both ship the 40-byte wire form (see surge_replay_pack_states): least link
 * traffic, at the price of a pack pass on the sender and an expansion pass (40 -> 64 B for the WHOLE gathered snapshot)
 * on every receiver.  SURGE_GATHER_P2P_RAW — the same per-peer send/recv of the 64-byte states as they are, straight
 * between the state arrays: 1.6x the link bytes, but no pack / expand passes — about a third less HBM traffic per
 * exchange, which is what an exchange overlapped with an HBM-bound fold competes for (v2 handles always travel raw). */
#define SURGE_COMM_ID_BYTES    128
#define SURGE_GATHER_P2P       0
#define SURGE_GATHER_ALLGATHER 1
#define SURGE_GATHER_P2P_RAW   2
int32_t surge_replay_comm_unique_id(uint8_t id_out[SURGE_COMM_ID_BYTES]);
int32_t surge_replay_comm_init(surge_replay_handle* h, int32_t rank, int32_t world, const uint8_t id[SURGE_COMM_ID_BYTES]);
int32_t surge_replay_comm_destroy(surge_replay_handle* h);
/* library = path the RCCL symbols came from (owned by the library); version = ncclGetVersion */
int32_t surge_replay_comm_info(surge_replay_handle* h, int32_t* rank, int32_t* world, int32_t* rccl_version,
                               const char** library);
/* Exchanges the shard sizes: a COLLECTIVE on every call (8-byte all-gather + host sync) — every rank calls it or none does.
 * An exchange (surge_replay_allgather_snapshot) reuses the sizes of the last exchange; a rank's very first exchange does
 * the size exchange implicitly (every rank of a new communicator is in that state).  When ANY rank's n_local may have
 * changed (surge_replay_grow, a new shard), every rank calls surge_replay_comm_counts again first: a rank that passes a
 * different n_local to an exchange without it gets SURGE_E_STATE — it never starts a collective its peers are not in. */
int32_t surge_replay_comm_counts(surge_replay_handle* h, int64_t n_local, int64_t* counts_out, int64_t* max_count_out);
int32_t surge_replay_allgather_snapshot(surge_replay_handle* h, const void* d_states, int64_t n_local, void* d_out,
                                        int64_t rows_per_rank, int32_t slot, int32_t mode);
int32_t surge_replay_comm_wait(surge_replay_handle* h, int32_t slot, int32_t host_sync);
/* Hosts without device pointers (a JVM): pass d_out = NULL to surge_replay_allgather_snapshot and the handle keeps the
 * gathered snapshot of that slot in a buffer it owns (rows_per_rank = the largest shard).  surge_replay_gathered_read
 * waits for the slot's exchange and copies rows [first_row, first_row + n_rows) of rank `rank`'s block to host memory;
 * surge_replay_gathered returns the device pointer and row pitch for hosts that can use them. */
int32_t surge_replay_gathered(surge_replay_handle* h, int32_t slot, void** d_out, int64_t* rows_per_rank);
int32_t surge_replay_gathered_read(surge_replay_handle* h, int32_t slot, int32_t rank, int64_t first_row, int64_t n_rows,
                                   void* states_out);

/* The same exchange for ONE host process that drives several GPUs (a JVM with one handle per device) — the literal
 * SURVEY §8b form `surge_replay_allgather(h[], n_gpus, states_out_per_gpu[])`.  No RCCL and no rendezvous: hs[r] becomes
 * rank r of an in-process group of n, every rank stages its shard in wire form on its own side stream and every
 * destination pulls the n shards with peer copies (hipMemcpyPeerAsync: one xGMI link per source, all of a
 * destination's links at once; handles that share a device copy locally).  One call covers all ranks:
 *   n_local[r]  states rank r contributes (NULL = every handle's whole resident state)
 *   d_out[r]    rank r's copy of the snapshot, n x rows_per_rank x 64 B on ITS device (NULL array = each handle keeps
 *               it, as with surge_replay_allgather_snapshot(d_out = NULL); rows_per_rank is then the largest shard)
 * Asynchronous like the RCCL form; afterwards surge_replay_comm_wait / _gathered / _gathered_read / _comm_info /
 * _comm_counts / _comm_destroy work per handle.  A handle holds either an RCCL rank or an in-process rank, not both
 * (SURGE_E_STATE); the handles must all be v1 or all v2 and must not be used from other threads during the call. */
int32_t surge_replay_allgather(surge_replay_handle* const* hs, int32_t n, const int64_t* n_local, void* const* d_out,
                               int64_t rows_per_rank, int32_t slot);

/* Redirect the fold's output to another device buffer (n_agg x 64 B, 16-byte aligned) without
 * re-analysing the bound log; lets a host double-buffer snapshots under an overlapped all-gather. */
int32_t surge_replay_set_state_out(surge_replay_handle* h, void* d_state_out);

/* ---- measurement --------------------------------------------------------------- */
int32_t surge_replay_stats(surge_replay_handle* h, surge_replay_stats_t* out);
int32_t surge_replay_stats_reset(surge_replay_handle* h);
/* ms_out[i] = HIP-event time of the dominant kernel of the i-th fold since stats_reset (at most 256 are kept; the
 * events sit on the handle's stream right around the launch).  *n_out = how many were written (<= cap). */
int32_t surge_replay_fold_times(surge_replay_handle* h, double* ms_out, int64_t cap, int64_t* n_out);

/* HBM read-stream ceiling probe: reads n_bytes (multiple of 16) from d_src with
 * 16 B/lane loads and returns the HIP-event time of one launch.  Used by
 * bench.py to report the achievable streaming ceiling beside the 8 TB/s spec. */
int32_t surge_replay_stream_probe(surge_replay_handle* h, const void* d_src, int64_t n_bytes,
                                  double* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* SURGE_REPLAY_H */
