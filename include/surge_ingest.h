/*
 * surge_ingest.h — C ABI of the events-topic ingest: the step immediately BEFORE
 * the fold.  Kafka record batches of one partition of the events topic in, records in offset order with
 * their aggregate index out (ready for surge_replay_load_csr / surge_replay_append_events): entirely on
 * the host, or — "device decode" below — framed on the host and decoded on the GPU.
 *
 * What it restates (third party, NOT vendored under /root/reference: org.apache.kafka:kafka-clients 3.2.3,
 * message format v2 / KIP-98; LZ4 frame format — "parity unpinned"):
 *   - RecordBatch v2 framing, CRC-32C over [attributes .. end], zig-zag varint records
 *   - compression.type = lz4 (the reference's producer setting,
 *     modules/common/src/main/resources/reference.conf:112) and none
 *   - isolation.level = read_committed (modules/common/src/main/scala/surge/kafka/streams/
 *     SurgeStateStoreConsumer.scala:38): transactional batches are held back until their producer's
 *     COMMIT / ABORT control marker; aborted ones are dropped; nothing after an open transaction is
 *     delivered (last-stable-offset order); control batches are never delivered
 *   - the per-flush transaction the reference writes: events..., extra records, state
 *     (modules/command-engine/core/src/main/scala/surge/internal/kafka/KafkaProducerActorImpl.scala:421-453)
 *   - the producer's "flush" record, empty key and empty value (KafkaProducerActorImpl.scala:322-329), is skipped
 *   - aggregate id = record key up to the first ':' (PartitionStringUpToColon,
 *     modules/common/src/main/scala/surge/kafka/KafkaPartitioner.scala:38-42; event keys are "<id>:<seq>",
 *     TestBoundedContext.scala:122-124)
 */
#ifndef SURGE_INGEST_H
#define SURGE_INGEST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SURGE_INGEST_READ_UNCOMMITTED 0
#define SURGE_INGEST_READ_COMMITTED   1

/* status codes are surge_replay.h's: 0 OK, -1 INVALID, -2 STATE, -3 DEVICE, -4 NOMEM, -5 UNSUPPORTED; plus: */
#define SURGE_E_CORRUPT (-7) /* bad magic / CRC mismatch / malformed varint / bad LZ4 stream */

typedef struct surge_ingest surge_ingest;

typedef struct surge_ingest_record {
  int64_t offset;      /* Kafka offset of the record                                   */
  int64_t agg_idx;     /* dense index of the aggregate id (key up to ':') in the key table */
  int64_t key_off;     /* span of the full key in the byte arena (surge_ingest_arena)   */
  int32_t key_len;     /* -1 = null key                                                */
  int32_t value_len;   /* -1 = null value (tombstone)                                  */
  int64_t value_off;
} surge_ingest_record;

int32_t surge_ingest_create(int32_t isolation_level, surge_ingest** out);
int32_t surge_ingest_destroy(surge_ingest* g);
const char* surge_ingest_last_error(const surge_ingest* g);

/* Feed bytes of consecutive record batches (a fetch response's record set / a log segment).  Whole
 * batches are consumed; *consumed_out tells how many bytes were (a trailing partial batch is left for
 * the next call, exactly like a fetch that cuts a batch).  Decoded records accumulate until drained.
 * On a failure in batch k (CRC, codec, malformed record) the status is returned with *consumed_out = the
 * byte offset of batch k: batches 0..k-1 of the buffer stay decoded and queued, so a caller must not feed
 * them again. */
int32_t surge_ingest_feed(surge_ingest* g, const uint8_t* data, int64_t len, int64_t* consumed_out);
/* Host threads that verify the CRC-32C of the batches of one feed (default 1; 1 .. 64).  A batch's checksum depends on
 * its own bytes only, so a feed of at least 1 MiB first verifies all its whole batches in parallel and then walks them
 * in order as before: same results, same errors at the same batches.  What it is for: uncompressed topics, where the
 * CRC over the fetch (11 GB/s per thread) is the longest stage between the wire and the GPU. */
int32_t surge_ingest_set_threads(surge_ingest* g, int32_t n_threads);

/* Records that are deliverable now (committed / non-transactional, before any open transaction). */
int64_t surge_ingest_ready(const surge_ingest* g);

/* Pops up to max deliverable records in offset order.  Spans point into the arena, valid until the next
 * feed/drain/destroy. */
int32_t surge_ingest_drain(surge_ingest* g, int64_t max, surge_ingest_record* out, int64_t* n_out);
const uint8_t* surge_ingest_arena(const surge_ingest* g);

/* Convenience for GPU-ready topics whose record value IS the 16-byte fixed event (surge_event16): pops
 * records straight into the arrays surge_replay_append_events / a CSR packer take.  A value of any other
 * length is an error (SURGE_E_INVALID). */
int32_t surge_ingest_drain_fixed16(surge_ingest* g, int64_t max, int64_t* agg_idx_out, void* events16_out,
                                   int64_t* offsets_out, int64_t* n_out);

/* ---- record VALUES as the reference's plugins write them: JSON text -> the 16-byte fixed event -----------------------
 * The reference's event writers are `Json.toJson(evt).toString().getBytes()` over play-json case-class formats
 * (TestBoundedContext.scala:42-49,122-124; BankAccountSurgeModel.scala:30-32): one flat JSON object per event, and for a
 * sealed family a discriminator field naming the case class.  A template tells the decoder which discriminator value is
 * which surge_event16.type and which fields carry the sequence number and the payload; field order, extra fields and
 * the spelling of numbers do not matter (play-json is not under /root/reference: its exact text is parity-unpinned, the
 * decoder does not depend on it).  The reference has no event READER at all (SurgeFormatting.scala:9-11 only writes):
 * this is additive, the mirror image of the GPU state encoders of surge_replay.h. */
#define SURGE_EVJ_MAX_TYPES 16
#define SURGE_EVJ_NAME      64
#define SURGE_EVJ_ARG_NONE  0
#define SURGE_EVJ_ARG_I32   1 /* JSON integer that fits a JVM Int -> payload low word (high word zero)              */
#define SURGE_EVJ_ARG_F64   2 /* JSON number -> IEEE double, correctly rounded (BigDecimal.doubleValue semantics) */
typedef struct surge_event_json_type {
  char     name[SURGE_EVJ_NAME];      /* discriminator value selecting this entry, NUL-terminated ("" if none)   */
  uint32_t event_type;                /* surge_event16.type to emit                                              */
  uint32_t arg_kind;                  /* SURGE_EVJ_ARG_*                                                         */
  char     seq_field[SURGE_EVJ_NAME]; /* integer field -> surge_event16.seq ("" = 0)                             */
  char     arg_field[SURGE_EVJ_NAME]; /* field -> payload ("" with SURGE_EVJ_ARG_NONE)                           */
} surge_event_json_type;
typedef struct surge_event_json_template {
  uint32_t n_types;
  uint32_t reserved;
  char     discriminator[SURGE_EVJ_NAME]; /* e.g. "_type"; "" = every value is types[0]                            */
  surge_event_json_type types[SURGE_EVJ_MAX_TYPES];
} surge_event_json_template;
int32_t surge_event_json_validate(const surge_event_json_template* t);
/* One record value -> one 16-byte event.  0 or SURGE_E_CORRUPT (malformed JSON, unknown type, missing field, a number
 * that is not what the template says); the reason is in surge_event_json_last_error() (thread-local). */
int32_t surge_event_json_decode(const surge_event_json_template* t, const uint8_t* value, int64_t len, void* event16_out);
const char* surge_event_json_last_error(void);
/* A JSON number -> the bits of the nearest IEEE double, ties to even (what BigDecimal(text).doubleValue gives play-json):
 * the Eisel-Lemire algorithm (surge_amd/csrc/f64_parse.h — the same code decodes Doubles on the device), with strtod
 * for the values it reports as undecidable.  0: decided by the fast path; 1: decided by strtod; SURGE_E_CORRUPT: not a
 * JSON number. */
int32_t surge_parse_f64_json(const uint8_t* text, int64_t len, uint64_t* bits_out);
/* Like surge_ingest_drain_fixed16 for topics whose values are JSON events: pops up to max deliverable records and
 * decodes each value through the template — no per-record work in the host language.  Nothing is popped when a value
 * does not decode (SURGE_E_CORRUPT; surge_ingest_last_error names the record's offset and the reason). */
int32_t surge_ingest_drain_json(surge_ingest* g, int64_t max, const surge_event_json_template* tmpl, int64_t* agg_idx_out,
                                void* events16_out, int64_t* offsets_out, int64_t* n_out);

/* ---- device decode (N1 on the GPU) ---------------------------------------------------------------------------------------
 * Per-record work — parsing the varint-framed records, interning the aggregate ids, decoding the event values — is what
 * the host decoder above spends its time on (about 90 ns per record and thread for 16-byte values, 640 ns for JSON), in
 * front of a fold that takes 3 ps per event.  In FRAMES mode the host keeps only what is sequential and cheap per byte —
 * walking the 61-byte batch headers, the CRC-32C, read_committed, LZ4 — and hands the records SECTIONS of the
 * deliverable batches over as they are; a surge_device_decoder turns them into device-resident (aggregate index,
 * 16-byte event, offset) arrays and a device key table: what surge_replay_append_events_device / a CSR build take.
 *   g = surge_ingest_create(isolation | SURGE_INGEST_FRAMES, ..)   feed as usual
 *   surge_ingest_drain_sections(g, ..)  ->  surge_device_decoder_push(d, surge_ingest_arena(g), sections, n)
 * Same results as surge_ingest_drain_fixed16 / _json on the same bytes (tests/test_ingest_gpu.py): same records in the
 * same order, aggregate ids numbered in first-delivered order, flush records skipped, the same values rejected. */
#define SURGE_INGEST_FRAMES     0x100 /* OR into surge_ingest_create's isolation_level */
#define SURGE_INGEST_DEVICE_LZ4 0x200 /* FRAMES, and lz4 batches keep their LZ4 frame: the device decoder decodes the blocks on the
                                         GPU (one wave per 64 KiB block, assembled in LDS); frames kafka-clients would not write —
                                         blocks above 64 KiB, dependent blocks — it decompresses on the host                       */
#define SURGE_INGEST_DEVICE_CRC 0x400 /* FRAMES, and a data batch's CRC-32C is verified ON THE DEVICE: the host checksums only the 40
                                         header bytes the CRC covers in front of the records section (the state of the CRC register
                                         after them travels with the section), the device decoder continues over the section's bytes
                                         where they already are and fails the push (SURGE_E_CORRUPT, nothing delivered, no key kept)
                                         when the result is not the batch's CRC.  The per-byte host work of framing is then gone:
                                         what remains per batch is the header walk and the transaction bookkeeping.  Control batches
                                         (78 bytes) are still verified on the host.  Sections of such a framer carry
                                         SURGE_SECTION_CRC_PENDING in `codec` and are only good for a surge_device_decoder.
                                         The price of verifying late: the framer has already acted on the batch's header (its
                                         transaction bookkeeping) when the device finds the mismatch — like kafka-clients'
                                         CorruptRecordException the failure ends this consumer's pass over the partition: discard
                                         framer and decoder, start again from the last good offsets.                               */
#define SURGE_SECTION_CRC_PENDING 0x100 /* in surge_batch_section.codec: the 8 bytes in front of byte_off hold {the batch's CRC-32C,
                                           the CRC register after the covered header bytes}, little-endian u32 each                  */
#define SURGE_SECTION_CRC_WIRE    0x200 /* ... (in-place framing) the 44 bytes in front of byte_off are the batch's own crc field (big-endian)
                                           and the 40 header bytes it covers, as received: the device runs the whole CRC             */
typedef struct surge_batch_section {
  int64_t byte_off;    /* the batch's records section inside the arena (surge_ingest_arena)            */
  int64_t byte_len;
  int64_t base_offset; /* Kafka offset of the batch's first record                                     */
  int32_t n_records;
  int32_t codec;       /* low byte 0: the records themselves; 3: one LZ4 frame that holds them (SURGE_INGEST_DEVICE_LZ4);
                          | SURGE_SECTION_CRC_PENDING                                                             */
} surge_batch_section;
/* Pops up to max deliverable batches (committed / non-transactional, before any open transaction), in offset order.
 * SURGE_E_STATE on a decoder that was not created in FRAMES mode.
 * Lifetime of the spans: a FRAMES decoder rotates through six arenas, one per feed, so the sections handed out here
 * (and the address surge_ingest_arena returned right after this drain) stay valid THROUGH the next five feeds — host
 * threads can frame fetches i + 1 .. i + 5 (feed, drain_sections, surge_ingest_arena) while a device decoder's pushes of
 * fetch i .. i + 4 are still in flight (surge_device_decoder_push_async).  The handle itself is for one thread at a
 * time; what the other threads touch is only the arena memory.  Batches still queued at a feed (open transactions) move to
 * the new arena with it. */
int32_t surge_ingest_drain_sections(surge_ingest* g, int64_t max_sections, surge_batch_section* out, int64_t* n_out);

/* A consumer that is assigned several partitions gets, per fetch response, the next bytes of each of them; transactions,
 * last stable offsets and cut batches are per partition, so each partition needs a framer of its own.  A group owns n
 * FRAMES handles and, instead of an arena per handle, six rotating SLABS; a feed lays partition 0's records sections,
 * then partition 1's ... out in the next slab (every partition's slice is sized for the feed up front; batches a
 * partition still holds — open transactions — move along), so what comes back is one array of sections, partition after
 * partition, whose spans index one buffer: one part of surge_device_decoder_push_parts_async, one host-to-device copy.
 * A slab stays as it is through the next five feeds.  data[p] / len[p]: partition p's bytes of this fetch response (len
 * 0: nothing, the partition is only carried along); consumed_out (nullable, n entries); sections_out holds up to
 * max_sections entries: the total of len[] / 61 + surge_ingest_group_queued_sections() is always enough (a feed also
 * delivers what earlier feeds left queued — a transaction whose COMMIT marker arrives in this one).  `threads` host
 * threads frame the partitions side by side: the calling thread + a pool the group keeps for its lifetime.
 * SURGE_INGEST_DEVICE_LZ4 / isolation level as in surge_ingest_create (FRAMES is implied; without DEVICE_LZ4 the host
 * decompresses lz4 batches into the slab, sized by trial).
 * A feed is ALL OR NOTHING: when a partition fails (its status is returned, its message in
 * surge_ingest_group_last_error) or sections_out is too small (SURGE_E_INVALID with the count the feed delivers in
 * *n_sections_out), every partition's framer is put back exactly as it was before the call — nothing consumed, nothing
 * drained, consumed_out all 0 — so the caller can feed again (a larger table; without the failing partition's bytes). */
typedef struct surge_ingest_group surge_ingest_group;
int32_t surge_ingest_group_create(int32_t n_partitions, int32_t isolation_level, surge_ingest_group** out);
int32_t surge_ingest_group_destroy(surge_ingest_group* g);
const char* surge_ingest_group_last_error(const surge_ingest_group* g);
int32_t surge_ingest_group_feed(surge_ingest_group* g, const uint8_t* const* data, const int64_t* len, int32_t threads, int64_t* consumed_out,
                                int64_t max_sections, surge_batch_section* sections_out, int64_t* n_sections_out, const uint8_t** slab_out);
/* Framing IN PLACE: where the consumer can choose where a fetch response lands (a socket read, a JNI direct buffer), it
 * receives straight into the group's next page-locked slab: receive_buffer returns room for `bytes` bytes there (behind the few
 * sections the partitions carry over — open transactions); the feed that follows, with every data[p] inside that room,
 * copies NOTHING: the sections it hands out are the received bytes themselves.  With SURGE_INGEST_DEVICE_LZ4 |
 * SURGE_INGEST_DEVICE_CRC the host then reads a batch's 61-byte header and not one byte more (sections carry
 * SURGE_SECTION_CRC_WIRE: the device runs the whole CRC-32C over the header bytes and the section as received).  A feed whose
 * data lies elsewhere frames by copy as before.  lz4 batches need SURGE_INGEST_DEVICE_LZ4 in place. */
int32_t surge_ingest_group_receive_buffer(surge_ingest_group* g, int64_t bytes, uint8_t** buf_out);
/* ... and for a host whose fetch responses already lie somewhere else: receive_buffer for the sum of len[], then the one copy the
 * socket read would have made — partition after partition, on `threads` threads — placed_out[p] = where partition p's bytes
 * now are (NULL for len[p] == 0): what the in-place feed takes as data[p]. */
int32_t surge_ingest_group_receive_copy(surge_ingest_group* g, const uint8_t* const* data, const int64_t* len, int32_t threads, const uint8_t** placed_out);
/* What the group's host threads have cost so far, as thread CPU time (not wall): out[0] seconds inside receive_copy's copies,
 * out[1] seconds framing (surge_ingest_group_feed's per-partition work: headers, transactions — and the sections' copy and
 * CRC-32C where those still run on the host). */
int32_t surge_ingest_group_cpu_seconds(const surge_ingest_group* g, double out[2]);
/* The group's six slabs: bytes allocated so far, and whether they come from an allocator the host set
 * (surge_ingest_group_use_pinned_slabs: page-locked memory — the footprint a host has to budget per consumer). */
int32_t surge_ingest_group_slab_bytes(const surge_ingest_group* g, int64_t* bytes_out, int32_t* custom_allocator_out);
int64_t surge_ingest_group_queued_sections(const surge_ingest_group* g); /* batches the partitions hold from earlier feeds (upper bound of what the next feed delivers beyond its own) */
int32_t surge_ingest_group_counters(const surge_ingest_group* g, int64_t out[8]); /* surge_ingest_counters, summed */
int32_t surge_ingest_group_set_allocator(surge_ingest_group* g, void* (*alloc)(size_t), void (*release)(void*));
int32_t surge_ingest_group_use_pinned_slabs(surge_ingest_group* g); /* page-locked slabs (SURGE_E_DEVICE without a HIP runtime) */

/* Where the arena's memory comes from (before the first feed; NULL / NULL = malloc / free).  surge_ingest_use_pinned_arena
 * makes it page-locked host memory of the HIP runtime: a device decoder then copies the sections to the GPU straight out
 * of the arena (no staging copy).  SURGE_E_DEVICE without a usable HIP runtime. */
int32_t surge_ingest_set_allocator(surge_ingest* g, void* (*alloc)(size_t), void (*release)(void*));
int32_t surge_ingest_use_pinned_arena(surge_ingest* g);

typedef struct surge_device_decoder surge_device_decoder;
/* (Calls on the same decoder must not overlap, with one exception made for throughput: ONE thread may enqueue pushes
 * (push_async / push_parts_async) while ONE other thread finishes them and consumes the results (push_finish*, result,
 * clear, surge_replay_append_decoded*): the host work of framing tables and launches of fetch i + depth then runs beside
 * the interning and the fold of fetch i.) */
/* tmpl == NULL: record values are 16-byte surge_event16; otherwise the reference's JSON event text, decoded by the
 * template with surge_event_json_decode's rules (Doubles correctly rounded on the device — f64_parse.h — the rare value
 * its fast path cannot decide is re-parsed on the host).  hip_stream: the stream the decoder works on (NULL = default). */
int32_t surge_device_decoder_create(int32_t device_id, void* hip_stream, const surge_event_json_template* tmpl, surge_device_decoder** out);
int32_t surge_device_decoder_destroy(surge_device_decoder* d);
const char* surge_device_decoder_last_error(const surge_device_decoder* d);
/* Decodes the sections (spans of `bytes`, a HOST buffer) and APPENDS their records to the device-resident result.  A
 * record that does not decode fails the whole push (SURGE_E_CORRUPT, the message names the record's offset): nothing of
 * the push is appended.  Synchronous. */
int32_t surge_device_decoder_push(surge_device_decoder* d, const uint8_t* bytes, const surge_batch_section* sections, int64_t n_sections);
/* The two halves of a push, for a consumer that keeps the GPU busy across fetches.  push_async enqueues everything of a
 * push that does not touch the key table — the copy to the device, the LZ4 blocks, chaining / parsing the records and
 * decoding their values — on a stream of its own and returns; push_finish interns the keys of the OLDEST unfinished
 * push, appends its records to the result and returns what surge_device_decoder_push would have returned.  Up to five
 * pushes may be unfinished at a time (SURGE_E_STATE beyond that, and from push / push_records while any is), so the
 * copy engine, the latency-bound LZ4 kernels and the compute-bound decode of consecutive fetches overlap on the chip:
 *     push_async(fetch 0) ... push_async(fetch 3);
 *     loop i: push_finish() -> surge_replay_append_decoded_async (fold of fetch i, no host wait); push_async(fetch i + 4)
 * (what surge_amd/store.py and bench.py --workload e2e do; 8.9 - 9.2e8 events/s on a 10^7-aggregate topic).
 * Stage 1 rotates over three LOW-priority streams: the runtime maps streams onto four hardware queues per priority, a queue
 * runs in order, and a stage-1 stream that shares the queue of the decoder's own stream puts a push's interning behind a
 * later push's whole stage 1 — low-priority streams come out of another pool of queues than the decoder's stream and the
 * fold's (SURGE_INGEST_PUSH_PRIORITY, SURGE_INGEST_PUSH_STREAMS: INTEGRATION.md).
 * `bytes` and `sections` must stay valid until the matching push_finish.  push_parts_async takes the sections of
 * several arenas — e.g. one per partition of a fetch response, framed on as many host threads — as ONE push: part p's
 * sections index bytes[p]; records are delivered part after part.  A push_async that fails occupies no slot. */
int32_t surge_device_decoder_push_async(surge_device_decoder* d, const uint8_t* bytes, const surge_batch_section* sections, int64_t n_sections);
int32_t surge_device_decoder_push_parts_async(surge_device_decoder* d, int32_t n_parts, const uint8_t* const* bytes,
                                              const surge_batch_section* const* sections, const int64_t* n_sections);
int32_t surge_device_decoder_push_finish(surge_device_decoder* d);
/* push_finish without its closing wait for the device: the results are complete in the order of the decoder's stream
 * (hand them over with surge_replay_append_decoded_async, or synchronise that stream before reading them).  The one
 * wait that remains is the one in the middle of the call — errors, new keys, the record count: what the host decides on. */
int32_t surge_device_decoder_push_finish_async(surge_device_decoder* d);
int32_t surge_device_decoder_pending(const surge_device_decoder* d); /* pushes enqueued and not finished */
/* Optional capacity hint — e.g. the aggregate count of the store's last snapshot: room for n_keys aggregate ids of
 * key_bytes bytes in all (hash table, key arena, offsets), so a recovery does not grow them step by step (every step
 * allocates, copies and frees device memory: tens of milliseconds at 10^7 keys).  Growing beyond it still works. */
int32_t surge_device_decoder_reserve(surge_device_decoder* d, int64_t n_keys, int64_t key_bytes);
/* The same for records that arrive already framed — what a JVM's KafkaConsumer hands over (ConsumerRecord key / value
 * bytes and offset): record i's key is keys[key_off[i] .. key_off[i+1]), its value values[value_off[i] .. value_off[i+1]),
 * offsets nullable (then 0, 1, 2 ..).  A record with an empty key AND an empty value is the producer's flush record and
 * is skipped, as in the wire path. */
int32_t surge_device_decoder_push_records(surge_device_decoder* d, const uint8_t* keys, const int64_t* key_off, const uint8_t* values,
                                          const int64_t* value_off, const int64_t* offsets, int64_t n);
/* Everything appended since the last clear: device arrays of n_records entries (valid until the next push / clear). */
int32_t surge_device_decoder_result(surge_device_decoder* d, int64_t* n_records, const int64_t** d_agg_idx, const void** d_events16,
                                    const int64_t** d_offsets, int64_t* n_keys);
int32_t surge_device_decoder_clear(surge_device_decoder* d); /* drops the records, keeps the key table */
/* Folds everything decoded since the last clear onto the replay handle's resident state and clears it: grows the
 * state for aggregate ids seen for the first time (surge_replay_grow), device group-by + fold
 * (surge_replay_append_events_device), synchronises.  The handle must hold a bound / restored state (an empty one will
 * do: load a CSR of zero aggregates and fold).  *n_events_out / *n_keys_out (nullable): what was folded / the key count. */
struct surge_replay_handle;
int32_t surge_replay_append_decoded(struct surge_replay_handle* h, surge_device_decoder* d, int64_t* n_events_out, int64_t* n_keys_out);
/* The same without a host wait: the handle's stream waits (event) for the decoder's, group-by + fold are enqueued, the
 * decoder is cleared; the next push_finish* waits (event, on the device) for the group-by's last read of the result
 * arrays before it writes them again.  With the handle on a stream of its own (surge_replay_set_stream, non-blocking),
 * interning of fetch i + 1 overlaps the fold of fetch i — and the decoder then rotates its stage 1 over two streams instead
 * of three (the handle's stream is one more for the four hardware queues); on the decoder's stream (the default of both) the
 * host thread simply does not wait for the fold, which measured the same or better.  Errors of the fold surface at the next
 * surge_replay_synchronize. */
int32_t surge_replay_append_decoded_async(struct surge_replay_handle* h, surge_device_decoder* d, int64_t* n_events_out, int64_t* n_keys_out);
/* For a recovery that ends with ONE fold of the whole topic (surge_replay_pack_staged in surge_replay.h): instead of
 * folding, everything decoded since the last clear is appended to the handle's staging log (surge_replay_stage_events_device)
 * and the decoder is cleared.  No host wait: the hand-over is ordered by events exactly like append_decoded_async's.
 * When the topic ends: surge_replay_pack_staged(h, <the decoder's key count>) and one surge_replay_fold. */
int32_t surge_replay_stage_decoded(struct surge_replay_handle* h, surge_device_decoder* d, int64_t* n_events_out, int64_t* n_keys_out);
/* The key table (aggregate ids in first-delivered order): to the host (NULL / NULL = size query), or where it lives on
 * the device (n_keys + 1 offsets; what the GPU state encoders and K4 take). */
int32_t surge_device_decoder_keys(surge_device_decoder* d, uint8_t* utf8_out, int64_t utf8_capacity, int64_t* key_off_out, int64_t* n_keys_out,
                                  int64_t* utf8_bytes_out);
int32_t surge_device_decoder_key_table(surge_device_decoder* d, const uint8_t** d_utf8, const int64_t** d_key_off);
/* [0] records seen, [1] delivered, [2] flush records skipped, [3] Double values re-parsed on the host */
int32_t surge_device_decoder_counters(const surge_device_decoder* d, int64_t out[4]);
/* [0..3] the counters above, [4] re-seeds of the key table's hash function — two different aggregate ids with the same
 * 64-bit hash are detected (every record's key is compared byte for byte with the key its slot stands for) and handled:
 * nothing of the push is committed, the table gets another hash function (every known key re-hashed from the key arena) and
 * the push runs again; after four functions in a row the push fails with SURGE_E_UNSUPPORTED — [5] slots of the key table,
 * [6] pushes that delivered, [7] the hash function in use (0 = the first). */
int32_t surge_device_decoder_stats(const surge_device_decoder* d, int64_t out[8]);

/* Key table: aggregate ids in first-DELIVERED order (a key is interned when its record is drained, so the
 * keys of aborted or still-open transactions never appear). */
int64_t surge_ingest_key_count(const surge_ingest* g);
int32_t surge_ingest_key(const surge_ingest* g, int64_t idx, const char** utf8_out, int64_t* len_out);

/* Counters: [0] batches, [1] records decoded, [2] records delivered, [3] records dropped (aborted),
 * [4] control batches, [5] flush records skipped, [6] bytes decompressed, [7] open transactions. */
int32_t surge_ingest_counters(const surge_ingest* g, int64_t out[8]);

/* Exposed for tests / other bindings. */
uint32_t surge_crc32c(const uint8_t* data, int64_t len);          /* SSE4.2 CRC32 instruction when the CPU has it */
uint32_t surge_crc32c_portable(const uint8_t* data, int64_t len); /* table walk; must always agree with the above */
/* LZ4 frame -> bytes.  Returns the decompressed size, or a negative status (-6: dst too small, -7: not a valid
 * frame, which includes a wrong header checksum — kafka-clients rejects those for message format v2 as well). */
int64_t surge_lz4_frame_decompress(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap);
/* bytes -> LZ4 frame as kafka-clients writes it (version 01, block independence, 64 KiB blocks, header checksum).
 * dst_cap must be at least surge_lz4_frame_bound(n); returns the frame size or a negative status. */
int64_t surge_lz4_frame_bound(int64_t n);
int64_t surge_lz4_frame_compress(const uint8_t* src, int64_t n, uint8_t* dst, int64_t dst_cap);
/* XXH32 (the frame's header checksum is its second byte over the descriptor). */
uint32_t surge_xxh32(const uint8_t* data, int64_t len, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif /* SURGE_INGEST_H */
