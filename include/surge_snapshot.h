/*
 * surge_snapshot.h — C ABI of the state-topic snapshot writer: the step immediately AFTER the
 * fold.  Host-side, no GPU: serialized states in, Kafka record batches out — what an unmodified Surge node then
 * indexes through its normal KTable (SurgeStateStoreConsumer.scala:57-76).
 *
 * One record per published aggregate, exactly what SurgeModel.serializeState builds
 * (modules/command-engine/core/src/main/scala/surge/internal/SurgeModel.scala:57-65):
 *     ProducerRecord(stateTopic, partition, key = aggregateId, value = writeState(state).value | null)
 * framed as message-format-v2 record batches (KIP-98; the same framing include/surge_ingest.h decodes — kafka-clients
 * 3.2.3 is not vendored under /root/reference: parity unpinned, pinned on the format's own CRC-32C and on a round trip
 * through the independent decoder and the test-side writer).  Batches are non-transactional (producerId -1: a bulk
 * snapshot is not part of any command's transaction) and either uncompressed or — like the reference's producer,
 * compression.type = lz4, modules/common/src/main/resources/reference.conf:112 — LZ4: the records section of a batch
 * is one LZ4 frame as kafka-clients writes it (surge_lz4_frame_compress in surge_ingest.h; the tests decode it with
 * liblz4 itself).
 *
 * Which aggregates get a record comes from surge_replay_snapshot_delta (include/surge_replay.h): kind[i] =
 * SURGE_SNAP_SKIP (nothing: state unchanged, PersistentActor.scala:212,257), SURGE_SNAP_VALUE, SURGE_SNAP_TOMBSTONE.
 */
#ifndef SURGE_SNAPSHOT_H
#define SURGE_SNAPSHOT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct surge_snapshot_writer surge_snapshot_writer;

/* One log per state-topic partition; a batch is closed after max_records_per_batch records or max_batch_bytes bytes
 * (0 = defaults: 10 000 records, 1 MiB — `batch.size`-like). */
int32_t surge_snapshot_writer_create(int32_t n_partitions, int32_t max_records_per_batch, int64_t max_batch_bytes,
                                     surge_snapshot_writer** out);
int32_t surge_snapshot_writer_destroy(surge_snapshot_writer* w);
/* Codec of the batches closed from now on (Kafka attribute bits 0-2): NONE (default) or LZ4.  Call between batches
 * (right after create, flush or reset). */
#define SURGE_SNAPSHOT_CODEC_NONE 0
#define SURGE_SNAPSHOT_CODEC_LZ4  3
int32_t surge_snapshot_writer_set_compression(surge_snapshot_writer* w, int32_t codec);
const char* surge_snapshot_writer_last_error(const surge_snapshot_writer* w);

/* For every i in [0, n) with kind[i] != SURGE_SNAP_SKIP append one record to partition[i]:
 *   key   = keys_utf8[key_off[i] .. key_off[i+1])
 *   value = values[val_off[i] .. val_off[i+1])   for SURGE_SNAP_VALUE   (the encoders' output: d_out / d_out_off)
 *           null                                 for SURGE_SNAP_TOMBSTONE
 * in index order (offsets inside a partition follow the index).  kind == NULL means "all VALUE". */
int32_t surge_snapshot_writer_append(surge_snapshot_writer* w, int64_t n, const uint8_t* kind, const int32_t* partition,
                                     const uint8_t* keys_utf8, const int64_t* key_off, const uint8_t* values,
                                     const int64_t* val_off, int64_t timestamp_ms);
/* The same for a COMPACT list of records: record r belongs to aggregate agg_idx[r] (< n_aggregates); kind[r] and
 * val_off[r .. r+1] are per record, partition[] and key_off[] stay the per-aggregate tables.  What a publisher uses when
 * few of many aggregates changed: nothing of size n_aggregates is scanned or copied (BulkSnapshotPublisher compacts
 * the delta on the device). */
int32_t surge_snapshot_writer_append_indexed(surge_snapshot_writer* w, int64_t n, const int64_t* agg_idx, int64_t n_aggregates, const uint8_t* kind,
                                             const int32_t* partition, const uint8_t* keys_utf8, const int64_t* key_off, const uint8_t* values,
                                             const int64_t* val_off, int64_t timestamp_ms);

/* Closes the open batch of every partition (call before reading the bytes). */
int32_t surge_snapshot_writer_flush(surge_snapshot_writer* w);

/* The record batches of one partition written so far (valid until the next append / reset / destroy), the records in
 * them and the next offset of the partition's log. */
int32_t surge_snapshot_writer_partition(const surge_snapshot_writer* w, int32_t partition, const uint8_t** data,
                                        int64_t* len, int64_t* n_records, int64_t* next_offset);

/* Drops the bytes, keeps each partition's next offset (the log continues). */
int32_t surge_snapshot_writer_reset(surge_snapshot_writer* w);

/* ---- the same record batches, framed on the GPU ---------------------------------------------------------------------
 * A bulk publish whose inputs are already in device memory — the delta's kind[] (surge_replay_snapshot_delta), the
 * encoder's text and offsets (surge_replay_encode_states with the delta as filter), the key table, the partition of every
 * aggregate — needs nothing from the host but the CRC-32C of the finished batches: the device selects the changed
 * aggregates, orders them by partition (stable: aggregate index order inside a partition, like the writer above), cuts
 * the batches by the same rule (a batch closes with the record that brings it to max_records or max_bytes) and writes
 * every record and header where it goes; one copy brings the bytes to page-locked host memory, where the batches get
 * their CRCs.  BYTE-IDENTICAL to surge_snapshot_writer_append + _flush on the same input (uncompressed batches;
 * tests/test_frame_gpu.py).  What it is for: the state-topic side of KafkaProducerActorImpl's publish
 * (modules/command-engine/core/src/main/scala/surge/internal/kafka/KafkaProducerActorImpl.scala:421-453) when a whole
 * commit interval's worth of changed aggregates is published at once (config C5). */
typedef struct surge_device_framer surge_device_framer;
/* hip_stream: the stream the inputs are produced on (NULL = the default stream).  max_records_per_batch <= 2^20;
 * 0 = the writer's defaults (10000 records, 1 MiB).  SURGE_E_DEVICE without a usable GPU: there is no CPU fallback
 * (surge_snapshot_writer_* is the host path). */
int32_t surge_device_framer_create(int32_t device_id, void* hip_stream, int32_t n_partitions, int32_t max_records_per_batch,
                                   int64_t max_batch_bytes, surge_device_framer** out);
int32_t surge_device_framer_destroy(surge_device_framer* f);
const char* surge_device_framer_last_error(const surge_device_framer* f);
/* One publish.  Device inputs, all indexed by aggregate: d_kind[n] (SURGE_SNAP_SKIP / VALUE / TOMBSTONE), d_partition[n],
 * d_key_off[n + 1] into d_keys_utf8, d_val_off[n + 1] into d_values (read for VALUE aggregates only; the filtered
 * encoder's output as it is).  Outputs: *bytes_out = the record batches of all partitions, partition after partition, in
 * host memory owned by the framer (valid until its next call); (*part_byte_off_out)[p .. p+1] = partition p's span in
 * it (n_partitions + 1 entries).  Each partition's log continues where the framer's previous call left it
 * (surge_device_framer_next_offsets).  Synchronous.  On an error nothing is advanced. */
int32_t surge_device_framer_frame(surge_device_framer* f, int64_t n_aggregates, const uint8_t* d_kind, const int32_t* d_partition,
                                  const uint8_t* d_keys_utf8, const int64_t* d_key_off, const uint8_t* d_values, const int64_t* d_val_off,
                                  int64_t timestamp_ms, const uint8_t** bytes_out, const int64_t** part_byte_off_out, int64_t* n_records_out,
                                  int64_t* n_batches_out);
int32_t surge_device_framer_next_offsets(const surge_device_framer* f, int64_t* out /* n_partitions */);

#ifdef __cplusplus
}
#endif
#endif /* SURGE_SNAPSHOT_H */
