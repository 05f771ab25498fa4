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

#ifdef __cplusplus
}
#endif
#endif /* SURGE_SNAPSHOT_H */
