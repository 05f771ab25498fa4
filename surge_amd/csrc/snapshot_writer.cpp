// snapshot_writer.cpp — state-topic snapshot writer, host side of libsurge_replay.so.
// Kafka RecordBatch v2 ENCODER, restated from the published format (KIP-98), the mirror image of ingest.cpp's decoder.
#include <cstdint>
#include <cstring>
#include <new>
#include <atomic>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/surge_ingest.h"    // surge_crc32c
#include "../../include/surge_replay.h"    // SURGE_SNAP_*
#include "../../include/surge_snapshot.h"

namespace {

constexpr int32_t OK = 0, E_INVALID = -1, E_NOMEM = -4, E_RANGE = -6;
constexpr size_t kHeader = 61;  // baseOffset .. recordCount

thread_local std::string g_err;

struct PartitionLog {
  std::vector<uint8_t> bytes;   // closed batches
  std::vector<uint8_t> open;    // records of the batch being filled
  int64_t next_offset = 0;      // offset of the next record
  int64_t base_offset = 0;      // offset of the open batch's first record
  int64_t base_ts = 0, max_ts = 0;
  int32_t open_records = 0;
  int64_t n_records = 0;
};

inline void put_be(std::vector<uint8_t>& v, uint64_t x, int n) {
  for (int i = n - 1; i >= 0; --i) v.push_back((uint8_t)(x >> (8 * i)));
}

// zig-zag varint / varlong (ByteUtils.writeVarlong)
inline void put_varlong(std::vector<uint8_t>& v, int64_t x) {
  uint64_t z = ((uint64_t)x << 1) ^ (uint64_t)(x >> 63);
  while (z >= 0x80) {
    v.push_back((uint8_t)(z | 0x80));
    z >>= 7;
  }
  v.push_back((uint8_t)z);
}

inline int varlong_size(int64_t x) {
  uint64_t z = ((uint64_t)x << 1) ^ (uint64_t)(x >> 63);
  int n = 1;
  while (z >= 0x80) { z >>= 7; ++n; }
  return n;
}

}  // namespace

struct surge_snapshot_writer {
  std::vector<PartitionLog> parts;
  int32_t max_records = 10000;
  int64_t max_bytes = 1 << 20;
  int32_t codec = SURGE_SNAPSHOT_CODEC_NONE;
  std::string err;
};

namespace {

int32_t fail(surge_snapshot_writer* w, int32_t code, const std::string& m) {
  if (w) w->err = m;
  g_err = m;
  return code;
}

void close_batch(PartitionLog& p, int32_t codec) {
  if (p.open_records == 0) return;
  if (codec == SURGE_SNAPSHOT_CODEC_LZ4) {  // the records section becomes ONE LZ4 frame (DefaultRecordBatch + KafkaLZ4BlockOutputStream)
    std::vector<uint8_t> frame((size_t)surge_lz4_frame_bound((int64_t)p.open.size()));
    const int64_t n = surge_lz4_frame_compress(p.open.data(), (int64_t)p.open.size(), frame.data(), (int64_t)frame.size());
    frame.resize((size_t)n);
    p.open.swap(frame);
  }
  std::vector<uint8_t>& o = p.bytes;
  put_be(o, (uint64_t)p.base_offset, 8);
  put_be(o, (uint64_t)(kHeader - 12 + p.open.size()), 4);  // batchLength: everything after this field
  put_be(o, 0, 4);                                         // partitionLeaderEpoch
  o.push_back(2);                                          // magic
  const size_t crc_at = o.size();
  put_be(o, 0, 4);                                         // crc, patched below
  const size_t crc_from = o.size();
  put_be(o, (uint64_t)(codec & 7), 2);                     // attributes: codec in bits 0-2, CreateTime, not transactional
  put_be(o, (uint64_t)(p.open_records - 1), 4);            // lastOffsetDelta
  put_be(o, (uint64_t)p.base_ts, 8);
  put_be(o, (uint64_t)p.max_ts, 8);
  put_be(o, (uint64_t)-1ll, 8);                            // producerId
  put_be(o, (uint64_t)0xffff, 2);                          // producerEpoch -1
  put_be(o, (uint64_t)0xffffffffu, 4);                     // baseSequence -1
  put_be(o, (uint64_t)p.open_records, 4);
  o.insert(o.end(), p.open.begin(), p.open.end());
  const uint32_t crc = surge_crc32c(o.data() + crc_from, (int64_t)(o.size() - crc_from));  // CRC-32C over attributes .. end
  o[crc_at] = (uint8_t)(crc >> 24); o[crc_at + 1] = (uint8_t)(crc >> 16); o[crc_at + 2] = (uint8_t)(crc >> 8); o[crc_at + 3] = (uint8_t)crc;
  p.open.clear();
  p.open_records = 0;
}

}  // namespace

extern "C" {

int32_t surge_snapshot_writer_create(int32_t n_partitions, int32_t max_records_per_batch, int64_t max_batch_bytes,
                                     surge_snapshot_writer** out) {
  if (!out) return fail(nullptr, E_INVALID, "out is NULL");
  *out = nullptr;
  if (n_partitions <= 0 || max_records_per_batch < 0 || max_batch_bytes < 0) return fail(nullptr, E_INVALID, "bad argument");
  surge_snapshot_writer* w = new (std::nothrow) surge_snapshot_writer();
  if (!w) return fail(nullptr, E_NOMEM, "out of host memory");
  try {
    w->parts.resize((size_t)n_partitions);
  } catch (const std::bad_alloc&) {
    delete w;
    return fail(nullptr, E_NOMEM, "out of host memory");
  }
  if (max_records_per_batch > 0) w->max_records = max_records_per_batch;
  if (max_batch_bytes > 0) w->max_bytes = max_batch_bytes;
  *out = w;
  return OK;
}

int32_t surge_snapshot_writer_set_compression(surge_snapshot_writer* w, int32_t codec) {
  if (!w) return fail(nullptr, E_INVALID, "writer is NULL");
  if (codec != SURGE_SNAPSHOT_CODEC_NONE && codec != SURGE_SNAPSHOT_CODEC_LZ4) return fail(w, E_INVALID, "codec must be NONE (0) or LZ4 (3)");
  for (const PartitionLog& p : w->parts)
    if (p.open_records != 0) return fail(w, E_INVALID, "flush before changing the codec (a batch has one codec)");
  w->codec = codec;
  return OK;
}

int32_t surge_snapshot_writer_destroy(surge_snapshot_writer* w) {
  delete w;
  return OK;
}

const char* surge_snapshot_writer_last_error(const surge_snapshot_writer* w) { return w ? w->err.c_str() : g_err.c_str(); }

// record r of the call: aggregate a = idx ? idx[r] : r; kind[r] and val_off[r] are per RECORD, partition[a] and key_off[a]
// per AGGREGATE (for idx == NULL the two numberings coincide)
static int32_t append_core(surge_snapshot_writer* w, int64_t n, const int64_t* idx, int64_t n_agg, const uint8_t* kind, const int32_t* partition,
                           const uint8_t* keys_utf8, const int64_t* key_off, const uint8_t* values, const int64_t* val_off,
                           int64_t timestamp_ms) {
  if (!w) return fail(nullptr, E_INVALID, "writer is NULL");
  if (n < 0) return fail(w, E_INVALID, "negative size");
  if (n == 0) return OK;
  if (!partition || !key_off) return fail(w, E_INVALID, "NULL buffer");
  auto agg = [&](int64_t r) { return idx ? idx[r] : r; };
  const int32_t P = (int32_t)w->parts.size();
  try {
    // pass 1 (validation + a counting sort of the published indices by partition): partitions are independent logs,
    // so pass 2 frames them on several host threads, each partition's records still in index order
    std::vector<int64_t> start((size_t)P + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
      const uint8_t k = kind ? kind[i] : (uint8_t)SURGE_SNAP_VALUE;
      if (k == SURGE_SNAP_SKIP) continue;
      if (k != SURGE_SNAP_VALUE && k != SURGE_SNAP_TOMBSTONE) return fail(w, E_INVALID, "unknown record kind");
      const int64_t a = agg(i);
      if (idx && (a < 0 || a >= n_agg)) return fail(w, E_RANGE, "aggregate index out of range");
      const int32_t pi = partition[a];
      if (pi < 0 || pi >= P) return fail(w, E_RANGE, "partition out of range");
      const int64_t klen = key_off[a + 1] - key_off[a];
      if (klen < 0 || (klen > 0 && !keys_utf8)) return fail(w, E_INVALID, "bad key span");
      if (k == SURGE_SNAP_VALUE) {
        if (!val_off) return fail(w, E_INVALID, "values expected");
        const int64_t vlen = val_off[i + 1] - val_off[i];
        if (vlen < 0 || (vlen > 0 && !values)) return fail(w, E_INVALID, "bad value span");
      }
      start[(size_t)pi + 1] += 1;
    }
    for (int32_t p = 0; p < P; ++p) start[(size_t)p + 1] += start[(size_t)p];
    const int64_t total = start[(size_t)P];
    if (total == 0) return OK;
    std::vector<int64_t> order((size_t)total);
    {
      std::vector<int64_t> cur(start.begin(), start.end() - 1);
      for (int64_t i = 0; i < n; ++i)
        if (!kind || kind[i] != SURGE_SNAP_SKIP) order[(size_t)cur[(size_t)partition[agg(i)]]++] = i;
    }
    auto encode_partition = [&](int32_t pi) {
      PartitionLog& p = w->parts[(size_t)pi];
      const int64_t r_end = start[(size_t)pi + 1];
      constexpr int64_t kAhead = 12;  // a partition's records are scattered over the key / value tables: ask for them early
      for (int64_t r = start[(size_t)pi]; r < r_end; ++r) {
        if (r + 2 * kAhead < r_end) {  // ... and, before that, the table entries that say where they are
          const int64_t j2 = order[(size_t)(r + 2 * kAhead)];
          __builtin_prefetch(key_off + agg(j2));
          if (val_off) __builtin_prefetch(val_off + j2);
        }
        if (r + kAhead < r_end) {
          const int64_t j = order[(size_t)(r + kAhead)];
          const int64_t aj = agg(j);
          if (keys_utf8) __builtin_prefetch(keys_utf8 + key_off[aj]);
          if (values && val_off && (!kind || kind[j] == SURGE_SNAP_VALUE)) {
            __builtin_prefetch(values + val_off[j]);
            __builtin_prefetch(values + val_off[j] + 64);
          }
        }
        const int64_t i = order[(size_t)r];
        const uint8_t k = kind ? kind[i] : (uint8_t)SURGE_SNAP_VALUE;
        const int64_t a = agg(i);
        const int64_t klen = key_off[a + 1] - key_off[a];
        const int64_t vlen = k == SURGE_SNAP_VALUE ? val_off[i + 1] - val_off[i] : -1;
        if (p.open_records == 0) {
          p.base_offset = p.next_offset;
          p.base_ts = p.max_ts = timestamp_ms;
        }
        if (timestamp_ms > p.max_ts) p.max_ts = timestamp_ms;
        const int64_t off_delta = p.next_offset - p.base_offset, ts_delta = timestamp_ms - p.base_ts;
        // record body: attributes, timestampDelta, offsetDelta, key, value, header count
        const int64_t body = 1 + varlong_size(ts_delta) + varlong_size(off_delta) + varlong_size(klen) + klen +
                             varlong_size(vlen) + (vlen > 0 ? vlen : 0) + 1;
        std::vector<uint8_t>& o = p.open;
        put_varlong(o, body);
        o.push_back(0);
        put_varlong(o, ts_delta);
        put_varlong(o, off_delta);
        put_varlong(o, klen);
        if (klen > 0) o.insert(o.end(), keys_utf8 + key_off[a], keys_utf8 + key_off[a + 1]);
        put_varlong(o, vlen);
        if (vlen > 0) o.insert(o.end(), values + val_off[i], values + val_off[i + 1]);
        put_varlong(o, 0);
        p.next_offset += 1;
        p.open_records += 1;
        p.n_records += 1;
        if (p.open_records >= w->max_records || (int64_t)o.size() >= w->max_bytes) close_batch(p, w->codec);
      }
    };
    unsigned hw = std::thread::hardware_concurrency();
    int n_threads = (int)(hw ? hw : 1);
    if (n_threads > 16) n_threads = 16;
    if (n_threads > P) n_threads = P;
    if (total < 50000) n_threads = 1;  // not worth a thread start
    if (n_threads <= 1) {
      for (int32_t p = 0; p < P; ++p) encode_partition(p);
    } else {
      (void)surge_crc32c((const uint8_t*)"", 0);  // initialise the CRC dispatch / tables before the threads race for them
      std::atomic<int32_t> next{0};
      std::atomic<bool> oom{false};
      auto worker = [&]() {
        try {
          for (int32_t p = next.fetch_add(1); p < P; p = next.fetch_add(1)) encode_partition(p);
        } catch (...) {  // bad_alloc, length_error ...: nothing may leave a thread (std::terminate would take the JVM down)
          oom = true;
          next = P;  // the other workers stop at their next draw
        }
      };
      std::vector<std::thread> th;
      bool started_all = true;
      try {
        th.reserve((size_t)n_threads);
        for (int t = 0; t < n_threads; ++t) th.emplace_back(worker);
      } catch (...) {  // std::system_error: the threads that did start finish the work (or the caller's thread does)
        started_all = false;
      }
      if (!started_all && th.empty()) worker();
      for (std::thread& t : th) t.join();
      if (oom) {
        // partitions are left partially appended: the writer must be reset before it is used again
        return fail(w, E_NOMEM, "out of host memory while encoding (call surge_snapshot_writer_reset before appending again)");
      }
    }
  } catch (const std::bad_alloc&) {
    return fail(w, E_NOMEM, "out of host memory while encoding");
  } catch (const std::system_error&) {
    return fail(w, E_NOMEM, "could not start an encoder thread");
  }
  return OK;
}

int32_t surge_snapshot_writer_append(surge_snapshot_writer* w, int64_t n, const uint8_t* kind, const int32_t* partition,
                                     const uint8_t* keys_utf8, const int64_t* key_off, const uint8_t* values,
                                     const int64_t* val_off, int64_t timestamp_ms) {
  return append_core(w, n, nullptr, n, kind, partition, keys_utf8, key_off, values, val_off, timestamp_ms);
}

int32_t surge_snapshot_writer_append_indexed(surge_snapshot_writer* w, int64_t n, const int64_t* agg_idx, int64_t n_aggregates, const uint8_t* kind,
                                             const int32_t* partition, const uint8_t* keys_utf8, const int64_t* key_off, const uint8_t* values,
                                             const int64_t* val_off, int64_t timestamp_ms) {
  if (n > 0 && !agg_idx) return fail(w, E_INVALID, "agg_idx is NULL");
  return append_core(w, n, agg_idx, n_aggregates, kind, partition, keys_utf8, key_off, values, val_off, timestamp_ms);
}

int32_t surge_snapshot_writer_flush(surge_snapshot_writer* w) {
  if (!w) return fail(nullptr, E_INVALID, "writer is NULL");
  try {
    for (PartitionLog& p : w->parts) close_batch(p, w->codec);
  } catch (const std::bad_alloc&) {
    return fail(w, E_NOMEM, "out of host memory while encoding");
  }
  return OK;
}

int32_t surge_snapshot_writer_partition(const surge_snapshot_writer* w, int32_t partition, const uint8_t** data, int64_t* len,
                                        int64_t* n_records, int64_t* next_offset) {
  if (!w) return fail(nullptr, E_INVALID, "writer is NULL");
  if (partition < 0 || partition >= (int32_t)w->parts.size()) return fail(nullptr, E_RANGE, "partition out of range");
  const PartitionLog& p = w->parts[(size_t)partition];
  if (data) *data = p.bytes.data();
  if (len) *len = (int64_t)p.bytes.size();
  if (n_records) *n_records = p.n_records;
  if (next_offset) *next_offset = p.next_offset;
  return OK;
}

int32_t surge_snapshot_writer_reset(surge_snapshot_writer* w) {
  if (!w) return fail(nullptr, E_INVALID, "writer is NULL");
  for (PartitionLog& p : w->parts) {
    p.bytes.clear();
    p.open.clear();
    p.open_records = 0;
    p.n_records = 0;
  }
  return OK;
}

}  // extern "C"
