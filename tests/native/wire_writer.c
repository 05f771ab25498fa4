/* TEST / BENCH INFRASTRUCTURE (not part of the product, shares no code with it): an events topic as the reference's
 * PRODUCER writes it, at scale — the C twin of tests/kafka_wire.py (the two are compared byte for byte in
 * tests/test_wire_writer.py), so that what bench.py --workload e2e and the ingest tests read was framed by code that is
 * not the product's record-batch writer (surge_amd/csrc/snapshot_writer.cpp) and not its LZ4 writer (lz4_frame.cpp).
 *
 * Restated from the published formats only: Kafka message format v2 (KIP-98: 61-byte batch header, CRC-32C from the
 * attributes on, zig-zag varint records, control batches with key {version, type} / value {version, coordinatorEpoch})
 * and the LZ4 frame format 1.6 as kafka-clients' KafkaLZ4BlockOutputStream emits it (FLG 0x60, BD 0x40 = 64 KiB
 * independent blocks, HC = second byte of XXH32 of the descriptor, no content size / checksum, EndMark).
 *
 * Shape (what the reference publishes): every KafkaProducerActor owns one partition and publishes, every flush interval
 * (50 ms, command-engine/core/src/main/resources/reference.conf:20), everything that is pending in ONE transaction —
 * beginTransaction, putRecords, commitTransaction (KafkaProducerActorImpl.scala:421-453; an error aborts it, :441) — with
 * an idempotent lz4 producer whose batches close at batch.size (common/src/main/resources/reference.conf:111-126).  On
 * the partition's log that is: transactional data batch(es) of the flush's records, then a COMMIT (or ABORT) control
 * batch of the same producer id, which takes an offset of its own.
 *
 * gcc -O2 -shared -fPIC tests/native/wire_writer.c -o tests/native/libwire_writer.so */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- CRC-32C (Castagnoli), bytewise table ----------------------------------------------------------------------------- */
static uint32_t crc_table[256];
static int crc_ready = 0;
static void crc_init(void) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    crc_table[i] = c;
  }
  crc_ready = 1;
}
uint32_t surge_test_wire_crc32c(const uint8_t* p, int64_t n) {
  if (!crc_ready) crc_init();
  uint32_t c = 0xFFFFFFFFu;
  for (int64_t i = 0; i < n; ++i) c = (c >> 8) ^ crc_table[(c ^ p[i]) & 0xFF];
  return c ^ 0xFFFFFFFFu;
}

/* ---- XXH32 of a short input (< 16 bytes: the LZ4 frame descriptor) ----------------------------------------------------- */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t surge_test_wire_xxh32_short(const uint8_t* p, int32_t n, uint32_t seed) {
  const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
  uint32_t h = seed + P5 + (uint32_t)n;
  int32_t i = 0;
  for (; i + 4 <= n; i += 4) {
    uint32_t w = (uint32_t)p[i] | ((uint32_t)p[i + 1] << 8) | ((uint32_t)p[i + 2] << 16) | ((uint32_t)p[i + 3] << 24);
    h = rotl32(h + w * P3, 17) * P4;
  }
  for (; i < n; ++i) h = rotl32(h + p[i] * P5, 11) * P1;
  h ^= h >> 15;
  h *= P2;
  h ^= h >> 13;
  h *= P3;
  h ^= h >> 16;
  return h;
}

/* ---- big-endian fields, zig-zag varints --------------------------------------------------------------------------------- */
static uint8_t* be16(uint8_t* p, int32_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; return p + 2; }
static uint8_t* be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; return p + 4; }
static uint8_t* be64(uint8_t* p, uint64_t v) { p = be32(p, (uint32_t)(v >> 32)); return be32(p, (uint32_t)v); }
static uint8_t* varlong(uint8_t* p, int64_t v) {
  uint64_t z = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);
  while (z >= 0x80) { *p++ = (uint8_t)(z | 0x80); z >>= 7; }
  *p++ = (uint8_t)z;
  return p;
}
static int varlong_len(int64_t v) {
  uint64_t z = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);
  int n = 1;
  while (z >= 0x80) { ++n; z >>= 7; }
  return n;
}

/* ---- LZ4 block (greedy, 4-byte hash) and frame ---------------------------------------------------------------------------- */
static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint8_t* lz4_len(uint8_t* d, int64_t rest) {
  while (rest >= 255) { *d++ = 255; rest -= 255; }
  *d++ = (uint8_t)rest;
  return d;
}
/* dst must hold n + n / 255 + 16 bytes */
static int64_t lz4_block(const uint8_t* s, int64_t n, uint8_t* dst) {
  int32_t table[4096];
  uint8_t* d = dst;
  int64_t anchor = 0, i = 0;
  for (int k = 0; k < 4096; ++k) table[k] = -1;
  if (n >= 13) {
    const int64_t mflimit = n - 12, matchlimit = n - 5; /* the last match starts 12 bytes before the end, the last 5 bytes are literals */
    while (i < mflimit) {
      const uint32_t seq = rd32(s + i);
      const uint32_t h = (seq * 2654435761u) >> 20;
      const int64_t cand = table[h];
      table[h] = (int32_t)i;
      if (cand >= 0 && i - cand <= 65535 && rd32(s + cand) == seq) {
        int64_t ml = 4;
        while (i + ml < matchlimit && s[cand + ml] == s[i + ml]) ++ml;
        const int64_t lit = i - anchor;
        *d++ = (uint8_t)(((lit < 15 ? lit : 15) << 4) | (ml - 4 < 15 ? ml - 4 : 15));
        if (lit >= 15) d = lz4_len(d, lit - 15);
        memcpy(d, s + anchor, (size_t)lit);
        d += lit;
        *d++ = (uint8_t)((i - cand) & 0xFF);
        *d++ = (uint8_t)((i - cand) >> 8);
        if (ml - 4 >= 15) d = lz4_len(d, ml - 4 - 15);
        i += ml;
        anchor = i;
      } else {
        ++i;
      }
    }
  }
  const int64_t lit = n - anchor;
  *d++ = (uint8_t)((lit < 15 ? lit : 15) << 4);
  if (lit >= 15) d = lz4_len(d, lit - 15);
  memcpy(d, s + anchor, (size_t)lit);
  d += lit;
  return d - dst;
}
/* dst must hold n + n / 255 + 64 + 8 * (n / 65536 + 1) bytes; returns the frame's length */
int64_t surge_test_wire_lz4_frame(const uint8_t* s, int64_t n, uint8_t* dst) {
  uint8_t* d = dst;
  const uint8_t desc[2] = {0x60, 0x40};
  d[0] = 0x04; d[1] = 0x22; d[2] = 0x4D; d[3] = 0x18; /* magic 0x184D2204, little-endian */
  d[4] = desc[0];
  d[5] = desc[1];
  d[6] = (uint8_t)(surge_test_wire_xxh32_short(desc, 2, 0) >> 8);
  d += 7;
  for (int64_t at = 0; at < n; at += 65536) {
    const int64_t len = n - at < 65536 ? n - at : 65536;
    const int64_t c = lz4_block(s + at, len, d + 4);
    if (c >= len) { /* stored: the high bit of the size word */
      const uint32_t w = (uint32_t)len | 0x80000000u;
      d[0] = (uint8_t)w; d[1] = (uint8_t)(w >> 8); d[2] = (uint8_t)(w >> 16); d[3] = (uint8_t)(w >> 24);
      memcpy(d + 4, s + at, (size_t)len);
      d += 4 + len;
    } else {
      d[0] = (uint8_t)c; d[1] = (uint8_t)(c >> 8); d[2] = (uint8_t)(c >> 16); d[3] = (uint8_t)(c >> 24);
      d += 4 + c;
    }
  }
  d[0] = d[1] = d[2] = d[3] = 0; /* EndMark */
  return d + 4 - dst;
}

/* ---- one record batch ------------------------------------------------------------------------------------------------------- */
#define WIRE_LZ4 1
#define WIRE_TRANSACTIONAL 2
#define WIRE_CONTROL 4

/* Header + payload at `out` (the caller made room: 61 + payload bound).  `recs` = the records section, uncompressed;
 * `scratch` holds an LZ4 frame of it when flags has WIRE_LZ4.  Returns the batch's length. */
static int64_t put_batch(uint8_t* out, int64_t base_offset, int32_t n_records, const uint8_t* recs, int64_t recs_len, int32_t flags, int64_t producer_id,
                         int32_t producer_epoch, int32_t base_sequence, int64_t first_ts, int64_t max_ts, uint8_t* scratch) {
  const uint8_t* payload = recs;
  int64_t payload_len = recs_len;
  if (flags & WIRE_LZ4) {
    payload_len = surge_test_wire_lz4_frame(recs, recs_len, scratch);
    payload = scratch;
  }
  const int32_t attrs = ((flags & WIRE_LZ4) ? 3 : 0) | ((flags & WIRE_TRANSACTIONAL) ? 0x10 : 0) | ((flags & WIRE_CONTROL) ? 0x20 : 0);
  uint8_t* p = out;
  p = be64(p, (uint64_t)base_offset);
  p = be32(p, (uint32_t)(49 + payload_len)); /* batchLength: everything behind this field */
  p = be32(p, 0);                            /* partitionLeaderEpoch */
  *p++ = 2;                                  /* magic */
  uint8_t* crc_at = p;
  p += 4;
  uint8_t* crc_from = p;
  p = be16(p, attrs);
  p = be32(p, (uint32_t)(n_records > 0 ? n_records - 1 : 0)); /* lastOffsetDelta */
  p = be64(p, (uint64_t)first_ts);
  p = be64(p, (uint64_t)max_ts);
  p = be64(p, (uint64_t)producer_id);
  p = be16(p, producer_epoch);
  p = be32(p, (uint32_t)base_sequence);
  p = be32(p, (uint32_t)n_records);
  memcpy(p, payload, (size_t)payload_len);
  p += payload_len;
  be32(crc_at, surge_test_wire_crc32c(crc_from, p - crc_from));
  return p - out;
}

/* The headers section every record written from now on carries (the encoded section as it stands in a record: varint
 * count, then per header varint key length, key, varint value length, value); len = 0: no headers (one byte, the count 0).
 * Kafka clients attach headers per record — the reference's publisher passes SerializedMessage.headers and its tracing
 * context on (SurgeModel.scala:46-52, HeadersHelper.scala:17) — and a reader has to step over them.  At most 255 bytes. */
static uint8_t g_headers[256];
static int64_t g_headers_len = 0;
int32_t surge_test_wire_set_record_headers(const uint8_t* section, int64_t len) {
  if (len < 0 || len > 255) return -1;
  if (len > 0) memcpy(g_headers, section, (size_t)len);
  g_headers_len = len;
  return 0;
}

/* One record: length, attributes 0, timestampDelta, offsetDelta, key, value, headers (none unless set above).  klen / vlen < 0 = null. */
static uint8_t* put_record_h(uint8_t* p, int64_t ts_delta, int32_t offset_delta, const uint8_t* key, int64_t klen, const uint8_t* val, int64_t vlen, int64_t headers_len) {
  const int64_t body = 1 + varlong_len(ts_delta) + varlong_len(offset_delta) + varlong_len(klen) + (klen > 0 ? klen : 0) + varlong_len(vlen) + (vlen > 0 ? vlen : 0) + (headers_len ? headers_len : 1);
  p = varlong(p, body);
  *p++ = 0;
  p = varlong(p, ts_delta);
  p = varlong(p, offset_delta);
  p = varlong(p, klen);
  if (klen > 0) { memcpy(p, key, (size_t)klen); p += klen; }
  p = varlong(p, vlen);
  if (vlen > 0) { memcpy(p, val, (size_t)vlen); p += vlen; }
  if (headers_len) { memcpy(p, g_headers, (size_t)headers_len); p += headers_len; }
  else *p++ = 0; /* headers: 0 (zig-zag of 0) */
  return p;
}

static uint8_t* put_record(uint8_t* p, int64_t ts_delta, int32_t offset_delta, const uint8_t* key, int64_t klen, const uint8_t* val, int64_t vlen) {
  return put_record_h(p, ts_delta, offset_delta, key, klen, val, vlen, g_headers_len); /* a data record */
}

/* The records idx[0 .. n) (NULL: 0 .. n) as ONE batch at `out`; a record's timestampDelta is ts_delta[i] (NULL: 0).
 * Returns the batch's length; out needs 61 + 64 + the LZ4 bound of n * (28 + key + value). */
int64_t surge_test_wire_batch(uint8_t* out, int64_t base_offset, int64_t n, const int64_t* idx, const uint8_t* keys, const int64_t* key_off, const uint8_t* vals,
                              const int64_t* val_off, const int64_t* ts_delta, int32_t flags, int64_t producer_id, int32_t producer_epoch, int32_t base_sequence,
                              int64_t first_ts) {
  int64_t bound = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t r = idx ? idx[i] : i;
    bound += 40 + g_headers_len + (key_off[r + 1] - key_off[r]) + (val_off[r + 1] - val_off[r]);
  }
  uint8_t* recs = (uint8_t*)malloc((size_t)bound + 16);
  uint8_t* scratch = (uint8_t*)malloc((size_t)(bound + bound / 255 + 64 + 8 * (bound / 65536 + 1)));
  if (!recs || !scratch) { free(recs); free(scratch); return -1; }
  uint8_t* p = recs;
  int64_t max_delta = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t r = idx ? idx[i] : i;
    const int64_t d = ts_delta ? ts_delta[i] : 0;
    if (d > max_delta) max_delta = d;
    p = put_record(p, d, (int32_t)i, keys + key_off[r], key_off[r + 1] - key_off[r], vals + val_off[r], val_off[r + 1] - val_off[r]);
  }
  const int64_t len = put_batch(out, base_offset, (int32_t)n, recs, p - recs, flags, producer_id, producer_epoch, base_sequence, first_ts, first_ts + max_delta, scratch);
  free(recs);
  free(scratch);
  return len;
}

/* A COMMIT (kind 1) / ABORT (kind 0) marker: a transactional control batch of one record, never compressed.  78 bytes. */
int64_t surge_test_wire_control(uint8_t* out, int64_t offset, int64_t producer_id, int32_t producer_epoch, int32_t kind, int64_t ts) {
  uint8_t key[4], val[6], recs[32];
  be16(be16(key, 0), kind);      /* version 0, type */
  be32(be16(val, 0), 0);         /* version 0, coordinatorEpoch 0 */
  uint8_t* p = put_record_h(recs, 0, 0, key, 4, val, 6, 0); /* (a marker carries no headers) */
  return put_batch(out, offset, 1, recs, p - recs, WIRE_TRANSACTIONAL | WIRE_CONTROL, producer_id, producer_epoch, -1, ts, ts, NULL);
}

/* ---- a whole topic, fetch response by fetch response ------------------------------------------------------------------------ */
typedef struct {
  uint8_t* buf;  /* the partition's bytes of the last fetch */
  int64_t cap, len;
  uint8_t *recs, *scratch; /* one batch's records section / its LZ4 frame */
  int64_t recs_cap;
  int64_t next_offset; /* the log end offset */
  int64_t flush_no;    /* flushes (transactions) written so far */
  int32_t base_seq;    /* the idempotent producer's next sequence number */
  int8_t marker_due;   /* -1 none, else the kind of the marker held back at the end of the last fetch */
  int64_t counts[8];
} wire_partition;

typedef struct {
  int32_t P;
  wire_partition* part;
  int64_t* order;
  int64_t order_cap;
  int64_t* start; /* P + 1 */
} wire_topic;

wire_topic* surge_test_wire_topic_create(int32_t P) {
  wire_topic* t = (wire_topic*)calloc(1, sizeof(wire_topic));
  if (!t) return NULL;
  t->P = P;
  t->part = (wire_partition*)calloc((size_t)P, sizeof(wire_partition));
  t->start = (int64_t*)calloc((size_t)P + 1, 8);
  for (int32_t p = 0; p < P; ++p) t->part[p].marker_due = -1;
  return t;
}
void surge_test_wire_topic_destroy(wire_topic* t) {
  if (!t) return;
  for (int32_t p = 0; p < t->P; ++p) { free(t->part[p].buf); free(t->part[p].recs); free(t->part[p].scratch); }
  free(t->part); free(t->order); free(t->start);
  free(t);
}
const uint8_t* surge_test_wire_topic_partition(const wire_topic* t, int32_t p, int64_t* len_out) {
  *len_out = t->part[p].len;
  return t->part[p].buf;
}
int64_t surge_test_wire_topic_end_offset(const wire_topic* t, int32_t p) { return t->part[p].next_offset; }

static int grow(wire_partition* w, int64_t more) {
  if (w->len + more <= w->cap) return 0;
  int64_t c = w->cap ? w->cap : (1 << 16);
  while (c < w->len + more) c += c / 2;
  uint8_t* nb = (uint8_t*)realloc(w->buf, (size_t)c);
  if (!nb) return -1;
  w->buf = nb;
  w->cap = c;
  return 0;
}

/* partition p's share of a fetch: records idx[0 .. np) */
static int32_t partition_fetch(wire_partition* w, int32_t p, const int64_t* idx, int64_t np, int64_t max_rec, const uint8_t* keys, const int64_t* key_off,
                               const uint8_t* vals, const int64_t* val_off, int64_t flush_events, int64_t max_batch_bytes, int32_t lz4, int64_t abort_every,
                               int32_t hold_markers) {
  const int64_t pid = 1000 + p;
  w->len = 0;
  const int64_t recs_cap = (max_batch_bytes > 0 ? max_batch_bytes : (1 << 20)) + max_rec + 64;
  if (w->recs_cap < recs_cap) {
    free(w->recs); free(w->scratch);
    w->recs = (uint8_t*)malloc((size_t)recs_cap);
    w->scratch = (uint8_t*)malloc((size_t)(recs_cap + recs_cap / 255 + 64 + 8 * (recs_cap / 65536 + 1)));
    w->recs_cap = (w->recs && w->scratch) ? recs_cap : 0;
    if (!w->recs_cap) return -1;
  }
  const int64_t batch_room = 61 + recs_cap + recs_cap / 255 + 64 + 8 * (recs_cap / 65536 + 1);
  if (w->marker_due >= 0) { /* the marker the last fetch held back */
    if (grow(w, 128)) return -1;
    w->len += surge_test_wire_control(w->buf + w->len, w->next_offset++, pid, 0, w->marker_due, 1700000000000ll + 50 * w->flush_no);
    w->marker_due = -1;
    w->counts[1] += 1;
  }
  const int64_t K = flush_events > 0 ? flush_events : (np > 0 ? np : 1);
  for (int64_t f0 = 0; f0 < np; f0 += K) {
    const int64_t fn = np - f0 < K ? np - f0 : K;
    const int txn = flush_events > 0;
    const int fails_first = txn && abort_every > 0 && (w->flush_no % abort_every) == abort_every - 1;
    for (int attempt = fails_first ? 0 : 1; attempt < 2; ++attempt) {
      const int64_t ts = 1700000000000ll + 50 * (w->flush_no + 1) + attempt;
      int64_t done = 0;
      while (done < fn) { /* one data batch */
        uint8_t* rp = w->recs;
        int64_t cnt = 0, first_delta = -1, max_delta = 0;
        while (done + cnt < fn) {
          const int64_t r = idx[f0 + done + cnt];
          const int64_t kl = key_off[r + 1] - key_off[r], vl = val_off[r + 1] - val_off[r];
          if (cnt > 0 && max_batch_bytes > 0 && (rp - w->recs) + kl + vl + 12 + g_headers_len > max_batch_bytes) break;
          const int64_t when = txn ? ((done + cnt) * 50) / fn : 0; /* ms into the flush interval */
          if (first_delta < 0) first_delta = when;
          if (when - first_delta > max_delta) max_delta = when - first_delta;
          rp = put_record(rp, when - first_delta, (int32_t)cnt, keys + key_off[r], kl, vals + val_off[r], vl);
          ++cnt;
        }
        if (grow(w, batch_room)) return -1;
        const int64_t first_ts = ts - 50 + first_delta;
        w->len += put_batch(w->buf + w->len, w->next_offset, (int32_t)cnt, w->recs, rp - w->recs, (lz4 ? WIRE_LZ4 : 0) | (txn ? WIRE_TRANSACTIONAL : 0), txn ? pid : -1, 0,
                            txn ? w->base_seq : -1, first_ts, first_ts + max_delta, w->scratch);
        w->next_offset += cnt;
        if (txn) w->base_seq += (int32_t)cnt;
        done += cnt;
        w->counts[0] += 1;
        w->counts[2] += cnt;
        if (attempt == 0) w->counts[3] += cnt;
      }
      if (txn) {
        const int kind = attempt == 0 ? 0 : 1; /* ABORT, then the retry's COMMIT */
        const int last = f0 + fn >= np && attempt == 1;
        if (last && hold_markers > 0 && p % hold_markers == 1) {
          w->marker_due = (int8_t)kind;
        } else {
          if (grow(w, 128)) return -1;
          w->len += surge_test_wire_control(w->buf + w->len, w->next_offset++, pid, 0, kind, ts);
          w->counts[1] += 1;
        }
        w->counts[4] += 1;
      }
    }
    if (txn) ++w->flush_no;
  }
  w->counts[5] += w->len;
  return 0;
}

/* The next fetch response: records 0 .. n of (partition[], keys, values), each partition's in array order.
 *   flush_events = 0: plain (non-transactional) batches closed at max_batch_bytes of records only — the product writer's layout;
 *   flush_events = K: partition p's records are published K per flush, each flush ONE transaction of producer id 1000 + p:
 *     its data batches (closed by the flush, or earlier at max_batch_bytes), then a COMMIT marker.  Every abort_every-th
 *     flush of a partition (abort_every > 0) first fails: the same records + an ABORT marker, then the publisher's retry
 *     (KafkaProducerActorImpl.scala:441: abortTransaction, the senders are told and publish again) commits them.
 *   hold_markers = H > 0: on partitions p % H == 1 the marker of the fetch's last flush is held back and opens the next
 *     fetch (a fetch response ends where the broker's byte budget ends — between a transaction's data and its marker as
 *     likely as anywhere); surge_test_wire_topic_fetch with n = 0 flushes what is held back.
 * Timestamps: flush f of a partition happens at 1.7e12 + 50 f ms, its records spread over the 50 ms before it.
 * counts[8] (SET to the totals so far): data batches, control batches, records written (aborted copies included), aborted
 * records, transactions, bytes, -, -.  Partitions are written side by side (OpenMP, when compiled with it).
 * Returns 0, or -1 (out of memory). */
int32_t surge_test_wire_topic_fetch(wire_topic* t, int64_t n, const int32_t* partition, const uint8_t* keys, const int64_t* key_off, const uint8_t* vals,
                                    const int64_t* val_off, int64_t flush_events, int64_t max_batch_bytes, int32_t lz4, int64_t abort_every, int32_t hold_markers,
                                    int64_t* counts) {
  const int32_t P = t->P;
  if (!crc_ready) crc_init(); /* (before the threads start) */
  /* stable counting sort of the record numbers by partition */
  int64_t* start = t->start;
  memset(start, 0, ((size_t)P + 1) * 8);
  if (t->order_cap < n) {
    free(t->order);
    t->order = (int64_t*)malloc((size_t)(n + 1) * 8);
    t->order_cap = t->order ? n : 0;
    if (!t->order) return -1;
  }
  int64_t max_rec = 64;
  for (int64_t i = 0; i < n; ++i) {
    ++start[partition[i] + 1];
    const int64_t rl = 40 + g_headers_len + (key_off[i + 1] - key_off[i]) + (val_off[i + 1] - val_off[i]);
    if (rl > max_rec) max_rec = rl;
  }
  for (int32_t p = 0; p < P; ++p) start[p + 1] += start[p];
  {
    int64_t* fill = (int64_t*)malloc((size_t)P * 8);
    if (!fill) return -1;
    memcpy(fill, start, (size_t)P * 8);
    for (int64_t i = 0; i < n; ++i) t->order[fill[partition[i]]++] = i;
    free(fill);
  }
  int32_t bad = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int32_t p = 0; p < P; ++p) {
    if (partition_fetch(&t->part[p], p, t->order + start[p], start[p + 1] - start[p], max_rec, keys, key_off, vals, val_off, flush_events, max_batch_bytes, lz4,
                        abort_every, hold_markers) != 0) {
#pragma omp atomic write
      bad = 1;
    }
  }
  for (int k = 0; k < 8; ++k) counts[k] = 0;
  for (int32_t p = 0; p < P; ++p)
    for (int k = 0; k < 8; ++k) counts[k] += t->part[p].counts[k];
  return bad ? -1 : 0;
}
