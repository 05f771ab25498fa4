// ingest_kernels.hip — SURVEY §8f N1 on the device: the records sections of Kafka record batches (message format v2,
// uncompressed or already decompressed by the host framer) -> (aggregate index, 16-byte event, offset) arrays and a key
// table, all resident in HBM, ready for surge_replay_append_events_device / a CSR build.
//
// Split of the work (include/surge_ingest.h, "device decode"): the HOST walks the 61-byte batch headers, verifies the
// CRC-32C (one instruction stream per partition thread), applies read_committed and undoes LZ4 — sequential, cheap per
// byte — and hands over the records sections as they are.  The DEVICE does everything that is per record: chains the
// varint-framed records of every batch, parses keys / values, interns the aggregate ids (key up to ':') in a hash table,
// decodes the event values (16-byte events as they are, or the reference's play-json text through the event template,
// surge_amd/csrc/event_decode.cpp's rules) and compacts away the producer's flush records.  The host decoder spends
// ≈ 90 ns (fixed-16) / 640 ns (JSON) per record and thread on exactly these steps.
//
// Kernels, per push:
//   chain      one thread per batch: record i's start = record i-1's start + its varint length (the only sequential step)
//   parse      one thread per record: varints -> key span, value span, offset; 64-bit hash of the key up to ':'
//   probe      one thread per record: open-addressing insert-or-find by hash (atomicCAS); a NEW slot remembers its first record
//   new keys   the new slots, ordered by first record (rocPRIM sort) -> dense ids in first-delivered order (the host
//              decoder's order), key bytes copied to the device key arena (offsets by rocPRIM scan)
//   resolve    one thread per record: aggregate index from its slot (key bytes compared with the arena: a 64-bit hash
//              collision is detected, not trusted), value -> event16
//   compact    exclusive scan of the keep flags, scatter to the result arrays
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/surge_ingest.h"
#include "../../include/surge_replay.h"
#include "f64_parse.h"

namespace {

constexpr int32_t OK = 0, E_INVALID = -1, E_DEVICE = -3, E_NOMEM = -4, E_UNSUPPORTED = -5;

// per-record status
enum : uint32_t {
  RS_OK = 0,
  RS_SKIP = 1,          // the producer's flush record (empty key, empty value): not an event
  RS_NULL = 2,          // null key or null value: not an event
  RS_MALFORMED = 3,     // varint / span runs past its record or batch
  RS_JSON = 4,          // value is not the JSON the template describes
  RS_TYPE = 5,          // unknown discriminator value
  RS_FIELD = 6,         // a field the template names is missing or is not the number it should be
  RS_SIZE = 7,          // fixed-16 topic: value is not 16 bytes
  RS_F64_HOST = 8,      // a Double the fast parser cannot decide: the host re-parses this value exactly
  RS_COLLISION = 9,     // two different keys with the same 64-bit hash
};

struct RecMeta {
  int64_t key_off, val_off, offset;
  uint64_t hash;
  int32_t key_len;   // aggregate id length (key up to ':')
  int32_t val_len;
  uint32_t slot;
  uint32_t status;
};

struct Section {  // = surge_batch_section + the batch's first record index in this push
  int64_t byte_off, byte_len, base_offset;
  int32_t n_records, reserved;
  int64_t rec_first;
};

struct ErrorCell {
  unsigned long long first_bad;  // min over (record index << 8 | status)
  unsigned int n_new, n_f64_host;
  unsigned int lz4_bad;          // the first section whose LZ4 frame did not decode (~0 = none)
  unsigned int pad;
};

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok;
  __device__ int64_t varlong() {
    uint64_t v = 0;
    int shift = 0;
    while (true) {
      if (p >= end || shift > 63) { ok = false; return 0; }
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) break;
      shift += 7;
    }
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
};

__device__ __forceinline__ void report(ErrorCell* err, int64_t rec, uint32_t status) {
  atomicMin(&err->first_bad, ((unsigned long long)rec << 8) | status);
}

// ---- LZ4 (the reference's producer publishes lz4: reference.conf:112) ------------------------------------------------
// A record batch's records section is ONE LZ4 frame; kafka-clients writes it with independent blocks of at most 64 KiB
// (KafkaLZ4BlockOutputStream's default), so every block but a frame's last decompresses to exactly 64 KiB and block k
// of a frame lands at k x 64 KiB of the frame's output.  The host walks the frame header and the block size words
// (nothing per byte); one WAVE decodes one block: the token / length / offset bytes are wave-uniform (every lane reads
// the same byte: one broadcast load), literal runs and matches are copied by the 64 lanes together, an overlapping
// match (offset < length: the period IS the data) as out[op + i] = out[op - offset + i % offset].  The block is
// assembled in LDS — lanes read what other lanes wrote a sequence ago, which global memory does not promise inside a
// wave — and leaves as whole 16-byte stores.
constexpr int kLz4BlockMax = 65536;
struct Lz4Block {
  int64_t src_off;   // in the staged bytes
  int64_t dst_off;   // in the decompressed area
  int32_t src_len;   // bit 31: stored uncompressed
  int32_t section;   // the section this block belongs to
  int32_t last;      // the frame's last block: sets the section's length
  int32_t index;     // k: this block's number inside its frame
};

// The compressed stream is read through a 256-byte WINDOW held in registers (lane l: bytes [4l, 4l + 4) from `wbase`):
// control bytes come out of it with v_readlane, short literal runs with one shuffle — no global load on the path from
// one sequence to the next, only a refill every ~250 consumed bytes.  (First version: token, length bytes and offset
// were three dependent global loads per sequence.)
struct Lz4Window {
  const uint8_t* in;
  int32_t n_in, wbase;
  uint32_t win;
  int lane;
  __device__ void refill(int32_t at) {
    wbase = at;
    const int32_t pos = at + 4 * lane;
    uint32_t v = 0;
    if (pos + 4 <= n_in) {
      v = (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16) | ((uint32_t)in[pos + 3] << 24);
    } else {
      for (int k = 0; k < 4; ++k)
        if (pos + k < n_in) v |= (uint32_t)in[pos + k] << (8 * k);
    }
    win = v;
  }
  __device__ uint32_t byte_at(int32_t at) {  // `at` is wave-uniform
    if (at - wbase >= 256 || at < wbase) refill(at);
    const int32_t idx = at - wbase;
    const uint32_t word = __builtin_amdgcn_readlane(win, __builtin_amdgcn_readfirstlane(idx >> 2));
    return (word >> ((idx & 3) * 8)) & 0xffu;
  }
  // one sequence's header at ip: literal length, then (after the literals) offset and match length; false = malformed
  __device__ bool literal_length(int32_t& ip, uint32_t& token, int32_t& lit) {
    token = byte_at(ip++);
    lit = (int32_t)(token >> 4);
    if (lit == 15) {
      uint32_t x;
      do {
        if (ip >= n_in) return false;
        x = byte_at(ip++);
        lit += (int32_t)x;
      } while (x == 255u && lit < (1 << 24));
    }
    return lit <= n_in - ip;
  }
  __device__ bool match(int32_t& ip, uint32_t token, int32_t& offset, int32_t& ml) {
    if (n_in - ip < 2) return false;
    offset = (int32_t)(byte_at(ip) | (byte_at(ip + 1) << 8));
    ip += 2;
    ml = (int32_t)(token & 15u);
    if (ml == 15) {
      uint32_t x;
      do {
        if (ip >= n_in) return false;
        x = byte_at(ip++);
        ml += (int32_t)x;
      } while (x == 255u && ml < (1 << 24));
    }
    ml += 4;
    return offset != 0;
  }
};

// One launch per LDS size class (16 / 32 / 64 KiB per wave), smallest first.  Nothing tells what a block decompresses to
// but decoding it — the reference's producer closes a batch at 16 KiB (kafka.publisher.batch-size = 16384,
// reference.conf:115), so its blocks need a quarter of the 64 KiB a block may take, and four times as many waves fit a CU
// — so a block is TRIED in the smallest class its compressed size does not already rule out, every write bounded by the
// class's LDS; a block that outgrows it is left for the next class (state[b] = -(class + 2)) and decoded again there from
// its start.  (Rounds before: a separate first pass walked every block's sequence headers for its size — 1.3 ms of the
// 5.8 ms a 1 M-record fetch spends on the device; the retry costs a producer of large batches up to a quarter of a
// block, the 16 KiB producer nothing.)   state[b]: >= 0 decoded (its size), -1 malformed, -2 / -3 waiting for class 1 / 2.
__global__ void __launch_bounds__(64) lz4_block_kernel(const uint8_t* __restrict__ bytes, uint8_t* __restrict__ out_base, const Lz4Block* __restrict__ blocks,
                                                       int64_t n_blocks, int32_t* __restrict__ state, int32_t cls, int32_t cap,
                                                       Section* __restrict__ sections, ErrorCell* err) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lz4_out[];
  const int lane = threadIdx.x;
  for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
    if (cls > 0 && state[b] != -(cls + 1)) continue;
    const Lz4Block blk = blocks[b];
    const uint8_t* in = bytes + blk.src_off;
    const int32_t n_in = blk.src_len & 0x7fffffff;
    int32_t op = 0;
    bool ok = true, grow = false;
    if (!blk.last && cap < kLz4BlockMax) {  // every block of a frame but its last holds exactly 64 KiB
      grow = true;
    } else if (blk.src_len < 0) {  // stored
      if (n_in > cap) {
        grow = true;
      } else {
        for (int i = lane; i < n_in; i += 64) lz4_out[i] = in[i];
        op = n_in;
      }
    } else if (n_in > cap + (cap >> 7) + 64) {  // a compressed block is never much longer than what it holds
      grow = true;
    } else {
      Lz4Window w{in, n_in, 0, 0u, lane};
      w.refill(0);
      int32_t ip = 0;
      while (ip < n_in) {
        uint32_t token;
        int32_t lit, offset, ml;
        if (!w.literal_length(ip, token, lit)) { ok = false; break; }
        if (lit > cap - op) { grow = true; break; }
        if (lit > 0) {
          if (lit <= 64 && ip >= w.wbase && ip + lit - w.wbase <= 256) {  // the whole run is in the window
            const int32_t idx = ip - w.wbase + lane;
            const uint32_t word = (uint32_t)__shfl((int)w.win, (idx >> 2) & 63, 64);
            if (lane < lit) lz4_out[op + lane] = (uint8_t)(word >> ((idx & 3) * 8));
          } else {
            for (int i = lane; i < lit; i += 64) lz4_out[op + i] = in[ip + i];
          }
        }
        ip += lit;
        op += lit;
        if (ip >= n_in) break;  // the last sequence carries literals only
        if (!w.match(ip, token, offset, ml) || offset > op) { ok = false; break; }
        if (ml > cap - op) { grow = true; break; }
        // (LDS operations of one wave execute in order: a read sees every earlier write of any lane of this wave)
        const uint8_t* src = lz4_out + op - offset;
        if (offset >= ml) {
          for (int i = lane; i < ml; i += 64) lz4_out[op + i] = src[i];
        } else {  // the period IS the data
          for (int i = lane; i < ml; i += 64) lz4_out[op + i] = src[i % offset];
        }
        op += ml;
      }
    }
    if (grow && cap >= kLz4BlockMax) { grow = false; ok = false; }   // larger than a block may be
    if (ok && !grow && !blk.last && op != kLz4BlockMax) ok = false;  // every block of a frame but its last is exactly full
    if (grow) {
      if (lane == 0) state[b] = -(cls + 2);
    } else if (!ok) {
      if (lane == 0) {
        state[b] = -1;
        atomicMin(&err->lz4_bad, (unsigned int)blk.section);
        if (blk.last) sections[blk.section].byte_len = 0;
      }
    } else {
      uint8_t* dst = out_base + blk.dst_off;  // 16-byte aligned: dst_off is a multiple of 64 KiB from an aligned base
      const int n16 = op >> 4;
      for (int i = lane; i < n16; i += 64) ((uint4*)dst)[i] = ((const uint4*)lz4_out)[i];
      for (int i = (n16 << 4) + lane; i < op; i += 64) dst[i] = lz4_out[i];
      if (lane == 0) {
        state[b] = op;
        if (blk.last) sections[blk.section].byte_len = (int64_t)blk.index * kLz4BlockMax + op;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the copy-out has read the LDS before the next block overwrites it
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void chain_kernel(const uint8_t* __restrict__ bytes, const Section* __restrict__ sections, int64_t n_sections,
                             int64_t* __restrict__ rec_pos, int64_t* __restrict__ rec_end, ErrorCell* err) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_sections) return;
  const Section sec = sections[s];
  Reader r{bytes + sec.byte_off, bytes + sec.byte_off + sec.byte_len, true};
  for (int32_t i = 0; i < sec.n_records; ++i) {
    const int64_t len = r.varlong();
    const int64_t at = r.p - bytes;
    if (!r.ok || len < 0 || r.end - r.p < len) {
      // everything from here to the end of the batch is unreadable: mark the rest empty and report the first
      for (int32_t k = i; k < sec.n_records; ++k) { rec_pos[sec.rec_first + k] = -1; rec_end[sec.rec_first + k] = -1; }
      report(err, sec.rec_first + i, RS_MALFORMED);
      return;
    }
    rec_pos[sec.rec_first + i] = at;
    rec_end[sec.rec_first + i] = at + len;
    r.p += len;
  }
}

__device__ __forceinline__ uint64_t hash_key(const uint8_t* p, int n) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
  for (int i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
  h ^= h >> 29;
  h *= 0xD6E8FEB86659FD93ull;
  h ^= h >> 32;
  return h == 0ull ? 1ull : h;  // 0 marks an empty slot
}

// section of record i: the last section with rec_first <= i (records of a section are contiguous)
__device__ __forceinline__ int64_t section_of(const Section* sections, int64_t n_sections, int64_t i) {
  int64_t lo = 0, hi = n_sections;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (sections[mid].rec_first <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void parse_kernel(const uint8_t* __restrict__ bytes, const Section* __restrict__ sections, int64_t n_sections,
                             const int64_t* __restrict__ rec_pos, const int64_t* __restrict__ rec_end, int64_t n_rec, RecMeta* __restrict__ meta,
                             ErrorCell* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  RecMeta m;
  m.key_off = m.val_off = m.offset = 0; m.hash = 0; m.key_len = m.val_len = 0; m.slot = 0; m.status = RS_MALFORMED;
  const int64_t at = rec_pos[i];
  if (at >= 0) {
    Reader q{bytes + at, bytes + rec_end[i], true};
    if (q.p < q.end) ++q.p; else q.ok = false;  // attributes
    (void)q.varlong();                            // timestampDelta
    const int64_t offset_delta = q.varlong();
    const int64_t klen = q.varlong();
    const uint8_t* key = q.p;
    if (q.ok && klen > 0) { if (q.end - q.p >= klen) q.p += klen; else q.ok = false; }
    const int64_t vlen = q.ok ? q.varlong() : 0;
    const uint8_t* val = q.p;
    if (q.ok && vlen > 0) { if (q.end - q.p >= vlen) q.p += vlen; else q.ok = false; }
    // (headers follow; the chain already knows where the record ends)
    if (q.ok && klen >= -1 && vlen >= -1 && klen < (1ll << 31) && vlen < (1ll << 31)) {
      const Section& sec = sections[section_of(sections, n_sections, i)];
      m.offset = sec.base_offset + offset_delta;
      if (klen == 0 && vlen == 0) {
        m.status = RS_SKIP;  // KafkaProducerActorImpl.scala:322-329
      } else if (klen < 0 || vlen < 0) {
        m.status = RS_NULL;
      } else {
        int n = 0;
        while (n < (int)klen && key[n] != (uint8_t)':') ++n;  // PartitionStringUpToColon (KafkaPartitioner.scala:38-42)
        m.key_off = key - bytes;
        m.key_len = n;
        m.val_off = val - bytes;
        m.val_len = (int32_t)vlen;
        m.hash = hash_key(key, n);
        m.status = RS_OK;
      }
    }
  }
  if (m.status >= RS_NULL) report(err, i, m.status);
  meta[i] = m;
}

// records that arrive already framed (a JVM's ConsumerRecords: key bytes, value bytes, offset per record): the bytes buffer
// holds the keys first, the values from `val_base` on
__global__ void records_meta_kernel(const uint8_t* __restrict__ bytes, const int64_t* __restrict__ key_off, const int64_t* __restrict__ val_off,
                                    const int64_t* __restrict__ offsets, int64_t val_base, int64_t n_rec, RecMeta* __restrict__ meta, ErrorCell* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  RecMeta m;
  const int64_t k0 = key_off[i], k1 = key_off[i + 1], v0 = val_off[i], v1 = val_off[i + 1];
  m.key_off = k0; m.val_off = val_base + v0; m.offset = offsets ? offsets[i] : i; m.hash = 0; m.key_len = 0; m.val_len = 0; m.slot = 0;
  if (k1 < k0 || v1 < v0 || k1 - k0 >= (1ll << 31) || v1 - v0 >= (1ll << 31)) {
    m.status = RS_MALFORMED;
  } else if (k1 == k0 && v1 == v0) {
    m.status = RS_SKIP;  // the producer's flush record
  } else {
    const uint8_t* key = bytes + k0;
    int n = 0;
    while (n < (int)(k1 - k0) && key[n] != (uint8_t)':') ++n;
    m.key_len = n;
    m.val_len = (int32_t)(v1 - v0);
    m.hash = hash_key(key, n);
    m.status = RS_OK;
  }
  if (m.status >= RS_NULL) report(err, i, m.status);
  meta[i] = m;
}

struct Table {
  unsigned long long* hash;  // 0 = empty
  uint32_t* key_id;          // 0xffffffff = not assigned yet (inserted by the push in flight)
  uint32_t* first_rec;       // of a slot inserted by the push in flight: its first record
  uint64_t mask;
};

__global__ void probe_kernel(RecMeta* __restrict__ meta, int64_t n_rec, Table t, uint32_t* __restrict__ new_slots, ErrorCell* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  if (meta[i].status != RS_OK) return;
  const unsigned long long h = meta[i].hash;
  uint64_t s = h & t.mask;
  while (true) {
    const unsigned long long old = atomicCAS(&t.hash[s], 0ull, h);
    if (old == 0ull) {  // inserted: this push discovers the key
      new_slots[atomicAdd(&err->n_new, 1u)] = (uint32_t)s;
      break;
    }
    if (old == h) break;
    s = (s + 1) & t.mask;
  }
  meta[i].slot = (uint32_t)s;
  if (t.key_id[s] == 0xffffffffu) atomicMin(&t.first_rec[s], (uint32_t)i);
}

__global__ void newkey_keys_kernel(const uint32_t* __restrict__ new_slots, uint32_t n_new, Table t, uint32_t* __restrict__ sort_keys) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_new) sort_keys[k] = t.first_rec[new_slots[k]];
}

// after the sort: slot k (in first-record order) gets id n_keys + k; lens[k] = its key length (scanned into arena offsets)
__global__ void newkey_len_kernel(const uint32_t* __restrict__ first_sorted, uint32_t n_new, const RecMeta* __restrict__ meta,
                                  int64_t* __restrict__ lens) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_new) lens[k] = meta[first_sorted[k]].key_len;
  if (k == n_new) lens[k] = 0;
}

__global__ void newkey_assign_kernel(const uint32_t* __restrict__ first_sorted, const uint32_t* __restrict__ slot_sorted, uint32_t n_new,
                                     const RecMeta* __restrict__ meta, const uint8_t* __restrict__ bytes, const int64_t* __restrict__ lens_scanned,
                                     int64_t n_keys, int64_t arena_base, Table t, uint8_t* __restrict__ arena, int64_t* __restrict__ key_off,
                                     unsigned long long* __restrict__ key_hash) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_new) return;
  const RecMeta m = meta[first_sorted[k]];
  const int64_t dst = arena_base + lens_scanned[k];
  for (int b = 0; b < m.key_len; ++b) arena[dst + b] = bytes[m.key_off + b];
  const int64_t id = n_keys + k;
  key_off[id + 1] = dst + m.key_len;
  key_hash[id] = m.hash;
  t.key_id[slot_sorted[k]] = (uint32_t)id;
}

__global__ void rehash_kernel(const unsigned long long* __restrict__ key_hash, int64_t n_keys, Table t) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_keys) return;
  const unsigned long long h = key_hash[id];
  uint64_t s = h & t.mask;
  while (atomicCAS(&t.hash[s], 0ull, h) != 0ull) s = (s + 1) & t.mask;  // distinct keys, distinct (verified) hashes
  t.key_id[s] = (uint32_t)id;
}

// ---- event values ------------------------------------------------------------------------------------------------------
struct JsonScan {
  const uint8_t* p;
  const uint8_t* end;
  __device__ void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  __device__ bool string(const uint8_t** s, int* len, bool* escaped) {
    if (p >= end || *p != '"') return false;
    ++p;
    *s = p;
    *escaped = false;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        *escaped = true;
        ++p;
        if (p >= end) return false;
      }
      ++p;
    }
    if (p >= end) return false;
    *len = (int)(p - *s);
    ++p;
    return true;
  }
  __device__ bool number(const uint8_t** s, int* len) {
    *s = p;
    if (p < end && (*p == '-' || *p == '+')) ++p;
    bool digits = false;
    while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) {
      digits = digits || (*p >= '0' && *p <= '9');
      ++p;
    }
    *len = (int)(p - *s);
    return digits;
  }
  __device__ bool skip_value() {
    ws();
    if (p >= end) return false;
    const uint8_t* s; int l; bool e;
    if (*p == '"') return string(&s, &l, &e);
    if (*p == '{' || *p == '[') {
      int depth = 0;
      while (p < end) {
        if (*p == '"') {
          if (!string(&s, &l, &e)) return false;
          continue;
        }
        if (*p == '{' || *p == '[') ++depth;
        if (*p == '}' || *p == ']') {
          --depth;
          if (depth == 0) { ++p; return true; }
        }
        ++p;
      }
      return false;
    }
    if (*p == 't' || *p == 'f' || *p == 'n') {
      while (p < end && *p >= 'a' && *p <= 'z') ++p;
      return true;
    }
    return number(&s, &l);
  }
};

__device__ __forceinline__ bool name_is(const char* want, const uint8_t* got, int got_len) {
  int n = 0;
  while (n < SURGE_EVJ_NAME && want[n]) ++n;
  if (n != got_len) return false;
  for (int i = 0; i < n; ++i)
    if ((uint8_t)want[i] != got[i]) return false;
  return true;
}

struct Found {
  const uint8_t* s;
  int len;
  bool is_num, is_str, escaped;
};

// One pass over the top-level object, looking for the fields named a / b (either may be NULL) exactly as the host
// decoder's lookups do on duplicated names: FIRST_NUMERIC = the first field of that name that is a number (find_num),
// otherwise the last field of that name whatever it is (the discriminator loop).  Returns false on malformed JSON.
template <bool FIRST_NUMERIC>
__device__ bool scan_object(const uint8_t* v, int len, const char* name_a, Found* fa, const char* name_b, Found* fb) {
  JsonScan sc{v, v + len};
  if (fa) fa->s = nullptr;
  if (fb) fb->s = nullptr;
  sc.ws();
  if (sc.p >= sc.end || *sc.p != '{') return false;
  ++sc.p;
  sc.ws();
  if (sc.p < sc.end && *sc.p == '}') {
    ++sc.p;
  } else {
    int n_fields = 0;
    for (;;) {
      sc.ws();
      const uint8_t* key; int key_len; bool esc = false;
      if (!sc.string(&key, &key_len, &esc)) return false;
      sc.ws();
      if (sc.p >= sc.end || *sc.p != ':') return false;
      ++sc.p;
      sc.ws();
      if (sc.p >= sc.end) return false;
      Found cur;
      cur.s = nullptr; cur.len = 0; cur.is_num = cur.is_str = cur.escaped = false;
      if (*sc.p == '"') {
        if (!sc.string(&cur.s, &cur.len, &cur.escaped)) return false;
        cur.is_str = true;
      } else if (*sc.p == '-' || (*sc.p >= '0' && *sc.p <= '9')) {
        if (!sc.number(&cur.s, &cur.len)) return false;
        cur.is_num = true;
      } else if (!sc.skip_value()) {
        return false;
      }
      if (!esc && n_fields < 24) {  // the host decoder remembers 24 fields and ignores names with escapes
        ++n_fields;
        if (fa && name_a && name_is(name_a, key, key_len) && (FIRST_NUMERIC ? (cur.is_num && !(fa->s && fa->is_num)) : true)) *fa = cur;
        if (fb && name_b && name_is(name_b, key, key_len) && (FIRST_NUMERIC ? (cur.is_num && !(fb->s && fb->is_num)) : true)) *fb = cur;
      }
      sc.ws();
      if (sc.p < sc.end && *sc.p == ',') { ++sc.p; continue; }
      if (sc.p < sc.end && *sc.p == '}') { ++sc.p; break; }
      return false;
    }
  }
  sc.ws();
  return sc.p == sc.end;
}

__device__ int parse_i32(const uint8_t* s, int len, int32_t* out) {  // 0 ok, else not an Int
  if (len <= 0 || len > 11) return 1;
  int i = 0;
  bool neg = false;
  if (s[0] == '-') { neg = true; i = 1; }
  if (i >= len) return 1;
  int64_t v = 0;
  for (; i < len; ++i) {
    if (s[i] < '0' || s[i] > '9') return 1;
    v = v * 10 + (s[i] - '0');
  }
  if (neg) v = -v;
  if (v < -2147483648ll || v > 2147483647ll) return 1;
  *out = (int32_t)v;
  return 0;
}

// the rules of surge_event_json_decode (event_decode.cpp), on the device
__device__ uint32_t decode_json_event(const surge_event_json_template* t, const surge::F64ParseTable* ptab, const uint8_t* v, int len, uint4* out) {
  const surge_event_json_type* ty = nullptr;
  if (t->discriminator[0] == 0) {
    ty = &t->types[0];
    if (!scan_object<false>(v, len, nullptr, nullptr, nullptr, nullptr)) return RS_JSON;
  } else {
    Found d;
    if (!scan_object<false>(v, len, t->discriminator, &d, nullptr, nullptr)) return RS_JSON;
    if (!d.s || !d.is_str) return RS_FIELD;
    for (uint32_t i = 0; i < t->n_types && !ty; ++i)
      if (!d.escaped && name_is(t->types[i].name, d.s, d.len)) ty = &t->types[i];
    if (!ty) return RS_TYPE;
  }
  Found fs, fa;
  fs.s = fa.s = nullptr;
  if (ty->seq_field[0] || ty->arg_kind != SURGE_EVJ_ARG_NONE)
    (void)scan_object<true>(v, len, ty->seq_field[0] ? ty->seq_field : nullptr, &fs, ty->arg_kind != SURGE_EVJ_ARG_NONE ? ty->arg_field : nullptr, &fa);
  int32_t seq = 0;
  if (ty->seq_field[0]) {
    if (!fs.s || !fs.is_num || parse_i32(fs.s, fs.len, &seq) != 0) return RS_FIELD;
  }
  uint64_t raw = 0;
  uint32_t status = RS_OK;
  if (ty->arg_kind != SURGE_EVJ_ARG_NONE) {
    if (!fa.s || !fa.is_num) return RS_FIELD;
    if (ty->arg_kind == SURGE_EVJ_ARG_I32) {
      int32_t a = 0;
      if (parse_i32(fa.s, fa.len, &a) != 0) return RS_FIELD;
      raw = (uint64_t)(uint32_t)a;
    } else {
      const int rc = surge::f64_parse_json_number(fa.s, fa.len, ptab, &raw);
      if (rc == surge::F64_PARSE_MALFORMED) return RS_FIELD;
      if (rc == surge::F64_PARSE_AMBIGUOUS) status = RS_F64_HOST;  // type and seq are final; the host fills in the payload
    }
  }
  *out = make_uint4(ty->event_type, (uint32_t)seq, (uint32_t)raw, (uint32_t)(raw >> 32));
  return status;
}

// One thread per record.  The records of a block lie next to each other in the staged bytes (a section's records are
// contiguous), so the block first copies the span that holds its keys and values into LDS with 16-byte loads and every
// thread then walks its JSON text there: the walk is a chain of dependent single-byte reads, and an LDS read answers in
// a tenth of the time a global one does (1.3 ms -> see DESIGN N1 per 1 M records).  A span that does not fit (records
// handed over as separate key / value arrays, very long values) is read in place.
constexpr int kResolveStage = 40960;
__global__ void __launch_bounds__(256) resolve_kernel(RecMeta* __restrict__ meta, int64_t n_rec, const uint8_t* __restrict__ bytes, Table t,
                                                      const uint8_t* __restrict__ arena, const int64_t* __restrict__ key_off,
                                                      const surge_event_json_template* tmpl, const surge::F64ParseTable* ptab,
                                                      int64_t* __restrict__ agg_tmp, uint4* __restrict__ ev_tmp, uint32_t* __restrict__ keep,
                                                      uint32_t* __restrict__ f64_host_list, ErrorCell* err) {
  extern __shared__ __attribute__((aligned(16))) uint8_t stage[];
  __shared__ unsigned long long s_lo, s_hi;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  RecMeta m;
  m.status = RS_MALFORMED;
  if (i < n_rec) m = meta[i];
  const bool live = i < n_rec && m.status == RS_OK;
  if (threadIdx.x == 0) { s_lo = ~0ull; s_hi = 0ull; }
  __syncthreads();
  if (live) {
    const int64_t v1 = m.val_off + (m.val_len > 0 ? m.val_len : 0), k1 = m.key_off + (m.key_len > 0 ? m.key_len : 0);
    atomicMin(&s_lo, (unsigned long long)(m.key_off < m.val_off ? m.key_off : m.val_off));
    atomicMax(&s_hi, (unsigned long long)(v1 > k1 ? v1 : k1));
  }
  __syncthreads();
  const unsigned long long lo = s_lo & ~15ull, hi = s_hi;
  const bool staged = hi > lo && hi - lo <= (unsigned long long)kResolveStage;  // block-uniform
  if (staged) {
    const int n16 = (int)((hi - lo + 15) >> 4);  // the staged bytes buffer ends 16 bytes after its last byte
    for (int c = threadIdx.x; c < n16; c += blockDim.x) ((uint4*)stage)[c] = *(const uint4*)(bytes + lo + 16ull * (unsigned)c);
    __syncthreads();
  }
  if (i >= n_rec) return;
  uint32_t k = 0;
  if (live) {
    const uint8_t* kp = staged ? (const uint8_t*)stage + (m.key_off - (int64_t)lo) : bytes + m.key_off;
    const uint8_t* vp = staged ? (const uint8_t*)stage + (m.val_off - (int64_t)lo) : bytes + m.val_off;
    const uint32_t id = t.key_id[m.slot];
    const int64_t a0 = key_off[id], a1 = key_off[id + 1];
    bool same = a1 - a0 == m.key_len;
    for (int b = 0; same && b < m.key_len; ++b) same = arena[a0 + b] == kp[b];
    uint4 e = make_uint4(0, 0, 0, 0);
    uint32_t st = RS_OK;
    if (!same) {
      st = RS_COLLISION;
    } else if (tmpl) {
      st = decode_json_event(tmpl, ptab, vp, m.val_len, &e);
    } else if (m.val_len == 16) {
      uint32_t w[4];
      for (int q = 0; q < 4; ++q) w[q] = (uint32_t)vp[4 * q] | ((uint32_t)vp[4 * q + 1] << 8) | ((uint32_t)vp[4 * q + 2] << 16) | ((uint32_t)vp[4 * q + 3] << 24);
      e = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      st = RS_SIZE;
    }
    if (st == RS_F64_HOST) {
      f64_host_list[atomicAdd(&err->n_f64_host, 1u)] = (uint32_t)i;
      st = RS_OK;
    }
    if (st != RS_OK) {
      report(err, i, st);
      meta[i].status = st;
    } else {
      agg_tmp[i] = (int64_t)id;
      ev_tmp[i] = e;
      k = 1;
    }
  }
  keep[i] = k;
}

__global__ void scatter_kernel(const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos, int64_t n_rec, const RecMeta* __restrict__ meta,
                               const int64_t* __restrict__ agg_tmp, const uint4* __restrict__ ev_tmp, int64_t out_base, int64_t* __restrict__ agg_out,
                               uint4* __restrict__ ev_out, int64_t* __restrict__ off_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec || !keep[i]) return;
  const int64_t o = out_base + pos[i];
  agg_out[o] = agg_tmp[i];
  ev_out[o] = ev_tmp[i];
  off_out[o] = meta[i].offset;
}

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes, bool keep, hipStream_t stream) {
    if (bytes <= cap) return hipSuccess;
    size_t want = cap * 2 > bytes ? cap * 2 : bytes;
    void* fresh = nullptr;
    hipError_t e = hipMalloc(&fresh, want);
    if (e != hipSuccess) return e;
    if (keep && p && cap) {
      e = hipMemcpyAsync(fresh, p, cap, hipMemcpyDeviceToDevice, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (e != hipSuccess) { (void)hipFree(fresh); return e; }
    }
    if (p) (void)hipFree(p);
    p = fresh;
    cap = want;
    return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

thread_local std::string g_dec_err;

}  // namespace

struct surge_device_decoder {
  int device = 0;
  hipStream_t stream = nullptr;
  bool json = false;
  std::string err;
  Buf d_tmpl, d_ptab, d_err;
  // per push
  Buf lz4_blocks, lz4_sizes;
  Buf d_bytes, d_sections, rec_pos, rec_end, meta, new_slots, sort_k_a, sort_k_b, sort_v_b, lens, agg_tmp, ev_tmp, keep, keep_pos, f64_list, temp;
  void* pinned = nullptr;
  size_t pinned_cap = 0;
  // hash table + key table
  Buf t_hash, t_key_id, t_first, arena, key_off, key_hash;
  uint64_t t_cap = 0;
  int64_t n_keys = 0, arena_bytes = 0;
  // result
  Buf r_agg, r_ev, r_off;
  int64_t n_records = 0;
  int64_t counters[4] = {0, 0, 0, 0};  // records seen, delivered, flush records skipped, f64 values re-parsed on the host
};

namespace {

int32_t dfail(surge_device_decoder* d, int32_t code, const std::string& m) {
  if (d) d->err = m;
  g_dec_err = m;
  return code;
}

#define DCHK(d, call)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return dfail(d, e_ == hipErrorOutOfMemory ? E_NOMEM : E_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

Table table_of(surge_device_decoder* d) {
  Table t;
  t.hash = (unsigned long long*)d->t_hash.p;
  t.key_id = (uint32_t*)d->t_key_id.p;
  t.first_rec = (uint32_t*)d->t_first.p;
  t.mask = d->t_cap - 1;
  return t;
}

// capacity for n_keys + extra more keys at a load factor of at most 1/2 (rebuilt from the key hashes when it grows)
int32_t ensure_table(surge_device_decoder* d, int64_t extra) {
  uint64_t need = 1024;
  while (need < (uint64_t)(d->n_keys + extra) * 2) need *= 2;
  if (need <= d->t_cap) return OK;
  if (need > (1ull << 32)) return dfail(d, E_UNSUPPORTED, "more than 2^31 aggregate ids");
  Buf h, k, f;
  DCHK(d, h.reserve(need * 8, false, d->stream));
  DCHK(d, k.reserve(need * 4, false, d->stream));
  DCHK(d, f.reserve(need * 4, false, d->stream));
  DCHK(d, hipMemsetAsync(h.p, 0, need * 8, d->stream));
  DCHK(d, hipMemsetAsync(k.p, 0xff, need * 4, d->stream));
  DCHK(d, hipMemsetAsync(f.p, 0xff, need * 4, d->stream));
  d->t_hash.release(); d->t_key_id.release(); d->t_first.release();
  d->t_hash = h; d->t_key_id = k; d->t_first = f;
  d->t_cap = need;
  if (d->n_keys > 0)
    hipLaunchKernelGGL(rehash_kernel, dim3((unsigned)((d->n_keys + 255) / 256)), dim3(256), 0, d->stream, (const unsigned long long*)d->key_hash.p,
                       d->n_keys, table_of(d));
  DCHK(d, hipGetLastError());
  return OK;
}

}  // namespace

extern "C" {

static void* pinned_alloc(size_t n) {
  void* p = nullptr;
  return hipHostMalloc(&p, n, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
static void pinned_release(void* p) { (void)hipHostFree(p); }

int32_t surge_ingest_use_pinned_arena(surge_ingest* g) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return dfail(nullptr, E_DEVICE, "no usable HIP device: the arena stays in pageable memory");
  return surge_ingest_set_allocator(g, pinned_alloc, pinned_release);
}

const char* surge_device_decoder_last_error(const surge_device_decoder* d) { return d ? d->err.c_str() : g_dec_err.c_str(); }

int32_t surge_device_decoder_create(int32_t device_id, void* hip_stream, const surge_event_json_template* tmpl, surge_device_decoder** out) {
  if (!out) return dfail(nullptr, E_INVALID, "out is NULL");
  *out = nullptr;
  if (tmpl && surge_event_json_validate(tmpl) != 0) return dfail(nullptr, E_INVALID, std::string("event template: ") + surge_event_json_last_error());
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return dfail(nullptr, E_DEVICE, "no usable HIP device (the device decoder has no CPU fallback: use surge_ingest_drain_*)");
  if (device_id < 0 || device_id >= n_dev) return dfail(nullptr, E_INVALID, "device_id out of range");
  surge_device_decoder* d = new (std::nothrow) surge_device_decoder();
  if (!d) return dfail(nullptr, E_NOMEM, "out of host memory");
  d->device = device_id;
  d->stream = (hipStream_t)hip_stream;
  d->json = tmpl != nullptr;
  int prev = 0;
  (void)hipGetDevice(&prev);
  int32_t rc = OK;
  auto init = [&]() -> int32_t {
    DCHK(d, hipSetDevice(device_id));
    DCHK(d, d->d_err.reserve(sizeof(ErrorCell), false, d->stream));
    DCHK(d, d->key_off.reserve(8, false, d->stream));
    DCHK(d, hipMemset(d->key_off.p, 0, 8));
    if (tmpl) {
      DCHK(d, d->d_tmpl.reserve(sizeof(*tmpl), false, d->stream));
      DCHK(d, hipMemcpy(d->d_tmpl.p, tmpl, sizeof(*tmpl), hipMemcpyHostToDevice));
      DCHK(d, d->d_ptab.reserve(sizeof(surge::F64ParseTable), false, d->stream));
      DCHK(d, hipMemcpy(d->d_ptab.p, surge::f64_parse_table_host(), sizeof(surge::F64ParseTable), hipMemcpyHostToDevice));
    }
    return OK;
  };
  rc = init();
  (void)hipSetDevice(prev);
  if (rc != OK) {
    surge_device_decoder_destroy(d);
    return rc;
  }
  *out = d;
  return OK;
}

int32_t surge_device_decoder_destroy(surge_device_decoder* d) {
  if (!d) return OK;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(d->device);
  (void)hipStreamSynchronize(d->stream);
  Buf* bufs[] = {&d->lz4_blocks, &d->lz4_sizes, &d->d_tmpl, &d->d_ptab, &d->d_err, &d->d_bytes, &d->d_sections, &d->rec_pos, &d->rec_end, &d->meta, &d->new_slots, &d->sort_k_a,
                 &d->sort_k_b, &d->sort_v_b, &d->lens, &d->agg_tmp, &d->ev_tmp, &d->keep, &d->keep_pos, &d->f64_list, &d->temp, &d->t_hash,
                 &d->t_key_id, &d->t_first, &d->arena, &d->key_off, &d->key_hash, &d->r_agg, &d->r_ev, &d->r_off};
  for (Buf* b : bufs) b->release();
  if (d->pinned) (void)hipHostFree(d->pinned);
  (void)hipSetDevice(prev);
  delete d;
  return OK;
}

}  // extern "C" (reopened below)

namespace {

// scratch of one push and a fresh error cell; the hash table sized for n_rec more keys
int32_t begin_push(surge_device_decoder* d, int64_t n_rec) {
  hipStream_t st = d->stream;
  const size_t R = (size_t)n_rec;
  DCHK(d, d->meta.reserve(R * sizeof(RecMeta), false, st));
  DCHK(d, d->new_slots.reserve(R * 4, false, st));
  DCHK(d, d->agg_tmp.reserve(R * 8, false, st));
  DCHK(d, d->ev_tmp.reserve(R * 16, false, st));
  DCHK(d, d->keep.reserve(R * 4, false, st));
  DCHK(d, d->keep_pos.reserve(R * 4, false, st));
  DCHK(d, d->f64_list.reserve(R * 4, false, st));
  ErrorCell zero{~0ull, 0u, 0u, ~0u, 0u};
  DCHK(d, hipMemcpyAsync(d->d_err.p, &zero, sizeof(zero), hipMemcpyHostToDevice, st));
  return ensure_table(d, n_rec);
}

// everything behind the per-record metadata: interning, value decode, compaction, append
int32_t finish_push(surge_device_decoder* d, int64_t n_rec) {
  hipStream_t st = d->stream;
  const size_t R = (size_t)n_rec;
  const uint8_t* dby = (const uint8_t*)d->d_bytes.p;
  ErrorCell* derr = (ErrorCell*)d->d_err.p;
  RecMeta* dmeta = (RecMeta*)d->meta.p;
  const unsigned rb = (unsigned)((n_rec + 255) / 256);
  hipLaunchKernelGGL(probe_kernel, dim3(rb), dim3(256), 0, st, dmeta, n_rec, table_of(d), (uint32_t*)d->new_slots.p, derr);
  ErrorCell ec;
  DCHK(d, hipMemcpyAsync(&ec, derr, sizeof(ec), hipMemcpyDeviceToHost, st));
  DCHK(d, hipStreamSynchronize(st));
  const uint32_t n_new = ec.n_new;
  if (n_new > 0) {
    // new keys in first-delivered order
    DCHK(d, d->sort_k_a.reserve((size_t)n_new * 4, false, st));
    DCHK(d, d->sort_k_b.reserve((size_t)n_new * 4, false, st));
    DCHK(d, d->sort_v_b.reserve((size_t)n_new * 4, false, st));
    DCHK(d, d->lens.reserve(((size_t)n_new + 1) * 8, false, st));
    DCHK(d, d->key_off.reserve((size_t)(d->n_keys + n_new + 1) * 8, true, st));
    DCHK(d, d->key_hash.reserve((size_t)(d->n_keys + n_new) * 8, true, st));
    size_t tb_sort = 0, tb_scan = 0;
    DCHK(d, rocprim::radix_sort_pairs(nullptr, tb_sort, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                      (size_t)n_new, 0u, 32u, st));
    DCHK(d, rocprim::exclusive_scan(nullptr, tb_scan, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)n_new + 1,
                                    rocprim::plus<int64_t>(), st));
    DCHK(d, d->temp.reserve(tb_sort > tb_scan ? tb_sort : tb_scan, false, st));
    const unsigned nb = (n_new + 256u) / 256u;
    hipLaunchKernelGGL(newkey_keys_kernel, dim3(nb), dim3(256), 0, st, (const uint32_t*)d->new_slots.p, n_new, table_of(d), (uint32_t*)d->sort_k_a.p);
    size_t tb = d->temp.cap;
    DCHK(d, rocprim::radix_sort_pairs(d->temp.p, tb, (const uint32_t*)d->sort_k_a.p, (uint32_t*)d->sort_k_b.p, (const uint32_t*)d->new_slots.p,
                                      (uint32_t*)d->sort_v_b.p, (size_t)n_new, 0u, 32u, st));
    hipLaunchKernelGGL(newkey_len_kernel, dim3(nb), dim3(256), 0, st, (const uint32_t*)d->sort_k_b.p, n_new, dmeta, (int64_t*)d->lens.p);
    tb = d->temp.cap;
    DCHK(d, rocprim::exclusive_scan(d->temp.p, tb, (const int64_t*)d->lens.p, (int64_t*)d->lens.p, (int64_t)0, (size_t)n_new + 1,
                                    rocprim::plus<int64_t>(), st));
    int64_t new_bytes = 0;
    DCHK(d, hipMemcpyAsync(&new_bytes, (int64_t*)d->lens.p + n_new, 8, hipMemcpyDeviceToHost, st));
    DCHK(d, hipStreamSynchronize(st));
    DCHK(d, d->arena.reserve((size_t)(d->arena_bytes + new_bytes) + 16, true, st));
    hipLaunchKernelGGL(newkey_assign_kernel, dim3(nb), dim3(256), 0, st, (const uint32_t*)d->sort_k_b.p, (const uint32_t*)d->sort_v_b.p, n_new, dmeta, dby,
                       (const int64_t*)d->lens.p, d->n_keys, d->arena_bytes, table_of(d), (uint8_t*)d->arena.p, (int64_t*)d->key_off.p,
                       (unsigned long long*)d->key_hash.p);
    d->n_keys += n_new;
    d->arena_bytes += new_bytes;
  }
  hipLaunchKernelGGL(resolve_kernel, dim3(rb), dim3(256), (size_t)kResolveStage, st, dmeta, n_rec, dby, table_of(d), (const uint8_t*)d->arena.p, (const int64_t*)d->key_off.p,
                     d->json ? (const surge_event_json_template*)d->d_tmpl.p : nullptr, (const surge::F64ParseTable*)d->d_ptab.p,
                     (int64_t*)d->agg_tmp.p, (uint4*)d->ev_tmp.p, (uint32_t*)d->keep.p, (uint32_t*)d->f64_list.p, derr);
  // the first-record marks of this push's new slots are spent
  if (n_new > 0) DCHK(d, hipMemsetAsync(d->t_first.p, 0xff, d->t_cap * 4, st));
  size_t tb_scan = 0;
  DCHK(d, rocprim::exclusive_scan(nullptr, tb_scan, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, R, rocprim::plus<uint32_t>(), st));
  DCHK(d, d->temp.reserve(tb_scan, false, st));
  size_t tb = d->temp.cap;
  DCHK(d, rocprim::exclusive_scan(d->temp.p, tb, (const uint32_t*)d->keep.p, (uint32_t*)d->keep_pos.p, 0u, R, rocprim::plus<uint32_t>(), st));
  uint32_t last_pos = 0, last_keep = 0;
  DCHK(d, hipMemcpyAsync(&last_pos, (uint32_t*)d->keep_pos.p + (R - 1), 4, hipMemcpyDeviceToHost, st));
  DCHK(d, hipMemcpyAsync(&last_keep, (uint32_t*)d->keep.p + (R - 1), 4, hipMemcpyDeviceToHost, st));
  DCHK(d, hipMemcpyAsync(&ec, derr, sizeof(ec), hipMemcpyDeviceToHost, st));
  DCHK(d, hipStreamSynchronize(st));
  d->counters[0] += n_rec;
  if (ec.first_bad != ~0ull) {
    const int64_t rec = (int64_t)(ec.first_bad >> 8);
    const uint32_t status = (uint32_t)(ec.first_bad & 0xff);
    RecMeta m;
    DCHK(d, hipMemcpy(&m, dmeta + rec, sizeof(m), hipMemcpyDeviceToHost));
    static const char* why[] = {"", "", "has a null key or value (not an event)", "is malformed (a length runs past its record or batch)",
                                "is not the JSON object the event template describes", "names an event type the template does not know",
                                "lacks a field the template names, or the field is not the number it should be",
                                "is not a 16-byte fixed event", "", "collides with another key on its 64-bit hash (decode this topic with the host decoder)"};
    // nothing of this push is delivered; keys it discovered stay interned (harmless: they are ids without events)
    return dfail(d, status == RS_COLLISION ? E_UNSUPPORTED : SURGE_E_CORRUPT,
                 "record " + std::to_string(rec) + " of the push (offset " + std::to_string(m.offset) + ") " + (status < 10 ? why[status] : "is bad"));
  }
  const int64_t kept = (int64_t)last_pos + last_keep;
  DCHK(d, d->r_agg.reserve((size_t)(d->n_records + kept) * 8 + 16, true, st));
  DCHK(d, d->r_ev.reserve((size_t)(d->n_records + kept) * 16 + 16, true, st));
  DCHK(d, d->r_off.reserve((size_t)(d->n_records + kept) * 8 + 16, true, st));
  hipLaunchKernelGGL(scatter_kernel, dim3(rb), dim3(256), 0, st, (const uint32_t*)d->keep.p, (const uint32_t*)d->keep_pos.p, n_rec, dmeta,
                     (const int64_t*)d->agg_tmp.p, (const uint4*)d->ev_tmp.p, d->n_records, (int64_t*)d->r_agg.p, (uint4*)d->r_ev.p, (int64_t*)d->r_off.p);
  DCHK(d, hipGetLastError());
  if (ec.n_f64_host > 0) {
    DCHK(d, hipStreamSynchronize(st));  // the scatter has to be done before the payloads are patched
    // Doubles the fast parser could not decide (more than 19 digits, or one of Eisel-Lemire's rare ambiguous products):
    // the host parses exactly those values with the library's host decoder and patches the payload in place
    std::vector<uint32_t> list(ec.n_f64_host);
    DCHK(d, hipMemcpy(list.data(), d->f64_list.p, (size_t)ec.n_f64_host * 4, hipMemcpyDeviceToHost));
    surge_event_json_template tmpl;
    DCHK(d, hipMemcpy(&tmpl, d->d_tmpl.p, sizeof(tmpl), hipMemcpyDeviceToHost));
    for (uint32_t i : list) {
      RecMeta m;
      uint32_t pos = 0;
      DCHK(d, hipMemcpy(&m, dmeta + i, sizeof(m), hipMemcpyDeviceToHost));
      DCHK(d, hipMemcpy(&pos, (uint32_t*)d->keep_pos.p + i, 4, hipMemcpyDeviceToHost));
      uint8_t ev[16];
      std::vector<uint8_t> value((size_t)m.val_len + 1);
      DCHK(d, hipMemcpy(value.data(), dby + m.val_off, (size_t)m.val_len, hipMemcpyDeviceToHost));  // (an LZ4 section exists decompressed on the device only)
      if (surge_event_json_decode(&tmpl, value.data(), m.val_len, ev) != 0)
        return dfail(d, SURGE_E_CORRUPT, "record at offset " + std::to_string(m.offset) + ": " + surge_event_json_last_error());
      DCHK(d, hipMemcpy((uint8_t*)d->r_ev.p + (size_t)(d->n_records + pos) * 16, ev, 16, hipMemcpyHostToDevice));
    }
    d->counters[3] += ec.n_f64_host;
  }
  d->n_records += kept;
  d->counters[1] += kept;
  d->counters[2] += n_rec - kept;
  DCHK(d, hipStreamSynchronize(st));
  return OK;
}

}  // namespace

extern "C" {

int32_t surge_device_decoder_push(surge_device_decoder* d, const uint8_t* bytes, const surge_batch_section* sections, int64_t n_sections) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  if (n_sections < 0 || (n_sections > 0 && (!bytes || !sections))) return dfail(d, E_INVALID, "bad argument");
  if (n_sections == 0) return OK;
  // the span of the arena this push needs, and every batch's first record index
  int64_t lo = INT64_MAX, hi = 0, n_rec = 0;
  std::vector<Section> secs;
  try {
    secs.resize((size_t)n_sections);
  } catch (const std::bad_alloc&) {
    return dfail(d, E_NOMEM, "out of host memory");
  }
  for (int64_t s = 0; s < n_sections; ++s) {
    const surge_batch_section& in = sections[s];
    if (in.byte_off < 0 || in.byte_len < 0 || in.n_records < 0) return dfail(d, E_INVALID, "negative section field");
    lo = in.byte_off < lo ? in.byte_off : lo;
    hi = in.byte_off + in.byte_len > hi ? in.byte_off + in.byte_len : hi;
    secs[(size_t)s] = Section{in.byte_off, in.byte_len, in.base_offset, in.n_records, 0, n_rec};
    n_rec += in.n_records;
  }
  if (n_rec == 0) return OK;
  if (n_rec >= (1ll << 32) - 1) return dfail(d, E_UNSUPPORTED, "more than 2^32 - 2 records in one push: push fewer sections at a time");
  for (Section& s : secs) s.byte_off -= lo;
  const int64_t n_raw = hi - lo;
  // LZ4 sections (codec 3: the batch's records section is still one LZ4 frame): the host reads the frame header and the
  // block size words, the device decodes the blocks.  Frames it cannot take block by block (blocks larger than 64 KiB,
  // dependent blocks) are decompressed here, on the host, and travel as plain bytes behind the raw span.
  std::vector<Lz4Block> blocks;
  std::vector<uint8_t> extra;
  int64_t area = 0;  // bytes of the device-side decompressed area handed out so far (multiples of 64 KiB)
  try {
    for (int64_t s = 0; s < n_sections; ++s) {
      if (sections[s].codec != 3 || sections[s].n_records == 0) continue;
      Section& sec = secs[(size_t)s];
      const uint8_t* f = bytes + sections[s].byte_off;
      const int64_t fl = sections[s].byte_len;
      bool device_ok = fl >= 7 && f[0] == 0x04 && f[1] == 0x22 && f[2] == 0x4D && f[3] == 0x18 && (f[4] >> 6) == 1 && (f[4] & 0x20) &&
                       ((f[5] >> 4) & 7) == 4;
      int64_t p = 6;
      if (device_ok) {
        if (f[4] & 0x08) p += 8;  // content size
        if (f[4] & 0x01) p += 4;  // dictionary id
        device_ok = p < fl && f[p] == (uint8_t)(surge_xxh32(f + 4, p - 4, 0) >> 8);
        ++p;
      }
      const size_t first_block = blocks.size();
      int32_t k = 0;
      while (device_ok) {
        if (fl - p < 4) { device_ok = false; break; }
        const uint32_t bs = (uint32_t)f[p] | ((uint32_t)f[p + 1] << 8) | ((uint32_t)f[p + 2] << 16) | ((uint32_t)f[p + 3] << 24);
        p += 4;
        if (bs == 0) break;  // EndMark
        const uint32_t size = bs & 0x7fffffffu;
        if ((int64_t)size > fl - p || size > (1u << 30)) { device_ok = false; break; }
        Lz4Block b;
        b.src_off = sec.byte_off + p;
        b.dst_off = area + (int64_t)k * kLz4BlockMax;
        b.src_len = (int32_t)size | (int32_t)(bs & 0x80000000u);
        b.section = (int32_t)s;
        b.last = 0;
        b.index = k++;
        blocks.push_back(b);
        p += size;
        if (f[4] & 0x10) p += 4;  // block checksum (not verified: the batch CRC already covers these bytes)
      }
      if (device_ok && k > 0) {
        blocks.back().last = 1;
        sec.byte_off = -1 - area;  // resolved below, once the raw span's final size is known
        sec.byte_len = 0;          // set by the kernel that decodes the frame's last block
        area += (int64_t)k * kLz4BlockMax;
      } else {
        blocks.resize(first_block);
        int64_t cap = fl * 8 + 1024, got;
        const size_t at = extra.size();
        while (true) {
          extra.resize(at + (size_t)cap);
          got = surge_lz4_frame_decompress(f, fl, extra.data() + at, cap);
          if (got != -6) break;
          cap *= 4;
          if (cap > (1ll << 31)) return dfail(d, SURGE_E_CORRUPT, "LZ4 batch expands beyond 2 GiB");
        }
        if (got < 0) return dfail(d, SURGE_E_CORRUPT, "bad LZ4 frame in the section at base offset " + std::to_string(sections[s].base_offset));
        extra.resize(at + (size_t)got);
        sec.byte_off = n_raw + (int64_t)at;
        sec.byte_len = got;
      }
    }
  } catch (const std::bad_alloc&) {
    return dfail(d, E_NOMEM, "out of host memory");
  }
  const int64_t n_bytes = n_raw + (int64_t)extra.size();              // what is staged and copied
  const int64_t area_base = (n_bytes + 15) & ~15ll;                    // where the device-decompressed frames start
  for (Section& s : secs)
    if (s.byte_off < 0) s.byte_off = area_base + (-1 - s.byte_off);
  int prev = 0;
  (void)hipGetDevice(&prev);
  struct Restore { int dev; ~Restore() { (void)hipSetDevice(dev); } } restore{prev};
  DCHK(d, hipSetDevice(d->device));
  hipStream_t st = d->stream;
  const size_t R = (size_t)n_rec;
  DCHK(d, d->d_bytes.reserve((size_t)(area_base + area) + 16, false, st));
  DCHK(d, d->d_sections.reserve(sizeof(Section) * (size_t)n_sections, false, st));
  DCHK(d, d->rec_pos.reserve(R * 8, false, st));
  DCHK(d, d->rec_end.reserve(R * 8, false, st));
  {
    const int32_t rc = begin_push(d, n_rec);
    if (rc != OK) return rc;
  }
  // H2D: straight out of the caller's arena when that is page-locked (surge_ingest_use_pinned_arena), else through the
  // decoder's own pinned staging (one extra host copy per push)
  hipPointerAttribute_t attr;
  const bool in_place = hipPointerGetAttributes(&attr, bytes + lo) == hipSuccess && attr.type == hipMemoryTypeHost;
  (void)hipGetLastError();  // a pageable pointer makes hipPointerGetAttributes fail: not an error of this call
  const size_t staged = in_place ? extra.size() : (size_t)n_bytes;
  if (staged > d->pinned_cap) {
    if (d->pinned) (void)hipHostFree(d->pinned);
    d->pinned = nullptr;
    d->pinned_cap = 0;
    DCHK(d, hipHostMalloc(&d->pinned, staged, hipHostMallocDefault));
    d->pinned_cap = staged;
  }
  if (in_place) {
    DCHK(d, hipMemcpyAsync(d->d_bytes.p, bytes + lo, (size_t)n_raw, hipMemcpyHostToDevice, st));
    if (!extra.empty()) {
      std::memcpy(d->pinned, extra.data(), extra.size());
      DCHK(d, hipMemcpyAsync((uint8_t*)d->d_bytes.p + n_raw, d->pinned, extra.size(), hipMemcpyHostToDevice, st));
    }
  } else {
    std::memcpy(d->pinned, bytes + lo, (size_t)n_raw);
    if (!extra.empty()) std::memcpy((uint8_t*)d->pinned + n_raw, extra.data(), extra.size());
    DCHK(d, hipMemcpyAsync(d->d_bytes.p, d->pinned, (size_t)n_bytes, hipMemcpyHostToDevice, st));
  }
  DCHK(d, hipMemcpyAsync(d->d_sections.p, secs.data(), sizeof(Section) * (size_t)n_sections, hipMemcpyHostToDevice, st));
  const uint8_t* dby = (const uint8_t*)d->d_bytes.p;
  Section* dsec = (Section*)d->d_sections.p;
  ErrorCell* derr = (ErrorCell*)d->d_err.p;
  RecMeta* dmeta = (RecMeta*)d->meta.p;
  if (!blocks.empty()) {
    DCHK(d, d->lz4_blocks.reserve(blocks.size() * sizeof(Lz4Block), false, st));
    DCHK(d, hipMemcpyAsync(d->lz4_blocks.p, blocks.data(), blocks.size() * sizeof(Lz4Block), hipMemcpyHostToDevice, st));
    DCHK(d, d->lz4_sizes.reserve(blocks.size() * 4, false, st));
    const int64_t nb = (int64_t)blocks.size();
    const unsigned grid = (unsigned)(nb < 8192 ? nb : 8192);
    const int32_t caps[3] = {16384, 32768, kLz4BlockMax};  // LDS per wave of the three launches
    for (int c = 0; c < 3; ++c)
      hipLaunchKernelGGL(lz4_block_kernel, dim3(grid), dim3(64), (size_t)caps[c], st, dby, (uint8_t*)d->d_bytes.p + area_base,
                         (const Lz4Block*)d->lz4_blocks.p, nb, (int32_t*)d->lz4_sizes.p, c, caps[c], dsec, derr);
    // a frame that does not decode fails the push HERE, before any key of the push is interned (`blocks` is host memory:
    // the copy has to be done before it goes out of scope anyway)
    ErrorCell lz;
    DCHK(d, hipMemcpyAsync(&lz, derr, sizeof(lz), hipMemcpyDeviceToHost, st));
    DCHK(d, hipStreamSynchronize(st));
    if (lz.lz4_bad != ~0u)
      return dfail(d, SURGE_E_CORRUPT, "bad LZ4 frame in the batch at base offset " + std::to_string(sections[lz.lz4_bad].base_offset) +
                                       " (malformed sequence, or a block that is not 64 KiB where it must be)");
  }
  const unsigned rb = (unsigned)((n_rec + 255) / 256);
  hipLaunchKernelGGL(chain_kernel, dim3((unsigned)((n_sections + 63) / 64)), dim3(64), 0, st, dby, (const Section*)dsec, n_sections, (int64_t*)d->rec_pos.p,
                     (int64_t*)d->rec_end.p, derr);
  hipLaunchKernelGGL(parse_kernel, dim3(rb), dim3(256), 0, st, dby, (const Section*)dsec, n_sections, (const int64_t*)d->rec_pos.p, (const int64_t*)d->rec_end.p,
                     n_rec, dmeta, derr);
  return finish_push(d, n_rec);
}

int32_t surge_device_decoder_push_records(surge_device_decoder* d, const uint8_t* keys, const int64_t* key_off, const uint8_t* values,
                                          const int64_t* value_off, const int64_t* offsets, int64_t n) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  if (n < 0 || (n > 0 && (!key_off || !value_off))) return dfail(d, E_INVALID, "bad argument");
  if (n == 0) return OK;
  if (n >= (1ll << 32) - 1) return dfail(d, E_UNSUPPORTED, "more than 2^32 - 2 records in one push");
  const int64_t kb = key_off[n] - key_off[0], vb = value_off[n] - value_off[0];
  if (kb < 0 || vb < 0 || (kb > 0 && !keys) || (vb > 0 && !values)) return dfail(d, E_INVALID, "bad key / value spans");
  int prev = 0;
  (void)hipGetDevice(&prev);
  struct Restore { int dev; ~Restore() { (void)hipSetDevice(dev); } } restore{prev};
  DCHK(d, hipSetDevice(d->device));
  hipStream_t st = d->stream;
  const size_t n_bytes = (size_t)(kb + vb), off_bytes = (size_t)(n + 1) * 8;
  const size_t stage = n_bytes + 2 * off_bytes + (offsets ? (size_t)n * 8 : 0) + 64;
  DCHK(d, d->d_bytes.reserve(n_bytes + 16, false, st));
  DCHK(d, d->rec_pos.reserve(off_bytes, false, st));  // reused as the device copies of key_off / value_off / offsets
  DCHK(d, d->rec_end.reserve(off_bytes, false, st));
  DCHK(d, d->d_sections.reserve((size_t)n * 8 + 8, false, st));
  {
    const int32_t rc = begin_push(d, n);
    if (rc != OK) return rc;
  }
  if (stage > d->pinned_cap) {
    if (d->pinned) (void)hipHostFree(d->pinned);
    d->pinned = nullptr;
    d->pinned_cap = 0;
    DCHK(d, hipHostMalloc(&d->pinned, stage, hipHostMallocDefault));
    d->pinned_cap = stage;
  }
  // pinned staging: [keys][values][key_off (rebased)][value_off (rebased)][offsets]
  uint8_t* pin = (uint8_t*)d->pinned;
  if (kb) std::memcpy(pin, keys + key_off[0], (size_t)kb);
  if (vb) std::memcpy(pin + kb, values + value_off[0], (size_t)vb);
  int64_t* p_ko = (int64_t*)(pin + ((n_bytes + 7) & ~(size_t)7));
  int64_t* p_vo = p_ko + (n + 1);
  int64_t* p_of = p_vo + (n + 1);
  for (int64_t i = 0; i <= n; ++i) { p_ko[i] = key_off[i] - key_off[0]; p_vo[i] = value_off[i] - value_off[0]; }
  if (offsets) std::memcpy(p_of, offsets, (size_t)n * 8);
  if (n_bytes) DCHK(d, hipMemcpyAsync(d->d_bytes.p, pin, n_bytes, hipMemcpyHostToDevice, st));
  DCHK(d, hipMemcpyAsync(d->rec_pos.p, p_ko, off_bytes, hipMemcpyHostToDevice, st));
  DCHK(d, hipMemcpyAsync(d->rec_end.p, p_vo, off_bytes, hipMemcpyHostToDevice, st));
  if (offsets) DCHK(d, hipMemcpyAsync(d->d_sections.p, p_of, (size_t)n * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(records_meta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t*)d->d_bytes.p, (const int64_t*)d->rec_pos.p,
                     (const int64_t*)d->rec_end.p, offsets ? (const int64_t*)d->d_sections.p : nullptr, kb, n, (RecMeta*)d->meta.p, (ErrorCell*)d->d_err.p);
  return finish_push(d, n);
}

int32_t surge_device_decoder_result(surge_device_decoder* d, int64_t* n_records, const int64_t** d_agg_idx, const void** d_events16,
                                    const int64_t** d_offsets, int64_t* n_keys) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  if (n_records) *n_records = d->n_records;
  if (d_agg_idx) *d_agg_idx = (const int64_t*)d->r_agg.p;
  if (d_events16) *d_events16 = d->r_ev.p;
  if (d_offsets) *d_offsets = (const int64_t*)d->r_off.p;
  if (n_keys) *n_keys = d->n_keys;
  return OK;
}

// result -> resident state: the composition a host would otherwise spell out (grow for the new keys, device group-by +
// fold, clear), behind one call so a JVM needs a single JNI crossing per poll
int32_t surge_replay_append_decoded(surge_replay_handle* h, surge_device_decoder* d, int64_t* n_events_out, int64_t* n_keys_out) {
  if (!h || !d) return dfail(d, E_INVALID, "NULL argument");
  if (n_events_out) *n_events_out = d->n_records;
  if (n_keys_out) *n_keys_out = d->n_keys;
  void* d_states = nullptr;
  int64_t n_agg = 0;
  int32_t rc = surge_replay_device_state(h, &d_states, &n_agg);
  if (rc != OK) return dfail(d, rc, surge_replay_last_error(h));
  if (d->n_keys > n_agg) {
    rc = surge_replay_grow(h, d->n_keys);
    if (rc != OK) return dfail(d, rc, surge_replay_last_error(h));
  }
  if (d->n_records > 0) {
    // the decoder's arrays are written on its stream and read on the handle's: make the hand-over explicit
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(d->device);
    const hipError_t e = hipStreamSynchronize(d->stream);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return dfail(d, E_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    rc = surge_replay_append_events_device(h, (const int64_t*)d->r_agg.p, d->r_ev.p, d->n_records);
    if (rc != OK) return dfail(d, rc, surge_replay_last_error(h));
    rc = surge_replay_synchronize(h);  // the arrays are reused by the next push
    if (rc != OK) return dfail(d, rc, surge_replay_last_error(h));
  }
  d->n_records = 0;
  return OK;
}

int32_t surge_device_decoder_clear(surge_device_decoder* d) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  d->n_records = 0;
  return OK;
}

int32_t surge_device_decoder_keys(surge_device_decoder* d, uint8_t* utf8_out, int64_t utf8_capacity, int64_t* key_off_out, int64_t* n_keys_out,
                                  int64_t* utf8_bytes_out) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  if (n_keys_out) *n_keys_out = d->n_keys;
  if (utf8_bytes_out) *utf8_bytes_out = d->arena_bytes;
  if (!utf8_out && !key_off_out) return OK;  // size query
  if (utf8_capacity < d->arena_bytes) return dfail(d, E_INVALID, "utf8_out is too small (see *utf8_bytes_out)");
  int prev = 0;
  (void)hipGetDevice(&prev);
  struct Restore { int dev; ~Restore() { (void)hipSetDevice(dev); } } restore{prev};
  DCHK(d, hipSetDevice(d->device));
  DCHK(d, hipStreamSynchronize(d->stream));
  if (utf8_out && d->arena_bytes > 0) DCHK(d, hipMemcpy(utf8_out, d->arena.p, (size_t)d->arena_bytes, hipMemcpyDeviceToHost));
  if (key_off_out) DCHK(d, hipMemcpy(key_off_out, d->key_off.p, (size_t)(d->n_keys + 1) * 8, hipMemcpyDeviceToHost));
  return OK;
}

int32_t surge_device_decoder_key_table(surge_device_decoder* d, const uint8_t** d_utf8, const int64_t** d_key_off) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  if (d_utf8) *d_utf8 = (const uint8_t*)d->arena.p;
  if (d_key_off) *d_key_off = (const int64_t*)d->key_off.p;
  return OK;
}

int32_t surge_device_decoder_counters(const surge_device_decoder* d, int64_t out[4]) {
  if (!d || !out) return E_INVALID;
  for (int i = 0; i < 4; ++i) out[i] = d->counters[i];
  return OK;
}

}  // extern "C"
