// ingest.cpp — events-topic ingest, host side of libsurge_replay.so.
// Kafka RecordBatch v2 + LZ4 frame + read_committed, restated from the published formats (kafka-clients
// 3.2.3 is not vendored under /root/reference: parity unpinned, see include/surge_ingest.h).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <time.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <immintrin.h>
#include <pthread.h>
#include <thread>
#include <vector>

#include "../../include/surge_ingest.h"

namespace {

constexpr int32_t OK = 0, E_INVALID = -1, E_NOMEM = -4, E_UNSUPPORTED = -5;

thread_local std::string g_err;

struct Rec {
  int64_t offset, key_off, value_off, agg_idx;
  int32_t key_len, value_len;
};

struct Batch {
  int64_t producer_id = -1;
  bool transactional = false;
  int decided = 1;  // 0 pending (open transaction), 1 committed / non-transactional, 2 aborted
  std::vector<Rec> recs;
  size_t next = 0;  // first record not yet drained
  // FRAMES mode (device decode): the batch's records section, verbatim, inside the arena
  int64_t sect_off = -1, sect_len = 0, base_offset = 0;
  int32_t count = 0, sect_codec = 0;
  // SURGE_INGEST_DEVICE_CRC: bytes in front of the section that travel with it (sect_off points at them): 8 = {crc, register after
  // the header bytes} written by the framer; 44 = the batch's own crc field and the 40 header bytes it covers, as received
  // (in-place framing: the device runs the whole CRC)
  int32_t crc_prefix = 0;
};

struct CrcTables {
  uint32_t t[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;  // Castagnoli, reflected
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
  }
};

const CrcTables& crc_tables() {
  static const CrcTables tables;  // initialised once, thread-safe (one decoder per partition thread)
  return tables;
}

#if defined(__x86_64__)
// The CRC32 instruction computes exactly this polynomial, 8 bytes per instruction with a latency of 3 cycles and a
// throughput of one per cycle: ONE dependent chain leaves two thirds of the unit idle (4.3 GB/s measured).  So a long
// buffer is cut into three equal blocks whose CRCs run interleaved in one loop; block A's register is then advanced
// over the length of B ("crc of A followed by |B| zero bytes", a linear map applied through four 256-entry tables built
// once from the byte table) and xor-ed into B's, likewise into C's — the classic three-way scheme of Intel's white paper
// / Mark Adler's crc32c.c, restated.
constexpr int64_t kCrcLong = 8192, kCrcShort = 256;

struct CrcShift {
  uint32_t t[4][256];
  // t[k][n]: the register value (n << 8k) after `len` zero bytes
  explicit CrcShift(int64_t len) {
    const CrcTables& T = crc_tables();
    for (int k = 0; k < 4; ++k)
      for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n << (8 * k);
        for (int64_t i = 0; i < len; ++i) c = (c >> 8) ^ T.t[0][c & 0xff];
        t[k][n] = c;
      }
  }
  uint32_t apply(uint32_t c) const { return t[0][c & 0xff] ^ t[1][(c >> 8) & 0xff] ^ t[2][(c >> 16) & 0xff] ^ t[3][c >> 24]; }
};

// three interleaved chains over data[0 .. 3 block), joined; returns the register after all three blocks
__attribute__((target("sse4.2"))) uint64_t crc32c_three_way(uint64_t c0, const uint8_t* data, int64_t block, const CrcShift& sh) {
  uint64_t c1 = 0, c2 = 0;
  const uint8_t* a = data;
  const uint8_t* b = data + block;
  const uint8_t* c = data + 2 * block;
  for (int64_t i = 0; i < block; i += 8) {
    uint64_t va, vb, vc;
    std::memcpy(&va, a + i, 8);
    std::memcpy(&vb, b + i, 8);
    std::memcpy(&vc, c + i, 8);
    c0 = __builtin_ia32_crc32di(c0, va);
    c1 = __builtin_ia32_crc32di(c1, vb);
    c2 = __builtin_ia32_crc32di(c2, vc);
  }
  c0 = sh.apply((uint32_t)c0) ^ c1;
  c0 = sh.apply((uint32_t)c0) ^ c2;
  return c0;
}

__attribute__((target("sse4.2"))) uint32_t crc32c_hw(const uint8_t* data, int64_t len) {
  static const CrcShift shift_long(kCrcLong), shift_short(kCrcShort);
  uint64_t c0 = 0xffffffffu;
  for (; len >= 3 * kCrcLong; data += 3 * kCrcLong, len -= 3 * kCrcLong) c0 = crc32c_three_way(c0, data, kCrcLong, shift_long);
  for (; len >= 3 * kCrcShort; data += 3 * kCrcShort, len -= 3 * kCrcShort) c0 = crc32c_three_way(c0, data, kCrcShort, shift_short);
  int64_t i = 0;
  for (; i + 8 <= len; i += 8) {
    uint64_t v;
    std::memcpy(&v, data + i, 8);
    c0 = __builtin_ia32_crc32di(c0, v);
  }
  uint32_t c32 = (uint32_t)c0;
  for (; i < len; ++i) c32 = __builtin_ia32_crc32qi(c32, data[i]);
  return c32 ^ 0xffffffffu;
}
#endif

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  bool need(int64_t n) {
    if (!ok || end - p < n) { ok = false; return false; }
    return true;
  }
  uint8_t u8() { return need(1) ? *p++ : 0; }
  int16_t i16() { if (!need(2)) return 0; int16_t v = (int16_t)((p[0] << 8) | p[1]); p += 2; return v; }
  int32_t i32() { if (!need(4)) return 0; uint32_t v = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; p += 4; return (int32_t)v; }
  int64_t i64() { if (!need(8)) return 0; uint64_t v = 0; for (int i = 0; i < 8; ++i) v = (v << 8) | p[i]; p += 8; return (int64_t)v; }
  // zig-zag varint (protobuf style), as Kafka's ByteUtils.readVarlong
  int64_t varlong() {
    uint64_t v = 0;
    int shift = 0;
    while (true) {
      if (!need(1) || shift > 63) { ok = false; return 0; }
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) break;
      shift += 7;
    }
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
};

// LZ4 block (sequence) decoder into dst[*op .. cap); matches may reach back into everything already written.
// LZ4_OK, LZ4_CORRUPT (malformed input) or LZ4_NOSPACE (dst too small: the caller grows it and retries).
enum { LZ4_OK = 0, LZ4_CORRUPT = 1, LZ4_NOSPACE = 2 };
int lz4_block(const uint8_t* ip, const uint8_t* iend, uint8_t* dst, int64_t* op_io, int64_t cap) {
  int64_t op = *op_io;
  while (ip < iend) {
    const uint8_t token = *ip++;
    int64_t lit = token >> 4;
    if (lit == 15) {
      uint8_t b;
      do {
        if (ip >= iend) return LZ4_CORRUPT;
        b = *ip++;
        lit += b;
      } while (b == 255);
    }
    if (iend - ip < lit) return LZ4_CORRUPT;
    if (cap - op < lit) return LZ4_NOSPACE;
    std::memcpy(dst + op, ip, (size_t)lit);
    ip += lit;
    op += lit;
    if (ip >= iend) break;  // the last sequence carries literals only
    if (iend - ip < 2) return LZ4_CORRUPT;
    const int64_t offset = ip[0] | (ip[1] << 8);
    ip += 2;
    if (offset == 0 || offset > op) return LZ4_CORRUPT;
    int64_t ml = token & 15;
    if (ml == 15) {
      uint8_t b;
      do {
        if (ip >= iend) return LZ4_CORRUPT;
        b = *ip++;
        ml += b;
      } while (b == 255);
    }
    ml += 4;
    if (cap - op < ml) return LZ4_NOSPACE;
    if (offset >= ml) {
      std::memcpy(dst + op, dst + op - offset, (size_t)ml);
    } else if (offset >= 8) {  // overlapping, but every 8-byte step reads bytes already final
      int64_t k = 0;
      for (; k + 8 <= ml; k += 8) std::memcpy(dst + op + k, dst + op + k - offset, 8);
      for (; k < ml; ++k) dst[op + k] = dst[op + k - offset];
    } else {
      for (int64_t k = 0; k < ml; ++k) dst[op + k] = dst[op + k - offset];  // short period: the overlap is the point
    }
    op += ml;
  }
  *op_io = op;
  return LZ4_OK;
}

}  // namespace

// The byte arena records / sections are copied into: a growable buffer whose memory comes from a pluggable allocator, so
// that a device decoder can have it PAGE-LOCKED (surge_ingest_set_allocator: its H2D copy then reads the arena in place
// instead of staging it through a pinned buffer of its own).
struct SliceOverflow : std::bad_alloc {};  // an external arena (a slice of a group's slab) is full

struct Arena {
  uint8_t* p = nullptr;
  size_t n = 0, cap = 0;
  void* (*alloc)(size_t) = nullptr;
  void (*release)(void*) = nullptr;
  bool external = false;  // a view into memory someone else owns (a slice of a surge_ingest_group's slab): never grown, never freed
  Arena() = default;
  Arena(const Arena&) = delete;
  Arena& operator=(const Arena&) = delete;
  ~Arena() { drop(); }
  void drop() {
    if (p && !external) (release ? release : std::free)(p);
    p = nullptr;
    n = cap = 0;
  }
  void view(uint8_t* at, size_t bytes) {
    external = true;
    p = at;
    cap = bytes;
    n = 0;
  }
  void reserve(size_t want) {  // an EMPTY arena gets room for `want` bytes in one allocation
    if (want <= cap) return;
    size_t c = cap ? cap : (size_t)1 << 16;
    while (c < want) c += c / 2 + 4096;
    uint8_t* fresh = (uint8_t*)(alloc ? alloc(c) : std::malloc(c));
    if (!fresh) throw std::bad_alloc();
    if (n) std::memcpy(fresh, p, n);
    if (p) (release ? release : std::free)(p);
    p = fresh;
    cap = c;
  }
  void reserve_exact(size_t want) {  // an EMPTY arena gets exactly `want` bytes (a group's slabs: page-locked memory is not cheap)
    if (want <= cap) return;
    uint8_t* fresh = (uint8_t*)(alloc ? alloc(want) : std::malloc(want));
    if (!fresh) throw std::bad_alloc();
    if (p) (release ? release : std::free)(p);
    p = fresh;
    n = 0;
    cap = want;
  }
  size_t size() const { return n; }
  const uint8_t* data() const { return p; }
  uint8_t* data() { return p; }
  void clear() { n = 0; }
  void append(const uint8_t* src, size_t len) {
    if (n + len > cap) {
      // a group sizes its members' slices for the whole feed; only host-side LZ4 (a group without SURGE_INGEST_DEVICE_LZ4)
      // can outgrow one: surge_ingest_group_feed undoes the feed and runs it again with more room
      if (external) throw SliceOverflow();
      size_t want = cap ? cap * 2 : (size_t)1 << 16;
      while (want < n + len) want *= 2;
      uint8_t* fresh = (uint8_t*)(alloc ? alloc(want) : std::malloc(want));
      if (!fresh) throw std::bad_alloc();
      if (n) std::memcpy(fresh, p, n);
      if (p) (release ? release : std::free)(p);
      p = fresh;
      cap = want;
    }
    if (len) std::memcpy(p + n, src, len);
    n += len;
  }
};

struct surge_ingest {
  int isolation = SURGE_INGEST_READ_COMMITTED;
  bool frames = false;  // SURGE_INGEST_FRAMES: records are not parsed here, their sections go to a surge_device_decoder
  bool device_lz4 = false;  // SURGE_INGEST_DEVICE_LZ4: ... and LZ4 frames stay compressed (the device decoder undoes them)
  bool device_crc = false;  // SURGE_INGEST_DEVICE_CRC: ... and a data batch's CRC-32C is finished and compared on the device
  bool inplace = false;     // (a group's in-place feed) `data` lies inside the slab this member's slice is a view of: sections stay where they were received
  std::string err;
  // FRAMES mode rotates through six arenas, one per feed: the sections a drain handed out stay where they are while
  // the next FIVE feeds fill the others, so host threads can frame fetches i + 1 .. i + 5 while a device decoder still
  // reads fetch i (surge_device_decoder_push_async keeps up to five pushes in flight).  The other modes only ever use
  // the first.
  static constexpr int kArenas = 6;
  Arena arenas[kArenas];
  int cur = 0;
  bool handed_out = false;  // a drain has handed out spans of arenas[cur] since the last switch
  int crc_threads = 1;      // surge_ingest_set_threads: host threads that verify the batches' CRC-32C of one feed
  // a member of a surge_ingest_group frames every feed into the slice of the group's slab it is given for that feed
  bool grouped = false;
  Arena ext[kArenas];
  uint8_t* ext_next = nullptr;
  size_t ext_next_cap = 0;
  bool slice_overflow = false;  // the last feed ran out of its slice (host-side LZ4 in a group: the group retries with more room)
  Arena& arena_now() { return grouped ? ext[cur] : arenas[cur]; }
  const Arena& arena_now() const { return grouped ? ext[cur] : arenas[cur]; }
  std::deque<Batch> queue;
  std::vector<std::string> keys;   // aggregate ids in first-seen order
  std::vector<uint64_t> key_hash;  // their hashes
  std::vector<int64_t> slots;      // open-addressing index over keys (power-of-two size, -1 = empty)
  std::vector<uint8_t> scratch;
  int64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {

int32_t fail(surge_ingest* g, int32_t code, const std::string& m) {
  if (g) g->err = m;
  g_err = m;
  return code;
}

inline uint64_t hash_bytes(const uint8_t* p, size_t n) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
  while (n >= 8) {
    uint64_t v;
    std::memcpy(&v, p, 8);
    h = (h ^ v) * 0xD6E8FEB86659FD93ull;
    h ^= h >> 32;
    p += 8;
    n -= 8;
  }
  uint64_t v = 0;
  if (n > 0) std::memcpy(&v, p, n);  // an empty key in an empty arena has a null pointer
  h = (h ^ v) * 0xD6E8FEB86659FD93ull;
  return h ^ (h >> 29);
}

void rehash(surge_ingest* g, size_t new_cap) {
  g->slots.assign(new_cap, -1);
  const size_t mask = new_cap - 1;
  for (size_t i = 0; i < g->keys.size(); ++i) {
    size_t s = (size_t)g->key_hash[i] & mask;
    while (g->slots[s] >= 0) s = (s + 1) & mask;
    g->slots[s] = (int64_t)i;
  }
}

int64_t intern(surge_ingest* g, const uint8_t* key, int32_t len) {
  size_t n = 0;
  while (n < (size_t)len && key[n] != (uint8_t)':') ++n;  // PartitionStringUpToColon
  if (g->slots.empty()) rehash(g, 1024);
  const uint64_t h = hash_bytes(key, n);
  size_t mask = g->slots.size() - 1;
  for (size_t s = (size_t)h & mask;; s = (s + 1) & mask) {
    const int64_t i = g->slots[s];
    if (i < 0) break;
    if (g->key_hash[(size_t)i] == h && g->keys[(size_t)i].size() == n && std::memcmp(g->keys[(size_t)i].data(), key, n) == 0) return i;
  }
  const int64_t idx = (int64_t)g->keys.size();
  g->keys.emplace_back((const char*)key, n);
  g->key_hash.push_back(h);
  if ((g->keys.size() + 1) * 2 > g->slots.size()) {
    rehash(g, g->slots.size() * 2);
  } else {
    size_t s = (size_t)h & mask;
    while (g->slots[s] >= 0) s = (s + 1) & mask;
    g->slots[s] = idx;
  }
  return idx;
}

// dense aggregate index of a record that is being delivered
inline int64_t deliver_idx(surge_ingest* g, Rec& r) {
  if (r.agg_idx == -2) r.agg_idx = intern(g, g->arena_now().data() + r.key_off, r.key_len);
  return r.agg_idx;
}

// number of records deliverable from the head of the queue
int64_t ready_count(const surge_ingest* g) {
  int64_t n = 0;
  for (const Batch& b : g->queue) {
    if (b.decided == 0) break;  // an open transaction: nothing behind it is stable yet
    if (b.decided == 1) n += b.sect_off >= 0 ? (int64_t)b.count : (int64_t)(b.recs.size() - b.next);
  }
  return n;
}

int32_t parse_records(surge_ingest* g, Batch& b, const uint8_t* data, int64_t len, int32_t count, int64_t base_offset,
                      bool control, int* control_type) {
  Reader r{data, data + len};
  if (count > 0 && !control) b.recs.reserve((size_t)(count < 65536 ? count : 65536));
  for (int32_t i = 0; i < count; ++i) {
    const int64_t rlen = r.varlong();
    if (!r.ok || rlen < 0 || !r.need(rlen)) return fail(g, SURGE_E_CORRUPT, "record length runs past the batch");
    Reader q{r.p, r.p + rlen};
    r.p += rlen;
    (void)q.u8();        // attributes
    (void)q.varlong();   // timestampDelta
    const int64_t offset_delta = q.varlong();
    const int64_t klen = q.varlong();
    if (!q.ok || klen < -1 || (klen >= 0 && !q.need(klen))) return fail(g, SURGE_E_CORRUPT, "bad record key");
    const uint8_t* key = q.p;
    if (klen > 0) q.p += klen;
    const int64_t vlen = q.varlong();
    if (!q.ok || vlen < -1 || (vlen >= 0 && !q.need(vlen))) return fail(g, SURGE_E_CORRUPT, "bad record value");
    const uint8_t* val = q.p;
    if (vlen > 0) q.p += vlen;
    const int64_t n_headers = q.varlong();
    if (!q.ok || n_headers < 0) return fail(g, SURGE_E_CORRUPT, "bad header count");
    for (int64_t hdr = 0; hdr < n_headers; ++hdr) {
      const int64_t hk = q.varlong();
      if (!q.ok || hk < 0 || !q.need(hk)) return fail(g, SURGE_E_CORRUPT, "bad header key");
      q.p += hk;
      const int64_t hv = q.varlong();
      if (!q.ok || hv < -1 || (hv > 0 && !q.need(hv))) return fail(g, SURGE_E_CORRUPT, "bad header value");
      if (hv > 0) q.p += hv;
    }
    g->counters[1] += 1;
    if (control) {
      // control record key: version int16, type int16 (0 = ABORT, 1 = COMMIT)
      if (klen >= 4) *control_type = (key[2] << 8) | key[3];
      continue;
    }
    if (klen == 0 && vlen == 0) {  // the producer's "flush" record (KafkaProducerActorImpl.scala:322-329)
      g->counters[5] += 1;
      continue;
    }
    Rec rec;
    rec.offset = base_offset + offset_delta;
    rec.key_len = (int32_t)klen;
    rec.value_len = (int32_t)vlen;
    rec.key_off = (int64_t)g->arena_now().size();
    if (klen > 0) g->arena_now().append(key, (size_t)klen);
    rec.value_off = (int64_t)g->arena_now().size();
    if (vlen > 0) g->arena_now().append(val, (size_t)vlen);
    rec.agg_idx = klen >= 0 ? -2 : -1;  // -2: interned when the record is DELIVERED (drain): the keys of aborted or
                                        // still-open transactions never enter the key table
    b.recs.push_back(rec);
  }
  return OK;
}

}  // namespace

extern "C" {

uint32_t surge_crc32c(const uint8_t* data, int64_t len) {
#if defined(__x86_64__)
  static const bool have_hw = __builtin_cpu_supports("sse4.2");
  if (have_hw) return crc32c_hw(data, len);
#endif
  const CrcTables& T = crc_tables();
  uint32_t c = 0xffffffffu;
  int64_t i = 0;
  for (; i + 8 <= len; i += 8) {  // slicing-by-8
    const uint32_t lo = c ^ ((uint32_t)data[i] | ((uint32_t)data[i + 1] << 8) | ((uint32_t)data[i + 2] << 16) | ((uint32_t)data[i + 3] << 24));
    c = T.t[7][lo & 0xff] ^ T.t[6][(lo >> 8) & 0xff] ^ T.t[5][(lo >> 16) & 0xff] ^ T.t[4][lo >> 24] ^
        T.t[3][data[i + 4]] ^ T.t[2][data[i + 5]] ^ T.t[1][data[i + 6]] ^ T.t[0][data[i + 7]];
  }
  for (; i < len; ++i) c = (c >> 8) ^ T.t[0][(c ^ data[i]) & 0xff];
  return c ^ 0xffffffffu;
}

/* table walk only (tests compare it with the instruction path) */
uint32_t surge_crc32c_portable(const uint8_t* data, int64_t len) {
  const CrcTables& T = crc_tables();
  uint32_t c = 0xffffffffu;
  for (int64_t i = 0; i < len; ++i) c = (c >> 8) ^ T.t[0][(c ^ data[i]) & 0xff];
  return c ^ 0xffffffffu;
}

int64_t surge_lz4_frame_decompress(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap) {
  if (!src || n < 7 || (!dst && cap > 0)) return E_INVALID;
  const uint8_t* p = src;
  const uint8_t* end = src + n;
  if (!(p[0] == 0x04 && p[1] == 0x22 && p[2] == 0x4D && p[3] == 0x18)) return SURGE_E_CORRUPT;  // 0x184D2204 LE
  p += 4;
  const uint8_t flg = *p++;
  p++;  // BD: block maximum size, not needed to decode
  if ((flg >> 6) != 1) return SURGE_E_CORRUPT;  // version 01
  const bool block_checksum = flg & 0x10, content_size = flg & 0x08, content_checksum = flg & 0x04, dict_id = flg & 0x01;
  const uint8_t* const descriptor = p - 2;
  if (content_size) p += 8;
  if (dict_id) p += 4;
  if (p >= end) return SURGE_E_CORRUPT;
  // header checksum: second byte of XXH32(descriptor); kafka-clients verifies it for message format v2
  if (*p != (uint8_t)(surge_xxh32(descriptor, (int64_t)(p - descriptor), 0) >> 8)) return SURGE_E_CORRUPT;
  p += 1;
  int64_t op = 0;
  while (true) {
    if (end - p < 4) return SURGE_E_CORRUPT;
    const uint32_t bs = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    p += 4;
    if (bs == 0) break;  // EndMark
    const uint32_t size = bs & 0x7fffffffu;
    if ((int64_t)size > end - p) return SURGE_E_CORRUPT;
    if (bs & 0x80000000u) {  // stored uncompressed
      if (cap - op < (int64_t)size) return -6;
      std::memcpy(dst + op, p, size);
      op += size;
    } else {
      const int rc = lz4_block(p, p + size, dst, &op, cap);
      if (rc == LZ4_NOSPACE) return -6;
      if (rc != LZ4_OK) return SURGE_E_CORRUPT;
    }
    p += size;
    if (block_checksum) p += 4;
  }
  if (content_checksum) p += 4;
  return op;
}

int32_t surge_ingest_create(int32_t isolation_level, surge_ingest** out) {
  if (!out) return fail(nullptr, E_INVALID, "out is NULL");
  *out = nullptr;
  const bool frames = (isolation_level & (SURGE_INGEST_FRAMES | SURGE_INGEST_DEVICE_LZ4 | SURGE_INGEST_DEVICE_CRC)) != 0;
  const bool device_lz4 = (isolation_level & SURGE_INGEST_DEVICE_LZ4) != 0;
  const bool device_crc = (isolation_level & SURGE_INGEST_DEVICE_CRC) != 0;
  isolation_level &= ~(SURGE_INGEST_FRAMES | SURGE_INGEST_DEVICE_LZ4 | SURGE_INGEST_DEVICE_CRC);
  if (isolation_level != SURGE_INGEST_READ_UNCOMMITTED && isolation_level != SURGE_INGEST_READ_COMMITTED)
    return fail(nullptr, E_INVALID, "unknown isolation level");
  surge_ingest* g = new (std::nothrow) surge_ingest();
  if (!g) return fail(nullptr, E_NOMEM, "out of host memory");
  g->isolation = isolation_level;
  g->frames = frames;
  g->device_lz4 = device_lz4;
  g->device_crc = device_crc;
  *out = g;
  return OK;
}

int32_t surge_ingest_destroy(surge_ingest* g) {
  delete g;
  return OK;
}

const char* surge_ingest_last_error(const surge_ingest* g) { return g ? g->err.c_str() : g_err.c_str(); }

namespace {

// The whole batches at the front of data[0, len), found the way surge_ingest_feed's walk finds them (it stops where this
// stops: a batchLength below the header size, a cut batch, a magic other than 2), and for each whether its CRC-32C holds.
void verify_crcs_in_parallel(const uint8_t* data, int64_t len, int n_threads, std::vector<uint8_t>* ok) {
  struct Span { const uint8_t* from; int64_t n; uint32_t crc; };
  std::vector<Span> spans;
  int64_t pos = 0;
  while (len - pos >= 12) {
    const uint8_t* h = data + pos;
    const int32_t batch_len = (int32_t)(((uint32_t)h[8] << 24) | ((uint32_t)h[9] << 16) | ((uint32_t)h[10] << 8) | h[11]);
    if (batch_len < 49 || len - pos - 12 < batch_len) break;
    if (h[16] != 2) break;
    const uint32_t crc = ((uint32_t)h[17] << 24) | ((uint32_t)h[18] << 16) | ((uint32_t)h[19] << 8) | h[20];
    spans.push_back(Span{h + 21, (int64_t)batch_len - 9, crc});  // attributes .. end of the batch
    pos += 12 + (int64_t)batch_len;
  }
  ok->assign(spans.size(), 0);
  if (spans.empty()) return;
  (void)surge_crc32c((const uint8_t*)"", 0);  // initialise the dispatch / tables before threads race for them
  std::atomic<size_t> next{0};
  auto work = [&]() {
    constexpr size_t kGrab = 32;
    for (size_t b = next.fetch_add(kGrab); b < spans.size(); b = next.fetch_add(kGrab))
      for (size_t k = b; k < spans.size() && k < b + kGrab; ++k) (*ok)[k] = surge_crc32c(spans[k].from, spans[k].n) == spans[k].crc;
  };
  std::vector<std::thread> th;
  try {
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work);
  } catch (...) {  // the threads that did start, and this one, do the work
  }
  work();
  for (std::thread& t : th) t.join();
}

}  // namespace

int32_t surge_ingest_set_threads(surge_ingest* g, int32_t n_threads) {
  if (!g || n_threads < 1 || n_threads > 64) return fail(g, E_INVALID, "threads must be in 1 .. 64");
  g->crc_threads = n_threads;
  return OK;
}

int32_t surge_ingest_feed(surge_ingest* g, const uint8_t* data, int64_t len, int64_t* consumed_out) {
  if (!g) return fail(nullptr, E_INVALID, "handle is NULL");
  if (len < 0 || (!data && len > 0)) return fail(g, E_INVALID, "bad buffer");
  if (consumed_out) *consumed_out = 0;
  if (g->frames && (g->handed_out || g->grouped)) {
    // switch arenas: the sections still queued (open transactions, undrained batches) move along, the ones the last
    // drain handed out stay untouched in the arena this feed leaves behind (valid through the five feeds after it).  A feed
    // that follows no drain keeps appending where the last one stopped: nothing is copied.
    try {
      Arena& next = g->grouped ? g->ext[(g->cur + 1) % surge_ingest::kArenas] : g->arenas[(g->cur + 1) % surge_ingest::kArenas];
      const Arena& prev = g->arena_now();
      if (g->grouped) next.view(g->ext_next, g->ext_next_cap); else next.clear();
      if (!g->grouped) {  // room for the whole feed at once: a page-locked arena that doubles its way up costs an allocation per step
        size_t queued = 0;
        for (const Batch& qb : g->queue) queued += qb.sect_off >= 0 ? (size_t)qb.sect_len : 0;
        next.reserve(queued + (size_t)len);
      }
      for (Batch& qb : g->queue) {
        if (qb.sect_off < 0) continue;
        const int64_t at = (int64_t)next.size();
        next.append(prev.data() + qb.sect_off, (size_t)qb.sect_len);
        qb.sect_off = at;
      }
      g->cur = (g->cur + 1) % surge_ingest::kArenas;
      g->handed_out = false;
    } catch (const SliceOverflow&) {
      g->slice_overflow = true;
      return fail(g, E_NOMEM, "the group's slice for this partition is full");
    } catch (const std::bad_alloc&) {
      return fail(g, E_NOMEM, "out of host memory while decoding");
    }
  } else if (g->queue.empty()) {
    // nothing queued and (FRAMES: no span of this arena handed out since the switch) nothing to keep: start over
    g->arena_now().clear();
  }
  int64_t pos = 0;
  // With more than one CRC thread the batches' checksums are verified up front, in parallel (a batch's CRC depends on
  // nothing but its own bytes; the walk below then looks the verdicts up in order, so what is reported, and when, is
  // exactly what the one-thread walk reports).
  std::vector<uint8_t> crc_ok;
  if (g->crc_threads > 1 && len >= (1 << 20) && !g->device_crc) {
    try {
      verify_crcs_in_parallel(data, len, g->crc_threads, &crc_ok);
    } catch (...) {  // no memory / no thread: the walk computes the CRCs itself
      crc_ok.clear();
    }
  }
  size_t batch_no = 0;
  // open transactions' batches in the queue (few at a feed's start; a marker looks for its producer's among them from the
  // back and stops at the oldest open one — the first version walked the whole queue per marker: every batch of the feed)
  int64_t n_open = 0;
  for (const Batch& qb : g->queue) n_open += qb.decided == 0;
  // A failure in batch k leaves batches 0..k-1 of this buffer decoded and queued: report them as consumed so a
  // caller that retries (or skips the bad batch) never feeds them twice.
  auto bail = [&](int32_t code, const char* msg) {
    if (consumed_out) *consumed_out = pos;
    return fail(g, code, msg);
  };
  try {
    while (len - pos >= 12) {
      Reader h{data + pos, data + len};
      const int64_t base_offset = h.i64();
      const int32_t batch_len = h.i32();
      if (batch_len < 49) return bail(SURGE_E_CORRUPT, "batchLength below the v2 header size");
      if (len - pos - 12 < batch_len) break;  // partial batch: wait for more bytes
      const uint8_t* body = data + pos + 12;
      __builtin_prefetch(body + batch_len);       // the next batch's header: a fresh cache line every few KB of a buffer that was just received
      __builtin_prefetch(body + batch_len + 64);
      Reader r{body, body + batch_len};
      (void)r.i32();  // partitionLeaderEpoch
      const uint8_t magic = r.u8();
      if (magic != 2) return bail(E_UNSUPPORTED, "only message format v2 (magic 2) is supported");
      const uint32_t crc = (uint32_t)r.i32();
      // SURGE_INGEST_DEVICE_CRC: the CRC of a data batch (the control bit is the 0x20 of the attributes, the two bytes behind
      // the CRC) is only STARTED here — over the 40 header bytes it covers in front of the records section — and finished
      // where the section's bytes go anyway: on the device.  What this thread still touches of a batch is its header.
      const bool defer_crc = g->device_crc && batch_len >= 49 && !(r.p[1] & 0x20);
      uint32_t crc_state = 0;
      if (defer_crc) {
        if (!g->inplace) crc_state = ~surge_crc32c(r.p, 40);  // the register (not finalised) after attributes .. recordCount
      } else {
        const bool crc_good = batch_no < crc_ok.size() ? crc_ok[batch_no] != 0 : surge_crc32c(r.p, r.end - r.p) == crc;
        if (!crc_good) return bail(SURGE_E_CORRUPT, "record batch CRC-32C mismatch");
      }
      ++batch_no;
      const int16_t attrs = r.i16();
      (void)r.i32();  // lastOffsetDelta
      (void)r.i64();  // baseTimestamp
      (void)r.i64();  // maxTimestamp
      const int64_t producer_id = r.i64();
      (void)r.i16();  // producerEpoch
      (void)r.i32();  // baseSequence
      const int32_t count = r.i32();
      if (!r.ok || count < 0) return bail(SURGE_E_CORRUPT, "truncated batch header");
      const int codec = attrs & 7;
      const bool transactional = attrs & 0x10, control = attrs & 0x20;
      // (a header that is only verified later, on the device, must not drive anything out of bounds before that: a record takes
      // at least 7 bytes, and LZ4 expands at most 255 x)
      if (defer_crc && (int64_t)count > ((r.end - r.p) / 7 + 1) * (codec ? 255 : 1))
        return bail(SURGE_E_CORRUPT, "recordCount impossible for the batch's size (damaged header)");
      const uint8_t* recs = r.p;
      int64_t recs_len = r.end - r.p;
      int sect_codec = 0;
      if (codec == 3 && g->device_lz4 && !control) {
        sect_codec = 3;  // the frame travels as it is: the device decoder walks its blocks and decodes them on the GPU
      } else if (codec == 3 && g->inplace && !control) {
        return bail(E_UNSUPPORTED, "an in-place feed leaves the sections where they were received: lz4 topics need SURGE_INGEST_DEVICE_LZ4");
      } else if (codec == 3) {
        // first guess: the frame's content-size field when the producer wrote one, else 8x (Kafka's LZ4 output
        // stream omits it); a too-small guess comes back as -6 (out of space) and is grown, never as "corrupt"
        int64_t cap = recs_len * 8 + 1024;
        if (recs_len >= 15 && (recs[4] & 0x08)) {
          uint64_t cs = 0;
          for (int k = 7; k >= 0; --k) cs = (cs << 8) | recs[6 + k];
          if (cs > 0 && cs <= (1ull << 31)) cap = (int64_t)cs;
        }
        int64_t got;
        while (true) {
          g->scratch.resize((size_t)cap);
          got = surge_lz4_frame_decompress(recs, recs_len, g->scratch.data(), cap);
          if (got != -6) break;
          cap *= 4;
          if (cap > (1ll << 31)) return bail(SURGE_E_CORRUPT, "LZ4 batch expands beyond 2 GiB");
        }
        if (got < 0) return bail(SURGE_E_CORRUPT, "bad LZ4 frame in record batch");
        g->counters[6] += got;
        recs = g->scratch.data();
        recs_len = got;
      } else if (codec != 0) {
        return bail(E_UNSUPPORTED, "compression codec not supported (only none and lz4; the reference publishes lz4)");
      }
      Batch b;
      b.producer_id = producer_id;
      b.transactional = transactional;
      int control_type = -1;
      if (g->frames && !control) {
        // device decode: the records section travels as it is (the device chains and parses the records)
        b.sect_off = (int64_t)g->arena_now().size();
        b.sect_len = recs_len;
        b.base_offset = base_offset;
        b.count = count;
        b.sect_codec = sect_codec;
        if (defer_crc && recs != r.p) return bail(E_UNSUPPORTED, "SURGE_INGEST_DEVICE_CRC needs the sections to travel as they are on the wire (SURGE_INGEST_DEVICE_LZ4 for lz4 topics)");
        if (g->inplace) {
          // nothing is copied: the section is where the fetch response was received (inside the slab this slice is a view of);
          // with the device CRC the 44 bytes in front of it — the crc field and the header bytes it covers — go along as they are
          b.crc_prefix = defer_crc ? 44 : 0;
          b.sect_off = (int64_t)(recs - g->arena_now().data()) - b.crc_prefix;
          b.sect_len = recs_len + b.crc_prefix;
        } else {
          if (defer_crc) {
            const uint32_t pre[2] = {crc, crc_state};
            g->arena_now().append((const uint8_t*)pre, 8);
            b.sect_len += 8;
            b.crc_prefix = 8;
          }
          g->arena_now().append(recs, (size_t)recs_len);
        }
        g->counters[1] += count;
      } else {
        const int32_t rc = parse_records(g, b, recs, recs_len, count, base_offset, control, &control_type);
        if (rc != OK) {
          if (consumed_out) *consumed_out = pos;
          return rc;
        }
      }
      g->counters[0] += 1;
      if (control) {
        g->counters[4] += 1;
        if (control_type == 0 || control_type == 1) {  // ABORT / COMMIT ends this producer's open transaction
          int64_t still_to_see = n_open;
          for (auto qb = g->queue.rbegin(); qb != g->queue.rend() && still_to_see > 0; ++qb) {
            if (qb->decided != 0) continue;
            --still_to_see;
            if (qb->producer_id == producer_id) {
              qb->decided = control_type == 1 ? 1 : 2;
              --n_open;
              if (control_type == 0) g->counters[3] += qb->sect_off >= 0 ? (int64_t)qb->count : (int64_t)qb->recs.size();
            }
          }
        }
      } else {
        b.decided = (transactional && g->isolation == SURGE_INGEST_READ_COMMITTED) ? 0 : 1;
        if (!b.recs.empty() || b.count > 0) {
          n_open += b.decided == 0;
          g->queue.push_back(std::move(b));
        }
      }
      pos += 12 + batch_len;
    }
  } catch (const SliceOverflow&) {
    if (consumed_out) *consumed_out = pos;
    g->slice_overflow = true;
    return fail(g, E_NOMEM, "the group's slice for this partition is full");
  } catch (const std::bad_alloc&) {
    if (consumed_out) *consumed_out = pos;
    return fail(g, E_NOMEM, "out of host memory while decoding");
  }
  if (consumed_out) *consumed_out = pos;
  g->counters[7] = n_open;
  return OK;
}

int64_t surge_ingest_ready(const surge_ingest* g) { return g ? ready_count(g) : 0; }

int32_t surge_ingest_drain(surge_ingest* g, int64_t max, surge_ingest_record* out, int64_t* n_out) {
  if (!g || !n_out || max < 0 || (!out && max > 0)) return fail(g, E_INVALID, "bad argument");
  if (g->frames) return fail(g, -2, "this decoder frames batches for a surge_device_decoder (SURGE_INGEST_FRAMES): use surge_ingest_drain_sections");
  int64_t n = 0;
  while (n < max && !g->queue.empty()) {
    Batch& b = g->queue.front();
    if (b.decided == 0) break;
    if (b.decided == 2) { g->queue.pop_front(); continue; }
    while (n < max && b.next < b.recs.size()) {
      Rec& r = b.recs[b.next++];
      out[n].offset = r.offset; out[n].agg_idx = deliver_idx(g, r); out[n].key_off = r.key_off; out[n].key_len = r.key_len;
      out[n].value_len = r.value_len; out[n].value_off = r.value_off;
      ++n;
    }
    if (b.next == b.recs.size()) g->queue.pop_front();
  }
  g->counters[2] += n;
  *n_out = n;
  return OK;
}

const uint8_t* surge_ingest_arena(const surge_ingest* g) { return g ? g->arena_now().data() : nullptr; }

int32_t surge_ingest_set_allocator(surge_ingest* g, void* (*alloc)(size_t), void (*release)(void*)) {
  if (!g || !alloc != !release) return fail(g, E_INVALID, "bad argument");
  for (const Arena& a : g->arenas)
    if (a.cap) return fail(g, -2, "surge_ingest_set_allocator after the first feed");
  for (Arena& a : g->arenas) {
    a.alloc = alloc;
    a.release = release;
  }
  return OK;
}

int32_t surge_ingest_drain_fixed16(surge_ingest* g, int64_t max, int64_t* agg_idx_out, void* events16_out,
                                   int64_t* offsets_out, int64_t* n_out) {
  if (!g || !n_out || max < 0 || ((!agg_idx_out || !events16_out) && max > 0)) return fail(g, E_INVALID, "bad argument");
  // validate before popping anything
  if (g->frames) return fail(g, -2, "this decoder frames batches for a surge_device_decoder (SURGE_INGEST_FRAMES): use surge_ingest_drain_sections");
  int64_t avail = 0;
  for (const Batch& b : g->queue) {
    if (b.decided == 0) break;
    if (b.decided == 2) continue;
    for (size_t i = b.next; i < b.recs.size() && avail < max; ++i, ++avail)
      if (b.recs[i].value_len != 16 || b.recs[i].agg_idx == -1)
        return fail(g, E_INVALID, "record value is not a 16-byte fixed event (or the key is null)");
    if (avail >= max) break;
  }
  int64_t n = 0;
  uint8_t* ev = (uint8_t*)events16_out;
  while (n < max && !g->queue.empty()) {
    Batch& b = g->queue.front();
    if (b.decided == 0) break;
    if (b.decided == 2) { g->queue.pop_front(); continue; }
    while (n < max && b.next < b.recs.size()) {
      Rec& r = b.recs[b.next++];
      agg_idx_out[n] = deliver_idx(g, r);
      std::memcpy(ev + n * 16, g->arena_now().data() + r.value_off, 16);
      if (offsets_out) offsets_out[n] = r.offset;
      ++n;
    }
    if (b.next == b.recs.size()) g->queue.pop_front();
  }
  g->counters[2] += n;
  *n_out = n;
  return OK;
}

int32_t surge_ingest_drain_json(surge_ingest* g, int64_t max, const surge_event_json_template* tmpl, int64_t* agg_idx_out,
                                void* events16_out, int64_t* offsets_out, int64_t* n_out) {
  if (!g || !n_out || !tmpl || max < 0 || ((!agg_idx_out || !events16_out) && max > 0)) return fail(g, E_INVALID, "bad argument");
  if (g->frames) return fail(g, -2, "this decoder frames batches for a surge_device_decoder (SURGE_INGEST_FRAMES): use surge_ingest_drain_sections");
  if (surge_event_json_validate(tmpl) != 0) return fail(g, E_INVALID, std::string("event template: ") + surge_event_json_last_error());
  // decode before popping anything: a value that does not decode leaves the queue as it was
  uint8_t* ev = (uint8_t*)events16_out;
  int64_t avail = 0;
  for (const Batch& b : g->queue) {
    if (b.decided == 0) break;
    if (b.decided == 2) continue;
    for (size_t i = b.next; i < b.recs.size() && avail < max; ++i, ++avail) {
      const Rec& r = b.recs[i];
      if (r.agg_idx == -1 || r.value_len < 0)
        return fail(g, SURGE_E_CORRUPT, "record at offset " + std::to_string(r.offset) + " has a null key or value (not an event)");
      const int32_t rc = surge_event_json_decode(tmpl, g->arena_now().data() + r.value_off, r.value_len, ev + avail * 16);
      if (rc != OK)
        return fail(g, SURGE_E_CORRUPT, "record at offset " + std::to_string(r.offset) + ": " + surge_event_json_last_error());
    }
    if (avail >= max) break;
  }
  int64_t n = 0;
  while (n < max && !g->queue.empty()) {
    Batch& b = g->queue.front();
    if (b.decided == 0) break;
    if (b.decided == 2) { g->queue.pop_front(); continue; }
    while (n < max && b.next < b.recs.size()) {
      Rec& r = b.recs[b.next++];
      agg_idx_out[n] = deliver_idx(g, r);
      if (offsets_out) offsets_out[n] = r.offset;
      ++n;
    }
    if (b.next == b.recs.size()) g->queue.pop_front();
  }
  g->counters[2] += n;
  *n_out = n;
  return OK;
}

int32_t surge_ingest_drain_sections(surge_ingest* g, int64_t max_sections, surge_batch_section* out, int64_t* n_out) {
  if (!g || !n_out || max_sections < 0 || (!out && max_sections > 0)) return fail(g, E_INVALID, "bad argument");
  if (!g->frames) return fail(g, -2, "surge_ingest_drain_sections needs a decoder created with SURGE_INGEST_FRAMES");
  int64_t n = 0, recs = 0;
  while (n < max_sections && !g->queue.empty()) {
    Batch& b = g->queue.front();
    if (b.decided == 0) break;  // an open transaction: nothing behind it is stable yet
    if (b.decided == 1) {
      out[n].byte_off = b.sect_off + b.crc_prefix;
      out[n].byte_len = b.sect_len - b.crc_prefix;
      out[n].base_offset = b.base_offset;
      out[n].n_records = b.count;
      out[n].codec = b.sect_codec | (b.crc_prefix == 8 ? SURGE_SECTION_CRC_PENDING : b.crc_prefix == 44 ? SURGE_SECTION_CRC_WIRE : 0);
      recs += b.count;
      ++n;
    }
    g->queue.pop_front();
  }
  g->counters[2] += recs;  // handed to the device decoder (its own counters tell flush records from events)
  if (n > 0) g->handed_out = true;
  *n_out = n;
  return OK;
}

}  // extern "C" (the group's type)

// A consumer's partitions framed as ONE unit: every feed lays the partitions' records sections out in one slab — the
// group rotates through six, like a single framer's arenas — so a fetch response reaches the device in one copy.
//
// Threads: the group keeps a POOL of framing threads for its lifetime (started at the first feed that asks for them;
// a feed hands them one job and takes part in it itself) — a consumer feeds every couple of milliseconds, and creating
// and joining seven threads per feed cost more than some feeds' framing.
//
// Failure: a feed either frames every partition or leaves the group exactly as it was.  What a member's feed changes —
// its queue (open transactions move to the new slice), its counters, which arena is current — is saved before the feed
// (a FRAMES queue holds only the few batches of open transactions: cheap) and put back when any partition fails or the
// caller's section table is too small; the slab the failed feed wrote into is the one the next feed writes into again.
namespace {

struct FramingPool {
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::vector<std::thread> th;
  const std::function<void()>* job = nullptr;
  uint64_t gen = 0;
  int want = 0;     // workers 0 .. want-1 take part in the current job
  int pending = 0;  // ... and have not finished it yet
  bool stop = false;

  void worker(int idx) {
    char name[16];
    snprintf(name, sizeof name, "surge-frame-%d", idx);  // (shows in /proc/<pid>/task/*/comm: bench.py's per-thread CPU table)
    pthread_setname_np(pthread_self(), name);
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_job.wait(lk, [&] { return stop || gen != seen; });
      if (stop) return;
      seen = gen;
      if (idx >= want) continue;
      const std::function<void()>* j = job;
      lk.unlock();
      (*j)();  // (catches everything itself)
      lk.lock();
      if (--pending == 0) cv_done.notify_one();
    }
  }
  // Runs `work` on this thread and on up to `helpers` pool threads; returns when all of them have left it.
  void run(const std::function<void()>& work, int helpers) {
    {
      std::lock_guard<std::mutex> lk(mu);
      while ((int)th.size() < helpers) {
        try {
          const int idx = (int)th.size();
          th.emplace_back([this, idx] { worker(idx); });
        } catch (...) {  // std::system_error: the threads that did start — and the caller — do the work
          break;
        }
      }
      if (helpers > (int)th.size()) helpers = (int)th.size();
      job = &work;
      want = pending = helpers;
      ++gen;
    }
    if (helpers > 0) cv_job.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
    job = nullptr;
  }
  ~FramingPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_job.notify_all();
    for (std::thread& t : th) t.join();
  }
};

struct MemberSave {
  std::deque<Batch> queue;
  int64_t counters[8];
  int cur;
  bool handed_out;
};

}  // namespace

struct surge_ingest_group {
  std::vector<surge_ingest*> g;
  Arena slabs[surge_ingest::kArenas];
  int cur = 0;
  std::string err;
  std::vector<std::vector<surge_batch_section>> drained;
  std::vector<int64_t> off;
  std::vector<MemberSave> saved;
  std::vector<int32_t> status;
  std::vector<int64_t> consumed;
  // in-place feeds (surge_ingest_group_receive_buffer): the slab the caller is receiving into and the room in front of the
  // receive region that the partitions' carried sections (open transactions) move into
  int recv_slab = -1;
  int64_t recv_carry = 0, recv_bytes = 0;
  std::atomic<int64_t> cpu_ns[2] = {{0}, {0}};  // thread CPU time spent in {receive copies, framing} since the group was created (surge_ingest_group_cpu_seconds)
  int64_t expand_hint = 1;  // the slice factor the last feed needed (host-side lz4: the topic's compression ratio does not change from feed to feed)
  FramingPool pool;
  ~surge_ingest_group() {
    for (surge_ingest* x : g) delete x;
  }
};

extern "C" {

int32_t surge_ingest_group_create(int32_t n_partitions, int32_t isolation_level, surge_ingest_group** out) {
  if (!out) return fail(nullptr, E_INVALID, "out is NULL");
  *out = nullptr;
  if (n_partitions < 1 || n_partitions > (1 << 20)) return fail(nullptr, E_INVALID, "n_partitions out of range");
  surge_ingest_group* grp = nullptr;
  try {
    grp = new surge_ingest_group();
    grp->g.reserve((size_t)n_partitions);
    grp->drained.resize((size_t)n_partitions);
    grp->off.resize((size_t)n_partitions + 1);
    grp->saved.resize((size_t)n_partitions);
    grp->status.resize((size_t)n_partitions);
    grp->consumed.resize((size_t)n_partitions);
    for (int32_t p = 0; p < n_partitions; ++p) {
      surge_ingest* x = nullptr;
      const int32_t rc = surge_ingest_create(isolation_level | SURGE_INGEST_FRAMES, &x);
      if (rc != OK) {
        delete grp;
        return rc;
      }
      x->grouped = true;
      grp->g.push_back(x);
    }
  } catch (const std::bad_alloc&) {
    delete grp;
    return fail(nullptr, E_NOMEM, "out of host memory");
  }
  *out = grp;
  return OK;
}

int32_t surge_ingest_group_destroy(surge_ingest_group* grp) {
  delete grp;
  return OK;
}

const char* surge_ingest_group_last_error(const surge_ingest_group* grp) { return grp ? grp->err.c_str() : g_err.c_str(); }

int32_t surge_ingest_group_set_allocator(surge_ingest_group* grp, void* (*alloc)(size_t), void (*release)(void*)) {
  if (!grp || !alloc != !release) return fail(nullptr, E_INVALID, "bad argument");
  for (Arena& a : grp->slabs) {
    if (a.cap) { grp->err = "surge_ingest_group_set_allocator after the first feed"; return -2; }
    a.alloc = alloc;
    a.release = release;
  }
  return OK;
}

int64_t surge_ingest_group_queued_sections(const surge_ingest_group* grp) {
  if (!grp) return 0;
  int64_t n = 0;
  for (const surge_ingest* x : grp->g) n += (int64_t)x->queue.size();
  return n;
}

static int64_t thread_cpu_ns() {
  struct timespec ts;
  if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) != 0) return 0;
  return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

int32_t surge_ingest_group_slab_bytes(const surge_ingest_group* grp, int64_t* bytes_out, int32_t* custom_allocator_out) {
  if (!grp || !bytes_out) return E_INVALID;
  int64_t total = 0;
  bool custom = false;
  for (const Arena& a : grp->slabs) {
    total += (int64_t)a.cap;
    custom = custom || a.alloc != nullptr;
  }
  *bytes_out = total;
  if (custom_allocator_out) *custom_allocator_out = custom ? 1 : 0;
  return OK;
}

int32_t surge_ingest_group_cpu_seconds(const surge_ingest_group* grp, double out[2]) {
  if (!grp || !out) return E_INVALID;
  out[0] = (double)grp->cpu_ns[0].load() * 1e-9;
  out[1] = (double)grp->cpu_ns[1].load() * 1e-9;
  return OK;
}

// what the partitions still hold (open transactions, batches not drained yet), 16-byte aligned per partition
static int64_t group_carry_bytes(const surge_ingest_group* grp) {
  int64_t total = 0;
  for (const surge_ingest* x : grp->g) {
    int64_t queued = 0;
    for (const Batch& qb : x->queue) queued += qb.sect_off >= 0 ? qb.sect_len : 0;
    total = (total + queued + 15) & ~15ll;
  }
  return total;
}

int32_t surge_ingest_group_receive_buffer(surge_ingest_group* grp, int64_t bytes, uint8_t** buf_out) {
  if (!grp || !buf_out || bytes < 0) return fail(nullptr, E_INVALID, "bad argument");
  *buf_out = nullptr;
  const int next = (grp->cur + 1) % surge_ingest::kArenas;
  Arena& slab = grp->slabs[next];
  const int64_t carry = (group_carry_bytes(grp) + 63) & ~63ll;
  try {
    slab.clear();
    const size_t want = (size_t)(carry + bytes) + 64;
    if (slab.cap < want) {
      const size_t roomy = want + want / 4 + 65536;  // (as surge_ingest_group_feed sizes them: every slab by the first response)
      bool first = true;
      for (const Arena& a : grp->slabs) first = first && a.cap == 0;
      if (first) {
        for (Arena& a : grp->slabs) a.reserve_exact(roomy);
      } else {
        slab.reserve_exact(roomy);
      }
    }
  } catch (const std::bad_alloc&) {
    grp->err = "out of host memory for the group's slab";
    return E_NOMEM;
  }
  grp->recv_slab = next;
  grp->recv_carry = carry;
  grp->recv_bytes = bytes;
  *buf_out = slab.data() + carry;
  return OK;
}

// For a host whose fetch responses lie elsewhere (a test harness, a consumer library that owns its buffers): the one copy a
// socket read would have made — every partition's bytes into the group's receive buffer, on `threads` threads (the calling
// thread + the group's pool), cut into pieces of 1 MiB — and where each partition's bytes now are.
namespace {
// A fetch response's bytes into the slab with non-temporal stores.  What is written is read again by the framer — one header line
// per few KB, prefetched a batch ahead — and by the copy engine; a cached copy reads every destination line first (read for
// ownership: a third more memory traffic) and evicts what the other threads work on.  Falls back to memcpy without AVX2.
__attribute__((target("avx2"))) void stream_copy_avx2(uint8_t* to, const uint8_t* from, size_t n) {
  size_t head = (size_t)(-(intptr_t)to) & 31u;
  if (head > n) head = n;
  std::memcpy(to, from, head);
  to += head; from += head; n -= head;
  size_t i = 0;
  for (; i + 128 <= n; i += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(from + i)), b = _mm256_loadu_si256((const __m256i*)(from + i + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i*)(from + i + 64)), d = _mm256_loadu_si256((const __m256i*)(from + i + 96));
    _mm256_stream_si256((__m256i*)(to + i), a);
    _mm256_stream_si256((__m256i*)(to + i + 32), b);
    _mm256_stream_si256((__m256i*)(to + i + 64), c);
    _mm256_stream_si256((__m256i*)(to + i + 96), d);
  }
  _mm_sfence();
  std::memcpy(to + i, from + i, n - i);
}
void stream_copy(uint8_t* to, const uint8_t* from, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2") && std::getenv("SURGE_INGEST_RECEIVE_COPY") == nullptr;  // (any value: plain memcpy, for A/B)
  if (avx2 && n >= 4096) stream_copy_avx2(to, from, n);
  else std::memcpy(to, from, n);
}
}  // namespace

int32_t surge_ingest_group_receive_copy(surge_ingest_group* grp, const uint8_t* const* data, const int64_t* len, int32_t threads, const uint8_t** placed_out) {
  if (!grp || !data || !len || !placed_out) return fail(nullptr, E_INVALID, "bad argument");
  const int32_t n = (int32_t)grp->g.size();
  int64_t total = 0;
  for (int32_t p = 0; p < n; ++p) {
    if (len[p] < 0 || (!data[p] && len[p] > 0)) { grp->err = "bad buffer for partition " + std::to_string(p); return E_INVALID; }
    total += len[p];
  }
  uint8_t* base = nullptr;
  const int32_t rc = surge_ingest_group_receive_buffer(grp, total, &base);
  if (rc != OK) return rc;
  struct Piece { uint8_t* to; const uint8_t* from; size_t n; };
  std::vector<Piece> pieces;
  try {
    int64_t at = 0;
    for (int32_t p = 0; p < n; ++p) {
      placed_out[p] = len[p] ? base + at : nullptr;
      for (int64_t o = 0; o < len[p]; o += 1 << 20) pieces.push_back(Piece{base + at + o, data[p] + o, (size_t)(len[p] - o < (1 << 20) ? len[p] - o : (1 << 20))});
      at += len[p];
    }
  } catch (const std::bad_alloc&) {
    grp->err = "out of host memory";
    return E_NOMEM;
  }
  std::atomic<size_t> next{0};
  const std::function<void()> work = [&]() {
    const int64_t t0 = thread_cpu_ns();
    for (;;) {
      const size_t k = next.fetch_add(1);
      if (k >= pieces.size()) break;
      stream_copy(pieces[k].to, pieces[k].from, pieces[k].n);
    }
    grp->cpu_ns[0] += thread_cpu_ns() - t0;
  };
  int32_t t = threads < 1 ? 1 : threads;
  if ((size_t)t > pieces.size()) t = (int32_t)(pieces.size() ? pieces.size() : 1);
  grp->pool.run(work, t - 1);
  return OK;
}

int32_t surge_ingest_group_feed(surge_ingest_group* grp, const uint8_t* const* data, const int64_t* len, int32_t threads, int64_t* consumed_out,
                                int64_t max_sections, surge_batch_section* sections_out, int64_t* n_sections_out, const uint8_t** slab_out) {
  if (!grp || !data || !len || !n_sections_out || !slab_out || max_sections < 0 || (!sections_out && max_sections > 0)) return fail(nullptr, E_INVALID, "bad argument");
  const int32_t n = (int32_t)grp->g.size();
  *n_sections_out = 0;
  *slab_out = nullptr;
  if (consumed_out)
    for (int32_t p = 0; p < n; ++p) consumed_out[p] = 0;
  for (int32_t p = 0; p < n; ++p)
    if (len[p] < 0 || (!data[p] && len[p] > 0)) { grp->err = "bad buffer for partition " + std::to_string(p); return E_INVALID; }
  const int group_cur = grp->cur;
  // In place: every partition's bytes lie inside the region surge_ingest_group_receive_buffer handed out for this feed.  The
  // sections then stay where they were received — the slab is not written at all beyond the few carried sections in front.
  bool inplace = grp->recv_slab == (group_cur + 1) % surge_ingest::kArenas;
  if (inplace) {
    const uint8_t* lo = grp->slabs[grp->recv_slab].data() + grp->recv_carry;
    const uint8_t* hi = lo + grp->recv_bytes;
    bool any = false;
    for (int32_t p = 0; p < n; ++p) {
      if (len[p] == 0) continue;
      any = true;
      if (data[p] < lo || data[p] + len[p] > hi) inplace = false;
    }
    inplace = inplace && any;
  }
  grp->recv_slab = -1;
  auto undo = [&]() {
    for (int32_t p = 0; p < n; ++p) {
      surge_ingest* x = grp->g[(size_t)p];
      MemberSave& sv = grp->saved[(size_t)p];
      x->queue.swap(sv.queue);
      std::memcpy(x->counters, sv.counters, sizeof sv.counters);
      x->cur = sv.cur;
      x->handed_out = sv.handed_out;
      grp->drained[(size_t)p].clear();
    }
    grp->cur = group_cur;
    if (inplace) grp->recv_slab = (group_cur + 1) % surge_ingest::kArenas;  // (the received bytes are still there: the caller may feed them again)
    if (consumed_out)
      for (int32_t p = 0; p < n; ++p) consumed_out[p] = 0;
  };
  // Every partition's slice: what it still holds (open transactions, batches not drained yet) + this feed.  With
  // SURGE_INGEST_DEVICE_LZ4 a section is never longer than its batch; without it the host decompresses lz4 batches INTO the
  // slice, whose size is then only known afterwards: the feed is undone and run again with `expand` times the room.
  // (the trial starts from the factor the last feed needed, not from 1: a group without SURGE_INGEST_DEVICE_LZ4 on a compressed
  // topic otherwise framed — CRC, decompression and all — every fetch two or three times over, for ever; the factor is capped
  // at 256: an LZ4 block expands at most 255 x)
  for (int64_t expand = grp->expand_hint;; expand *= 4) {
    // what a failed feed is undone from
    try {
      for (int32_t p = 0; p < n; ++p) {
        const surge_ingest* x = grp->g[(size_t)p];
        MemberSave& sv = grp->saved[(size_t)p];
        sv.queue = x->queue;
        std::memcpy(sv.counters, x->counters, sizeof sv.counters);
        sv.cur = x->cur;
        sv.handed_out = x->handed_out;
      }
    } catch (const std::bad_alloc&) {
      grp->err = "out of host memory";
      return E_NOMEM;
    }
    int64_t total = 0;
    for (int32_t p = 0; p < n; ++p) {
      int64_t queued = 0;
      for (const Batch& qb : grp->g[(size_t)p]->queue) queued += qb.sect_off >= 0 ? qb.sect_len : 0;
      grp->off[(size_t)p] = total;
      total = (total + queued + (inplace ? 0 : len[p] * expand) + 15) & ~15ll;
    }
    grp->off[(size_t)n] = total;
    Arena& slab = grp->slabs[(group_cur + 1) % surge_ingest::kArenas];
    try {
      if (inplace) {
        if (total > grp->recv_carry) {  // (cannot happen: nothing was fed between receive_buffer and this call)
          grp->err = "the partitions' carried sections outgrew the room surge_ingest_group_receive_buffer left for them";
          return E_INVALID;
        }
      } else {
        slab.clear();
      }
      if (!inplace && slab.cap < (size_t)total + 16) {
        // Page-locking a 30 - 45 MB slab takes 1.5 ms on a good day and 20 ms on a bad one (bytes -> states on the small-flush
        // topic: two of a dozen runs lost a 24 ms fetch to it and with it two thirds of their rate).  So the group's FIRST feed
        // sizes ALL its slabs, with a quarter to spare — fetch responses of one consumer are alike — and a recovery pays for
        // page-locking once, before its pipeline fills, not once per slab over its first six fetches.
        const size_t roomy = (size_t)total + (size_t)total / 4 + 65536;
        bool first = true;
        for (const Arena& a : grp->slabs) first = first && a.cap == 0;
        if (first) {
          for (Arena& a : grp->slabs) a.reserve_exact(roomy);
        } else {
          slab.reserve_exact(roomy);
        }
      }
    } catch (const std::bad_alloc&) {
      grp->err = "out of host memory for the group's slab";
      return E_NOMEM;
    }
    grp->cur = (group_cur + 1) % surge_ingest::kArenas;
    std::atomic<int32_t> next{0};
    // A thread takes the partitions eight at a time and first walks their batch LENGTHS side by side — one step per partition
    // in turn, the next header of each prefetched while the other seven take theirs: a batch header is a cache line nobody has
    // touched since the response was received (with non-temporal stores: not in any cache), and a framer that meets them one
    // after the other waits out a memory latency per batch — most of what a batch costs it.
    constexpr int32_t kTogether = 8;
    auto touch_headers = [&](int32_t p0, int32_t p1) {
      const uint8_t* d[kTogether];
      int64_t ln[kTogether], at[kTogether];
      int32_t live = 0;
      for (int32_t p = p0; p < p1; ++p) {
        d[p - p0] = data[p];
        ln[p - p0] = len[p];
        at[p - p0] = 0;
        if (len[p] >= 12) { __builtin_prefetch(data[p]); ++live; } else at[p - p0] = -1;
      }
      while (live > 0) {
        for (int32_t i = 0; i < p1 - p0; ++i) {
          if (at[i] < 0) continue;
          const uint8_t* q = d[i] + at[i];
          const int64_t bl = ((int64_t)q[8] << 24) | ((int64_t)q[9] << 16) | ((int64_t)q[10] << 8) | (int64_t)q[11];
          const int64_t nxt = at[i] + 12 + bl;
          if (bl < 49 || nxt + 12 > ln[i]) { at[i] = -1; --live; continue; }  // (the real walk decides what a bad length means)
          __builtin_prefetch(d[i] + nxt);
          __builtin_prefetch(d[i] + nxt + 60);
          at[i] = nxt;
        }
      }
    };
    const std::function<void()> work = [&]() {
      const int64_t t0 = thread_cpu_ns();
      for (;;) {
        const int32_t p0 = next.fetch_add(kTogether);
        if (p0 >= n) break;
        const int32_t p1 = p0 + kTogether < n ? p0 + kTogether : n;
        touch_headers(p0, p1);
        for (int32_t p = p0; p < p1; ++p) {
        surge_ingest* x = grp->g[(size_t)p];
        int32_t rc = OK;
        int64_t consumed = 0;
        try {  // (nothing may leave a thread)
          x->ext_next = slab.data() + grp->off[(size_t)p];
          x->ext_next_cap = (size_t)(grp->off[(size_t)p + 1] - grp->off[(size_t)p]);
          x->slice_overflow = false;
          x->inplace = inplace;
          rc = surge_ingest_feed(x, data[p], len[p], &consumed);
          x->inplace = false;
          std::vector<surge_batch_section>& out = grp->drained[(size_t)p];
          out.clear();
          if (rc == OK && !x->queue.empty()) {
            out.resize(x->queue.size());
            int64_t got = 0;
            rc = surge_ingest_drain_sections(x, (int64_t)out.size(), out.data(), &got);
            out.resize(rc == OK ? (size_t)got : 0);
            for (surge_batch_section& sct : out) sct.byte_off += grp->off[(size_t)p];
          }
        } catch (...) {
          rc = E_NOMEM;
        }
        grp->status[(size_t)p] = rc;
        grp->consumed[(size_t)p] = consumed;
        }
      }
      grp->cpu_ns[1] += thread_cpu_ns() - t0;
    };
    int32_t t = threads < 1 ? 1 : threads;
    if (t > n) t = n;
    grp->pool.run(work, t - 1);
    int32_t first_bad = OK;
    bool overflow = false;
    for (int32_t p = 0; p < n; ++p) {
      overflow |= grp->g[(size_t)p]->slice_overflow;
      if (grp->status[(size_t)p] != OK && first_bad == OK) {
        first_bad = grp->status[(size_t)p];
        grp->err = "partition " + std::to_string(p) + ": " + grp->g[(size_t)p]->err;
      }
    }
    if (overflow && expand < 256) {
      undo();
      continue;
    }
    if (first_bad == OK) grp->expand_hint = expand;
    if (first_bad != OK) {
      undo();
      return first_bad;
    }
    int64_t need = 0;
    for (int32_t p = 0; p < n; ++p) need += (int64_t)grp->drained[(size_t)p].size();
    if (need > max_sections) {
      undo();
      *n_sections_out = need;
      grp->err = "sections_out is too small: this feed delivers " + std::to_string(need) + " sections (n_sections_out; the feed was undone — call again with room for them)";
      return E_INVALID;
    }
    int64_t at = 0;
    for (int32_t p = 0; p < n; ++p) {
      for (const surge_batch_section& sct : grp->drained[(size_t)p]) sections_out[at++] = sct;
      if (consumed_out) consumed_out[p] = grp->consumed[(size_t)p];
    }
    *n_sections_out = at;
    *slab_out = slab.data();
    return OK;
  }
}

int32_t surge_ingest_group_counters(const surge_ingest_group* grp, int64_t out[8]) {
  if (!grp || !out) return E_INVALID;
  for (int i = 0; i < 8; ++i) out[i] = 0;
  for (const surge_ingest* x : grp->g)
    for (int i = 0; i < 8; ++i) out[i] += x->counters[i];
  return OK;
}

int64_t surge_ingest_key_count(const surge_ingest* g) { return g ? (int64_t)g->keys.size() : 0; }

int32_t surge_ingest_key(const surge_ingest* g, int64_t idx, const char** utf8_out, int64_t* len_out) {
  if (!g || !utf8_out || !len_out) return fail(nullptr, E_INVALID, "bad argument");
  if (idx < 0 || idx >= (int64_t)g->keys.size()) return E_INVALID;
  *utf8_out = g->keys[(size_t)idx].data();
  *len_out = (int64_t)g->keys[(size_t)idx].size();
  return OK;
}

int32_t surge_ingest_counters(const surge_ingest* g, int64_t out[8]) {
  if (!g || !out) return E_INVALID;
  for (int i = 0; i < 8; ++i) out[i] = g->counters[i];
  return OK;
}

}  // extern "C"
