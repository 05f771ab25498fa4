// lz4_frame.cpp — the WRITING half of the LZ4 frame format (the reading half lives in ingest.cpp) and XXH32, restated
// from the published specifications (lz4_Frame_format.md, lz4_Block_format.md, xxhash_spec.md).  Host side of
// libsurge_replay.so: the state-topic snapshot writer compresses its record batches the way the reference's producer
// does (compression.type = lz4, modules/common/src/main/resources/reference.conf:112; kafka-clients frames a
// compressed batch's records as ONE LZ4 frame: version 01, block independence, 64 KiB blocks, no block / content
// checksum, header checksum = second byte of XXH32(descriptor) — KafkaLZ4BlockOutputStream).
//
// The block compressor is the textbook greedy one: a 4-byte hash table of the most recent position, matches extended
// forward, no lazy evaluation — the point is a correct, reasonably fast frame any LZ4 decoder accepts (the tests
// decode it with liblz4 as bundled by Apache Arrow), not the best ratio.
#include <cstdint>
#include <cstring>

#include "../../include/surge_ingest.h"

namespace {

constexpr uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
inline uint32_t rd32(const uint8_t* p) {
  uint32_t v;
  std::memcpy(&v, p, 4);
  return v;  // little-endian hosts only (x86-64)
}

constexpr int kMinMatch = 4, kMfLimit = 12, kLastLiterals = 5, kHashLog = 14;
constexpr int64_t kBlock = 65536;

inline void put_len(uint8_t*& op, size_t n) {  // the 255-continued length that follows a saturated token nibble
  while (n >= 255) { *op++ = 255; n -= 255; }
  *op++ = (uint8_t)n;
}

// one block; returns the compressed size (dst must hold n + n/255 + 16 bytes)
int64_t compress_block(const uint8_t* src, int64_t n, uint8_t* dst) {
  uint8_t* op = dst;
  const uint8_t* ip = src;
  const uint8_t* anchor = src;
  const uint8_t* const iend = src + n;
  if (n >= kMfLimit + 1) {
    const uint8_t* const mflimit = iend - kMfLimit;   // a match may not START after this ...
    const uint8_t* const matchlimit = iend - kLastLiterals;  // ... nor extend beyond this
    int32_t table[1 << kHashLog];
    for (int32_t& t : table) t = -1;
    while (ip <= mflimit) {
      const uint32_t h = (rd32(ip) * 2654435761u) >> (32 - kHashLog);
      const int32_t cand = table[h];
      table[h] = (int32_t)(ip - src);
      if (cand < 0 || (ip - src) - cand > 65535 || rd32(src + cand) != rd32(ip)) {
        ++ip;
        continue;
      }
      const uint8_t* match = src + cand;
      const uint8_t* m = match + kMinMatch;
      const uint8_t* q = ip + kMinMatch;
      while (q < matchlimit && *q == *m) { ++q; ++m; }
      const size_t lit = (size_t)(ip - anchor), ml = (size_t)(q - ip) - kMinMatch;
      uint8_t* token = op++;
      *token = (uint8_t)((lit >= 15 ? 15 : lit) << 4 | (ml >= 15 ? 15 : ml));
      if (lit >= 15) put_len(op, lit - 15);
      std::memcpy(op, anchor, lit);
      op += lit;
      const uint16_t off = (uint16_t)(ip - match);
      *op++ = (uint8_t)off;
      *op++ = (uint8_t)(off >> 8);
      if (ml >= 15) put_len(op, ml - 15);
      ip = q;
      anchor = ip;
    }
  }
  const size_t lit = (size_t)(iend - anchor);  // the last sequence is literals only
  uint8_t* token = op++;
  *token = (uint8_t)((lit >= 15 ? 15 : lit) << 4);
  if (lit >= 15) put_len(op, lit - 15);
  std::memcpy(op, anchor, lit);
  op += lit;
  return op - dst;
}

}  // namespace

extern "C" {

uint32_t surge_xxh32(const uint8_t* p, int64_t len, uint32_t seed) {
  const uint8_t* const end = p + len;
  uint32_t h;
  if (len >= 16) {
    uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    const uint8_t* const limit = end - 16;
    do {
      v1 = rotl(v1 + rd32(p) * P2, 13) * P1;
      v2 = rotl(v2 + rd32(p + 4) * P2, 13) * P1;
      v3 = rotl(v3 + rd32(p + 8) * P2, 13) * P1;
      v4 = rotl(v4 + rd32(p + 12) * P2, 13) * P1;
      p += 16;
    } while (p <= limit);
    h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
  } else {
    h = seed + P5;
  }
  h += (uint32_t)len;
  while (p + 4 <= end) {
    h = rotl(h + rd32(p) * P3, 17) * P4;
    p += 4;
  }
  while (p < end) {
    h = rotl(h + *p * P5, 11) * P1;
    ++p;
  }
  h ^= h >> 15;
  h *= P2;
  h ^= h >> 13;
  h *= P3;
  h ^= h >> 16;
  return h;
}

int64_t surge_lz4_frame_bound(int64_t n) {
  if (n < 0) return -1;
  const int64_t blocks = (n + kBlock - 1) / kBlock;
  return 7 + blocks * 4 + n + 4;  // header + per-block size words + (worst case: every block stored) + EndMark
}

int64_t surge_lz4_frame_compress(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap) {
  if (n < 0 || (!src && n > 0) || !dst) return -1;
  if (cap < surge_lz4_frame_bound(n)) return -6;
  uint8_t* op = dst;
  const uint8_t magic[4] = {0x04, 0x22, 0x4D, 0x18};
  std::memcpy(op, magic, 4);
  op += 4;
  op[0] = 0x60;  // FLG: version 01, block independence
  op[1] = 0x40;  // BD: 64 KiB blocks
  op[2] = (uint8_t)(surge_xxh32(op, 2, 0) >> 8);
  op += 3;
  uint8_t scratch[kBlock + kBlock / 255 + 32];
  for (int64_t s = 0; s < n; s += kBlock) {
    const int64_t len = n - s < kBlock ? n - s : kBlock;
    const int64_t c = compress_block(src + s, len, scratch);
    const bool stored = c >= len;  // incompressible: ship the bytes as they are (high bit of the size word)
    const uint32_t word = stored ? ((uint32_t)len | 0x80000000u) : (uint32_t)c;
    op[0] = (uint8_t)word; op[1] = (uint8_t)(word >> 8); op[2] = (uint8_t)(word >> 16); op[3] = (uint8_t)(word >> 24);
    op += 4;
    std::memcpy(op, stored ? src + s : scratch, (size_t)(stored ? len : c));
    op += stored ? len : c;
  }
  std::memset(op, 0, 4);  // EndMark
  op += 4;
  return op - dst;
}

}  // extern "C"
