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

#include <sys/prctl.h>
#include <time.h>

#include <atomic>
#include <chrono>
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
  unsigned int reserved, n_f64_host;
  unsigned int lz4_bad;          // the first section whose LZ4 frame did not decode (~0 = none)
  unsigned int crc_bad;          // the first section whose bytes do not give the batch's CRC-32C (~0 = none; SURGE_INGEST_DEVICE_CRC)
};

__device__ __forceinline__ void report(ErrorCell* err, int64_t rec, uint32_t status) {
  atomicMin(&err->first_bad, ((unsigned long long)rec << 8) | status);
}

// ---- CRC-32C of a batch, finished on the device (SURGE_INGEST_DEVICE_CRC) ----------------------------------------------------
// A record batch's CRC-32C (Castagnoli, reflected; kafka-clients: Crc32C over attributes .. end of the batch) covers 40 header
// bytes and then the records section — the bytes this decoder is handed anyway.  The host framer runs the CRC over the 40
// header bytes only and passes on the register (in-place framing: not even those); one WAVE per section takes it from there,
// 4 KiB at a time (a compressed batch of the reference's publisher is 2 - 5 KB: one or two tiles):
//   * the tile is laid right-aligned into a 4 KiB frame, lane l owns frame bytes [64 l, 64 l + 64) (the lanes in front of a
//     short tile are empty: a zero register is neutral under what follows) and runs the CRC over its piece a dword at a time
//     out of LDS (pieces of 16 dwords, 17 apart: lanes read different banks), four table look-ups per dword (slicing by four:
//     4 x 256 entries in LDS, loaded once per workgroup of four waves);
//   * a CRC register is linear in (register, data): crc(A || B) = shift(crc(A), |B|) ^ crc_0(B), and shifting by a FIXED
//     length is one multiplication mod P by a constant x^(8 |B|): lane l multiplies its register by x^(8 * 64 (63 - l)) — ONE
//     multiplication per lane, all lanes at once — and the tile's register is the XOR over the wave; one more multiplication
//     (by x^(8 * 4096), wave-uniform) chains a tile to the ones before it.  65 constants, computed once on the host.
// History: bit-serial CRC, 16 KiB tiles, 182 us per 10^6-record fetch (profiles/r06_e2e_inplace_kernel_stats.csv); slicing by
// four with 256-byte pieces and a six-level lane tree of multiplications, 122 us alone / 244 us on the wider topic's 66 MB
// (r06_e2e_*_depth1_kernel_stats.csv): 64 dependent look-up rounds and seven serial 32-step multiplications per tile, most
// lanes of a 4 KB section idle.  A mismatch is reported like a bad LZ4 frame: the push fails with SURGE_E_CORRUPT, nothing of
// it is delivered, no key it brought stays interned.
struct CrcSpan {
  int64_t off;      // first byte the device still has to run the CRC over, in the staged bytes
  int32_t len;
  int32_t section;
  uint32_t expect;  // the batch's CRC-32C
  uint32_t state;   // the CRC register in front of `off`: after the 40 header bytes (the host ran those), or ~0 (in-place framing:
                    // off points at the header bytes, the device runs everything)
};
constexpr uint32_t kCrcPoly = 0x82F63B78u;
constexpr int kCrcTile = 4096, kCrcPiece = 64, kCrcWaves = 4;
constexpr int kCrcLdsDwords = 65 * 17;  // 64 pieces + the dword a misaligned tile spills into, every piece padded by one dword

// a * b mod P, reflected representation (bit 31 = x^0): 32 fixed steps (zlib's multmodp stops early on a's last set bit —
// and never on a == 0, which an empty lane's register is)
__host__ __device__ inline uint32_t crc_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0u;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    p ^= b & (uint32_t)-(int32_t)((a >> (31 - i)) & 1u);
    b = (b >> 1) ^ (kCrcPoly & (uint32_t)-(int32_t)(b & 1u));
  }
  return p;
}

// x^(8 n) mod P
static uint32_t crc_x8n(uint64_t n) {
  uint32_t sq = 1u << 30, p = 1u << 31;  // x^1, x^0
  for (uint64_t bits = n * 8; bits; bits >>= 1) {
    if (bits & 1u) p = crc_mulmod(sq, p);
    sq = crc_mulmod(sq, sq);
  }
  return p;
}

// What the kernel reads from device memory: the slicing-by-four tables (T[k][b] = the register after byte b followed by k
// zero bytes), lane l's shift x^(8 * 64 (63 - l)), the tile's shift x^(8 * 4096).
struct CrcTables {
  uint32_t slice[4][256];
  uint32_t lane_shift[64];
  uint32_t tile_shift, pad[3];
};
static const CrcTables& crc_tables() {
  static const CrcTables T = [] {
    CrcTables x{};
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
      x.slice[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int k = 1; k < 4; ++k) x.slice[k][i] = (x.slice[k - 1][i] >> 8) ^ x.slice[0][x.slice[k - 1][i] & 0xffu];
    for (int l = 0; l < 64; ++l) x.lane_shift[l] = crc_x8n((uint64_t)kCrcPiece * (uint64_t)(63 - l));
    x.tile_shift = crc_x8n((uint64_t)kCrcTile);
    return x;
  }();
  return T;
}

__global__ void __launch_bounds__(64 * kCrcWaves) crc_kernel(const uint8_t* __restrict__ bytes, const CrcSpan* __restrict__ spans, int32_t n_spans,
                                                             const CrcTables* __restrict__ tables, ErrorCell* err) {
  __shared__ uint32_t frames[kCrcWaves][kCrcLdsDwords];
  __shared__ uint32_t Ts[4 * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 256; i += 64 * kCrcWaves) Ts[i] = (&tables->slice[0][0])[i];
  __syncthreads();
  const int32_t span = (int32_t)blockIdx.x * kCrcWaves + wave;
  if (span >= n_spans) return;  // (behind the workgroup's only barrier)
  uint32_t* A = frames[wave];
  const uint32_t lane_k = tables->lane_shift[lane], tile_k = tables->tile_shift;
  const CrcSpan sp = spans[span];
  const uint32_t expect = sp.expect, state = sp.state;
  uint32_t total = state;  // (a section of no bytes: the register as the host left it)
  int64_t done = 0;
  bool first = true;
  while (done < sp.len) {
    int32_t T = (int32_t)((sp.len - done) % kCrcTile);
    if (T == 0) T = kCrcTile;
    // frame byte p of this tile = global byte base + p, valid for p >= v0
    const int32_t v0 = kCrcTile - T;
    const int64_t base = sp.off + done - v0;   // (may lie in front of the staged bytes: only p >= v0 is ever read)
    const int32_t sh = (int32_t)(base & 3);
    const int64_t abase = base - sh;           // the aligned stream A[j] = dword at abase + 4 j, j in [0, 1024]
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k <= kCrcTile / 256; ++k) {
      const int j = lane + 64 * k;
      if (j > kCrcTile / 4) continue;  // (the 1025th dword: lane 0 only)
      const int64_t g = abase + 4ll * j;
      uint32_t w = 0u;
      if (g + 4 > base + v0 && g < base + kCrcTile) {  // overlaps the tile
        if (g >= sp.off - 4 && g + 4 <= sp.off + sp.len + 64) w = *(const uint32_t*)(bytes + g);  // inside what was staged (at least 4 bytes of prefix in front, 64 spare bytes behind)
        else
          for (int b = 0; b < 4; ++b)
            if (g + b >= sp.off && g + b < sp.off + sp.len) w |= (uint32_t)bytes[g + b] << (8 * b);
      }
      A[(j >> 4) * 17 + (j & 15)] = w;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // my 64 bytes: frame [64 lane, 64 lane + 64)
    const int32_t p0 = kCrcPiece * lane, p1 = p0 + kCrcPiece;
    uint32_t r = 0u;
    if (p1 > v0) {
      int32_t p = p0 > v0 ? p0 : v0;
      const bool holds_first = first && p0 <= v0;  // the section's very first byte is mine: the host's register goes in here
      if (holds_first) r = state;
      auto dword_at = [&](int32_t q) -> uint32_t {  // frame dword q (frame bytes [4 q, 4 q + 4))
        const uint32_t lo = A[(q >> 4) * 17 + (q & 15)], hi = A[((q + 1) >> 4) * 17 + ((q + 1) & 15)];
        return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)sh);
      };
      // the bytes in front of my first whole dword (a tile that starts inside one)
      while ((p & 3) && p < p1) {
        const uint32_t w = dword_at(p >> 2);
        r ^= (w >> (8 * (p & 3))) & 0xffu;
        r = (r >> 8) ^ Ts[r & 0xffu];
        ++p;
      }
      for (; p + 4 <= p1; p += 4) {
        const uint32_t x = r ^ dword_at(p >> 2);
        r = Ts[768 + (x & 0xffu)] ^ Ts[512 + ((x >> 8) & 0xffu)] ^ Ts[256 + ((x >> 16) & 0xffu)] ^ Ts[x >> 24];
      }
    }
    // every lane shifts its register over the pieces behind it; the tile's register is the XOR of them all
    r = crc_mulmod(r, lane_k);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) r ^= (uint32_t)__shfl_xor((int)r, s, 64);
    total = first ? r : crc_mulmod(total, tile_k) ^ r;
    first = false;
    done += T;
  }
  if (lane == 0 && ~total != expect) atomicMin(&err->crc_bad, (unsigned int)sp.section);
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
  int64_t seq_off;   // first entry of this block in the sequence table (two-pass decode); -1: the one-pass kernel takes it
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
    if (blk.seq_off >= 0) continue;  // decoded in two passes (lz4_parse_kernel / lz4_exec_kernel)
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

// ---- LZ4 in two passes (round 4) ---------------------------------------------------------------------------------------
// What the counters said about the one-pass kernel above (profiles/r04_e2e_r3sources_summary.txt: 1 M records = 7142
// blocks of ~16 KiB, ~1050 sequences each — play-json text compresses into sequences of 1.4 literal and 13.5 match bytes):
// 108 k SCALAR instructions per block — the sequence headers are wave-uniform, so every length / offset / bounds step is
// SALU work that 63 of 64 lanes only wait for — against 20 k VALU and 3 k LDS instructions; 2.4 ms per fetch, bound by the
// one scalar pipe of a CU.  Two things are serial in an LZ4 block — finding where the next sequence starts, and a match
// reading what an earlier match wrote — and neither needs a wave per sequence:
//   pass 1  lz4_parse_kernel   ONE LANE per block (64 blocks per wave) walks its block's sequence headers: a 16-byte
//           unaligned load at the sequence's first byte holds the token, up to 12 literal bytes, the offset and one length
//           byte — the whole header of all but a fraction of a percent of the sequences — so a step is one load and ~50
//           branch-free VALU instructions for 64 blocks at once.  It writes an 8-byte entry per sequence {position,
//           literal length, match length, offset}, stores the literal bytes where they belong in the decompressed area
//           (they come from the compressed stream, never from earlier output), and validates what the one-pass decoder
//           validates, so the block's exact decompressed size — the section's length and the LDS class of pass 2 — is
//           known before a match is copied.
//   pass 2  lz4_exec_kernel    one WAVE per block, in a launch with exactly the LDS its size needs: the block's image
//           (literals in place) is loaded into LDS; per 64 entries the lanes expand their matches into a byte map
//           (map[p] = the position byte p copies from; identity for literals), then the map is applied 64 output bytes at
//           a time, lane per byte: one gather when no byte of the window reads another byte of the same window, else once
//           more per byte that does (a chain inside a window cannot be longer).  ~5 instructions per sequence instead of
//           ~140, and the serial chain is per 64 bytes of output instead of per sequence.  Overlapping matches (offset <
//           length: the period IS the data), matches above 256 bytes and groups that span more than the map holds take
//           the sequence-by-sequence loop.
// Blocks whose compressed size is 64 KiB or more (a compressor that expands instead of storing) keep the one-pass kernel.
struct Lz4Work {
  int32_t dbg = 0;     // SURGE_DBG_DECODE (timing experiments only): 1 = image in, image out; 2 = the byte maps are built, never applied
  int32_t pad = 0;     // bytes between image and map (SURGE_INGEST_LZ4_PAD=64: round 4's layout, for same-box comparisons)
  int32_t* state;      // per block: >= 0 decoded (its size), -1 malformed
  int32_t* n_seq;      // per block: entries written
  uint2* seq;          // the sequence table
  int32_t* cls_count;  // [kLz4Classes + 1]: blocks per LDS class; the last is "stored" (no LDS: copied as they are)
  int32_t* cls_list;   // [kLz4Classes + 1][n_blocks]
};
constexpr int kLz4Classes = 6;
constexpr int kLz4Map = 2048;  // bytes of output one group's map may span (a power of two; 64 sequences of play-json text span ~1 KB)
__device__ __constant__ int32_t kLz4ClassCap[kLz4Classes] = {8192, 16384, 24576, 32768, 49152, 65536};
static const int32_t kLz4ClassCapHost[kLz4Classes] = {8192, 16384, 24576, 32768, 49152, 65536};

typedef const __attribute__((address_space(3))) uint8_t* lds_ptr_t;

struct __attribute__((packed)) Unaligned16 { uint64_t lo, hi; };

// bits [b, b + 64) of the 128-bit little-endian window hi:lo, b in [0, 128)
__device__ __forceinline__ uint64_t window_bits(uint64_t lo, uint64_t hi, int b) {
  const uint64_t low = (lo >> (b & 63)) | ((hi << 1) << (63 - (b & 63)));
  return b < 64 ? low : hi >> (b & 63);
}

__device__ __forceinline__ int32_t wave_inclusive_sum(int32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int32_t u = __shfl_up(v, d, 64);
    if (lane >= d) v += u;
  }
  return v;
}

// One sequence whose header does not fit the 16-byte window (literal runs above 12 bytes, lengths continued over several
// bytes) or that ends the block: its header byte by byte out of LDS — every lane reads the same bytes, the values are
// wave-uniform — and its literal run copied by the whole wave.  false = malformed.
__device__ bool lz4_slow_sequence(lds_ptr_t in, int32_t n_in, uint8_t* __restrict__ dst, int32_t& ip, int32_t& op, uint2* __restrict__ out, int32_t& ns,
                                  int lane) {
  const uint32_t token = in[ip++];
  int32_t lit = (int32_t)(token >> 4);
  if (lit == 15) {
    uint32_t x;
    do {
      if (ip >= n_in) return false;
      x = in[ip++];
      lit += (int32_t)x;
    } while (x == 255u && lit < (1 << 24));
  }
  if (lit > n_in - ip || lit > kLz4BlockMax - op) return false;  // (n_in < 64 KiB: a length always fits 16 bits)
  for (int32_t j = lane; j < lit; j += 64) dst[op + j] = in[ip + j];
  ip += lit;
  if (ip >= n_in) {  // the last sequence carries literals only
    // An EMPTY last sequence (token 0x00 behind the last match: encoders other than liblz4 write one) gets no entry: it has
    // nothing to copy, and when the block is full its position, 65536, does not fit the entry's 16 bits (it would read as
    // "op 0, one literal" and push the last group's matches out of lz4_exec_kernel's range — ADVICE r4).
    if (lit > 0) {
      if (lane == 0) out[ns] = make_uint2((uint32_t)op | ((uint32_t)lit << 16), 0u);
      ++ns;
    }
    op += lit;
    return true;
  }
  if (n_in - ip < 2) return false;
  const int32_t offset = (int32_t)in[ip] | ((int32_t)in[ip + 1] << 8);
  ip += 2;
  int32_t ml = (int32_t)(token & 15u);
  if (ml == 15) {
    uint32_t x;
    do {
      if (ip >= n_in) return false;
      x = in[ip++];
      ml += (int32_t)x;
    } while (x == 255u && ml < (1 << 24));
  }
  ml += 4;
  const int32_t D = op + lit;
  // offset <= D and offset >= 1 make D >= 1, so a match that fits the block is at most 65535 long
  if (offset == 0 || offset > D || ml > kLz4BlockMax - D) return false;
  if (lane == 0) out[ns] = make_uint2((uint32_t)op | ((uint32_t)lit << 16), (uint32_t)ml | ((uint32_t)offset << 16));
  ++ns;
  op = D + ml;
  return true;
}

// Pass 1, one WAVE per block.  Where a sequence starts is only known once the one before it is decoded — but decoding a
// header is cheap, so every lane decodes the header that WOULD start at its byte of the current 64-byte window of the
// compressed block (staged in LDS), and the wave then follows the chain of real starts through those 64 candidates with
// v_readlane (a play-json block has a sequence every ~5 bytes: a dozen real ones per window).  The real lanes get their
// output positions from a wave prefix sum, write their table entries side by side and store their literal bytes.  (The
// first version gave every block ONE LANE that walked its headers alone: ~150 dependent instructions per sequence on a
// wave that has the SIMD to itself — 0.9 ms per 1 M records however the loads were arranged.)  A wave takes its block
// when lo_excl < staged bytes <= cap (two launches: 8 KiB of LDS, 20 waves per CU, for the batches a 16 KiB producer
// writes; 64 KiB for the rest).
__global__ void __launch_bounds__(64) lz4_parse_kernel(const uint8_t* __restrict__ bytes, uint8_t* __restrict__ out_base, const Lz4Block* __restrict__ blocks,
                                                       int32_t n_blocks, Lz4Work w, Section* __restrict__ sections, ErrorCell* err, int32_t lo_excl, int32_t cap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lz4_in[];
  const int lane = threadIdx.x;
  const int32_t b = (int32_t)blockIdx.x;
  const Lz4Block blk = blocks[b];
  if (blk.seq_off < 0) return;  // the one-pass kernel's
  const int32_t n_in = blk.src_len & 0x7fffffff;
  const bool stored = blk.src_len < 0;
  const int32_t skew = (int32_t)(blk.src_off & 15);
  const int32_t need_lds = stored ? 0 : skew + n_in + 48;
  if (!(need_lds > lo_excl && need_lds <= cap)) return;
  uint8_t* __restrict__ dst = out_base + blk.dst_off;
  int32_t op = 0, ns = 0;
  bool ok = true;
  int32_t cls = kLz4Classes;  // stored
  if (stored) {
    op = n_in;
    ok = n_in <= kLz4BlockMax;
  } else {
    {  // the block's bytes, from the 16-byte line its first byte lies in (the staged bytes end 64 bytes after their last byte)
      const uint4* src = (const uint4*)(bytes + blk.src_off - skew);
      const int n16 = (skew + n_in + 47) >> 4;
      for (int i = lane; i < n16; i += 64) ((uint4*)lz4_in)[i] = src[i];
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const lds_ptr_t in = (lds_ptr_t)lz4_in + skew;
    const uint32_t* in32 = (const uint32_t*)lz4_in;
    uint2* __restrict__ out = w.seq + blk.seq_off;
    int32_t base = 0, t0 = 0;
    while (ok && base + t0 < n_in) {
      const int32_t q = base + lane;
      // the 16 bytes from this lane's position on: five aligned dwords, shifted into place
      const uint32_t A = (uint32_t)(skew + q);
      const uint32_t* d = in32 + (A >> 2);
      const uint32_t d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
      const uint32_t shb = A & 3u;
      const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, shb), w1 = __builtin_amdgcn_alignbyte(d2, d1, shb);
      const uint32_t w2 = __builtin_amdgcn_alignbyte(d3, d2, shb), w3 = __builtin_amdgcn_alignbyte(d4, d3, shb);
      const uint64_t lo = (uint64_t)w0 | ((uint64_t)w1 << 32), hi = (uint64_t)w2 | ((uint64_t)w3 << 32);
      const uint32_t token = w0 & 0xffu;
      const int32_t lit = (int32_t)(token >> 4), mlc = (int32_t)(token & 15u);
      const int32_t need = 3 + lit + (mlc == 15 ? 1 : 0);
      const uint32_t tail = (uint32_t)window_bits(lo, hi, 8 * (1 + (lit & 15)));  // offset (16 bits), then the length byte
      // (the block's last, literal-only sequence has no room for an offset: not simple)
      const bool simple = q < n_in && lit <= 12 && need <= n_in - q && !(mlc == 15 && ((tail >> 16) & 0xffu) == 255u);
      const int32_t offset = (int32_t)(tail & 0xffffu);
      const int32_t ml = mlc + 4 + (mlc == 15 ? (int32_t)((tail >> 16) & 0xffu) : 0);
      const int32_t nxt = lane + need;
      // the chain of real sequence starts through this window
      const unsigned long long smask = __ballot(simple);
      unsigned long long real = 0ull;
      int32_t t = t0;
      while (t < 64 && ((smask >> t) & 1ull)) {
        real |= 1ull << t;
        t = __builtin_amdgcn_readlane(nxt, t);
      }
      const bool is_real = (real >> lane) & 1ull;
      const int32_t v = is_real ? lit + ml : 0;
      const int32_t incl = wave_inclusive_sum(v, lane);
      const int32_t my_op = op + incl - v, D = my_op + lit;
      const bool bad = is_real && (lit > kLz4BlockMax - my_op || offset == 0 || offset > D || ml > kLz4BlockMax - D);
      if (__any(bad)) { ok = false; break; }
      if (is_real) {
        const int32_t rank = __builtin_popcountll(real & ((1ull << lane) - 1ull));
        out[ns + rank] = make_uint2((uint32_t)my_op | ((uint32_t)lit << 16), (uint32_t)ml | ((uint32_t)offset << 16));
      }
      // the literal bytes, exactly (two sequences of one window may lie closer together than a wide store is long)
      {
        const int32_t my_lit = is_real ? lit : 0;
        const uint64_t lits_lo = (lo >> 8) | (hi << 56), lits_hi = hi >> 8;
        for (int32_t j = 0; __any(j < my_lit); ++j)
          if (j < my_lit) dst[my_op + j] = (uint8_t)((j < 8 ? lits_lo : lits_hi) >> (8 * (j & 7)));
      }
      ns += __builtin_popcountll(real);
      op += __builtin_amdgcn_readlane(incl, 63);
      if (t >= 64) {
        base += 64;
        t0 = t - 64;
      } else {
        int32_t ip = base + t;
        if (ip >= n_in) break;
        // (stores of this wave to one address keep their order: the literal bytes written above are not overtaken)
        if (!lz4_slow_sequence(in, n_in, dst, ip, op, out, ns, lane)) { ok = false; break; }
        base = ip & ~63;
        t0 = ip & 63;
      }
    }
    cls = 0;
    while (cls < kLz4Classes - 1 && op > kLz4ClassCap[cls]) ++cls;
  }
  if (ok && !blk.last && op != kLz4BlockMax) ok = false;  // every block of a frame but its last is exactly full
  if (lane != 0) return;
  w.n_seq[b] = ns;
  if (!ok) {
    w.state[b] = -1;
    atomicMin(&err->lz4_bad, (unsigned int)blk.section);
    if (blk.last) sections[blk.section].byte_len = 0;
    return;
  }
  w.state[b] = op;
  if (blk.last) sections[blk.section].byte_len = (int64_t)blk.index * kLz4BlockMax + op;
  if (op > 0) w.cls_list[(int64_t)cls * n_blocks + atomicAdd(&w.cls_count[cls], 1)] = b;
}

// cls == kLz4Classes: stored blocks (launched without LDS).  LDS: the block's image (the class's capacity), then the byte
// map (kLz4Map 16-bit positions).  A masked-off lane may READ up to 63 bytes past the image — the start of the map, harmless;
// nothing is written past the class's capacity — so the image needs no pad of its own: 16 KiB + 4 KiB = 20 480 bytes is
// exactly an eighth of a CU's 160 KiB (the 64-byte pad of round 4 made it seven waves per CU for the batches the reference's
// 16 KiB producer writes).
// BATCHED = false: the map read per window and the dependency rounds through a cross-lane read, as first written — kept
// selectable (SURGE_INGEST_LZ4_WINDOWS=1) for a same-box comparison (profiles/r04_e2e_lz4_windows.txt: the batched loop
// is 1.8 % faster end to end — the kernels' LDS footprint x time bounds the path, not this loop's latency alone).
template <bool BATCHED>
__global__ void __launch_bounds__(64) lz4_exec_kernel(const uint8_t* __restrict__ bytes, uint8_t* __restrict__ out_base, const Lz4Block* __restrict__ blocks,
                                                      int32_t n_blocks, Lz4Work w, int32_t cls, int32_t cap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lz4_out[];
  uint16_t* const map = (uint16_t*)(lz4_out + cap + w.pad);
  const int lane = threadIdx.x;
  const int32_t count = w.cls_count[cls];
  for (int32_t k = (int32_t)blockIdx.x; k < count; k += (int32_t)gridDim.x) {
    const int32_t b = w.cls_list[(int64_t)cls * n_blocks + k];
    const Lz4Block blk = blocks[b];
    uint8_t* __restrict__ dst = out_base + blk.dst_off;  // 16-byte aligned: dst_off is a multiple of 64 KiB from an aligned base
    const int32_t size = w.state[b];
    if (cls == kLz4Classes) {  // stored: the bytes as they are
      const uint8_t* __restrict__ in = bytes + blk.src_off;
      for (int i = lane; i < size; i += 64) dst[i] = in[i];
      continue;
    }
    // the image with the literals the first pass put in place
    const int n16 = (size + 15) >> 4;
    for (int i = lane; i < n16; i += 64) ((uint4*)lz4_out)[i] = ((const uint4*)dst)[i];
    const int32_t n = w.n_seq[b];
    const uint2* __restrict__ sq = w.seq + blk.seq_off;
    uint2 e_next = lane < n ? sq[lane] : make_uint2(0u, 0u);
    for (int32_t g = 0; g < (w.dbg == 1 ? 0 : n); g += 64) {
      const uint2 e = e_next;
      if (g + 64 < n) e_next = g + 64 + lane < n ? sq[g + 64 + lane] : make_uint2(0u, 0u);  // in flight during this group
      const int32_t cnt = n - g < 64 ? n - g : 64;
      const int32_t op = (int32_t)(e.x & 0xffffu), lit = (int32_t)(e.x >> 16), M = (int32_t)(e.y & 0xffffu), O = (int32_t)(e.y >> 16);
      const int32_t D = op + lit, end = D + M;
      const int32_t g0 = __builtin_amdgcn_readfirstlane(op), g1 = __builtin_amdgcn_readlane(end, cnt - 1);
      const bool mapped = g1 - g0 <= kLz4Map && !__any(lane < cnt && M > 0 && (O < M || M > 256));
      if (mapped) {
        // (LDS operations of one wave execute in order: a read sees every earlier write of any lane of this wave)
        for (int32_t p = g0 + lane; p < g1; p += 64) map[p & (kLz4Map - 1)] = (uint16_t)p;
        const int32_t m_lane = lane < cnt ? M : 0;
        for (int32_t i = 0; __any(i < m_lane); ++i)
          if (i < m_lane) map[(D + i) & (kLz4Map - 1)] = (uint16_t)(D - O + i);
        // Eight windows' map entries are read together (the map does not change while it is applied): one LDS round trip
        // for eight windows instead of one each.  A window is then: gather, write, and per dependency round a ballot, a
        // gather and a write — whether a byte's source is written yet is a bit of the ballot, not a cross-lane read.
        constexpr int kWin = 8;
        if (w.dbg == 2) continue;
        if (!BATCHED) {
          for (int32_t p = g0; p < g1; p += 64) {
            const int32_t pos = p + lane;
            const bool act = pos < g1;
            const int32_t src = act ? (int32_t)map[pos & (kLz4Map - 1)] : pos;
            const bool dep = act && src >= p && src != pos;
            bool done = !dep;
            {
              const uint8_t v = lz4_out[src];
              if (act && !dep) lz4_out[pos] = v;
            }
            while (__any(!done)) {
              const bool src_done = __shfl((int)done, (src - p) & 63, 64) != 0;
              const bool can = !done && src_done;
              const uint8_t v = lz4_out[src];
              if (can) lz4_out[pos] = v;
              done = done || can;
            }
          }
        }
        for (int32_t p0 = g0; BATCHED && p0 < g1; p0 += 64 * kWin) {
          int32_t srcs[kWin];
#pragma unroll
          for (int k = 0; k < kWin; ++k) {
            const int32_t pos = p0 + 64 * k + lane;
            srcs[k] = pos < g1 ? (int32_t)map[pos & (kLz4Map - 1)] : pos;
          }
#pragma unroll
          for (int k = 0; k < kWin; ++k) {
            const int32_t p = p0 + 64 * k;
            if (p >= g1) break;  // (wave-uniform)
            const int32_t pos = p + lane;
            const bool act = pos < g1;
            const int32_t src = srcs[k];
            // Bytes that copy from THIS window wait for their source: first everything whose source lies below the window (or
            // is the byte itself: a literal), then, round by round, the bytes whose source was written in the round before —
            // as many rounds as the longest chain inside the window is deep (one or two; the first version ran one round per
            // dependent BYTE: a 13-byte match at offset 20 cost 14 rounds).
            const bool dep = act && src >= p && src != pos;
            bool done = !dep;
            {
              const uint8_t v = lz4_out[src];
              if (act && !dep) lz4_out[pos] = v;
            }
            unsigned long long dm = __ballot(done);
            while (dm != ~0ull) {
              const bool can = !done && ((dm >> ((src - p) & 63)) & 1ull) != 0ull;
              const uint8_t v = lz4_out[src];
              if (can) lz4_out[pos] = v;
              done = done || can;
              dm = __ballot(done);
            }
          }
        }
      } else {
        // sequence by sequence, strictly in order
        for (int32_t s = 0; s < cnt; ++s) {
          const int32_t Ms = __builtin_amdgcn_readlane(M, s);
          if (Ms == 0) continue;  // the block's last sequence
          const int32_t Os = __builtin_amdgcn_readlane(O, s), Ds = __builtin_amdgcn_readlane(D, s);
          const uint8_t* src = lz4_out + Ds - Os;
          if (Os >= Ms || Os >= 64) {  // chunks of 64 bytes never read what the same instruction writes
            for (int32_t i = lane; i < Ms; i += 64) lz4_out[Ds + i] = src[i];
          } else {                     // the period IS the data
            for (int32_t i = lane; i < Ms; i += 64) lz4_out[Ds + i] = src[i % Os];
          }
        }
      }
    }
    const int f16 = size >> 4;
    for (int i = lane; i < f16; i += 64) ((uint4*)dst)[i] = ((const uint4*)lz4_out)[i];
    for (int i = (f16 << 4) + lane; i < size; i += 64) dst[i] = lz4_out[i];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the copy-out has read the LDS before the next block overwrites it
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- records: chain, parse, decode (round 4: one workgroup per batch, out of LDS) -----------------------------------------
// Round 3 ran three kernels over the decompressed bytes in global memory — chain (one thread per batch walking its
// records' length varints), parse (one thread per record), resolve (one thread per record walking the JSON text byte by
// byte) — every step a dependent single-byte load: 1.45 ms per 1 M records, almost all of it waiting (SQ_WAIT_ANY 3/4 of
// the wave cycles; the LDS staging resolve_kernel had never fired on lz4 topics: a block's 256 records came from two
// batches whose decompressed bytes lie 64 KiB apart).  Now one workgroup owns one batch: it copies the batch's records
// section into LDS with 16-byte loads, one thread walks the length varints there (the only sequential step, now at LDS
// latency), then every thread parses its record and decodes its value from LDS.  A section that does not fit the LDS of
// its launch is read in place (same code, global pointers).

// Four bytes from p on, whatever its alignment, as ONE load instruction: the two aligned dwords that hold them
// (ds_read2_b32 / global_load_dwordx2), funnel-shifted into place.  Reads up to 7 bytes past p: the staged bytes and the
// LDS copies of a section end at least 16 bytes after their last byte.  (Round 5: the record and JSON walks read their bytes
// one ds_read_u8 at a time — ~200 dependent LDS reads per record; section_kernel waited 71 % of its wave cycles.)
// (the aligned pointer is derived from p by pointer arithmetic, not through an integer: the compiler keeps p's address space —
// global_load for the staged bytes, not flat_load)
__device__ __forceinline__ uint32_t load4(lds_ptr_t p) {
  const uint32_t sh = (uint32_t)(uintptr_t)p & 3u;
  const __attribute__((address_space(3))) uint32_t* q = (const __attribute__((address_space(3))) uint32_t*)(p - sh);
  return __builtin_amdgcn_alignbyte(q[1], q[0], sh);
}
__device__ __forceinline__ uint32_t load4(const uint8_t* p) {
  const uint32_t sh = (uint32_t)((uintptr_t)p & 3u);
  const uint32_t* q = (const uint32_t*)(p - sh);
  return __builtin_amdgcn_alignbyte(q[1], q[0], sh);
}
// 0x80 in every byte of x that equals c (exact in the lowest matching byte and below it — all a first-match search needs)
__device__ __forceinline__ uint32_t bytes_eq(uint32_t x, uint32_t c) {
  const uint32_t v = x ^ (c * 0x01010101u);
  return (v - 0x01010101u) & ~v & 0x80808080u;
}

template <typename P>
struct ReaderT {
  P p;
  P end;
  bool ok;
  __device__ int64_t varlong() {
    // one and two bytes without the loop: a record's length, its key / value lengths and its offset delta almost always
    // (the counters put most of section_kernel's scalar instructions into the exec-mask bookkeeping of these loops)
    if (end - p >= 2) {
      const uint32_t w2 = load4(p);
      const uint32_t b0 = w2 & 0xffu, b1 = (w2 >> 8) & 0xffu;
      if (!(b0 & 0x80u)) { ++p; return (int64_t)(b0 >> 1) ^ -(int64_t)(b0 & 1u); }
      if (!(b1 & 0x80u)) {
        p += 2;
        const uint32_t w = (b0 & 0x7fu) | (b1 << 7);
        return (int64_t)(w >> 1) ^ -(int64_t)(w & 1u);
      }
    }
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

// 64-bit hash of an aggregate id under the table's seed (a re-seed follows a detected collision); 0 marks an empty slot.
// (begin / end: the record parser hashes the id while it looks for its ':')
__device__ __forceinline__ uint64_t hash_key_begin(uint64_t seed) { return 0x9E3779B97F4A7C15ull + seed * 0xC2B2AE3D27D4EB4Full; }
__device__ __forceinline__ uint64_t hash_key_end(uint64_t h, int n, uint64_t seed) {
  h ^= (uint64_t)n * 0xFF51AFD7ED558CCDull;
  h ^= h >> 29;
  h *= 0xD6E8FEB86659FD93ull;
  h ^= h >> 32;
  if (seed >> 63) h &= 0xffull;  // test hook (SURGE_INGEST_DEBUG_WEAK_HASH): collisions guaranteed until the first re-seed
  return h == 0ull ? 1ull : h;
}
template <typename P>
__device__ __forceinline__ uint64_t hash_key(P p, int n, uint64_t seed) {
  uint64_t h = hash_key_begin(seed);
  for (int i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
  return hash_key_end(h, n, seed);
}

// ---- event values ------------------------------------------------------------------------------------------------------
template <typename P>
struct JsonScanT {
  P p;
  P end;
  __device__ void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  __device__ bool string(P* s, int* len, bool* escaped) {
    if (p >= end || *p != '"') return false;
    ++p;
    *s = p;
    *escaped = false;
    for (;;) {
      // four bytes at a time to the next quote or backslash
      while (end - p >= 4) {
        const uint32_t x = load4(p);
        const uint32_t m = bytes_eq(x, '"') | bytes_eq(x, '\\');
        if (m) { p += (__builtin_ctz(m) >> 3); break; }
        p += 4;
      }
      if (p >= end) return false;
      const uint8_t c = *p;
      if (c == '"') break;
      if (c == '\\') {
        *escaped = true;
        ++p;
        if (p >= end) return false;
      }
      ++p;
    }
    *len = (int)(p - *s);
    ++p;
    return true;
  }
  __device__ bool number(P* s, int* len) {
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
    P s; int l; bool e;
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

// (`want` is one of EvjDevice's name arrays: 8-byte aligned, zero-padded to 64 bytes)
template <typename P>
__device__ __forceinline__ bool name_is(const char* want, int want_len, P got, int got_len) {
  if (want_len != got_len) return false;
  const uint32_t* w4 = (const uint32_t*)want;
  for (int i = 0; i < want_len; i += 4) {
    const int left = want_len - i;
    const uint32_t keep = left >= 4 ? 0xffffffffu : ((1u << (8 * left)) - 1u);
    if (((w4[i >> 2] ^ load4(got + i)) & keep) != 0u) return false;
  }
  return true;
}

// The template as the kernels use it: the distinct field names the decoder has to look for, with their lengths, and per
// event type which of them carry its sequence number / argument — so ONE pass over the object finds everything (round 3
// walked the text twice: once for the discriminator, once for the fields of the type it selected).
constexpr int kEvjNames = 1 + 2 * SURGE_EVJ_MAX_TYPES;
struct EvjDevice {
  uint32_t n_types, n_names;                  // names[0] is the discriminator ("" when the template has none)
  alignas(8) char names[kEvjNames][SURGE_EVJ_NAME];
  uint8_t name_len[kEvjNames];
  alignas(8) char type_name[SURGE_EVJ_MAX_TYPES][SURGE_EVJ_NAME];
  uint8_t type_name_len[SURGE_EVJ_MAX_TYPES];
  uint8_t seq_name[SURGE_EVJ_MAX_TYPES], arg_name[SURGE_EVJ_MAX_TYPES];  // index into names, 0xff = none
  uint32_t event_type[SURGE_EVJ_MAX_TYPES], arg_kind[SURGE_EVJ_MAX_TYPES];
};

template <typename P>
struct FoundT {
  P s;
  int len;
  bool present, is_num, is_str, escaped;
};

constexpr int kEvjTrack = 8;  // numeric field names tracked in registers; templates with more use the generic lookup below

template <typename P>
__device__ int parse_i32(P s, int len, int32_t* out) {  // 0 ok, else not an Int
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

// the rules of surge_event_json_decode (event_decode.cpp), on the device, in one pass over the object: the discriminator
// is the LAST field of its name among the first 24 fields with unescaped names, a sequence / argument field the FIRST
// numeric field of its name among them
template <typename P>
__device__ uint32_t decode_json_event(const EvjDevice* __restrict__ t, const surge::F64ParseTable* ptab, P v, int len, uint4* out) {
  JsonScanT<P> sc{v, v + len};
  FoundT<P> disc;
  disc.present = false; disc.is_num = disc.is_str = disc.escaped = false; disc.len = 0; disc.s = v;
  P num_s[kEvjTrack];
  int num_len[kEvjTrack];
#pragma unroll
  for (int j = 0; j < kEvjTrack; ++j) { num_s[j] = v; num_len[j] = -1; }
  const int n_names = (int)t->n_names;
  sc.ws();
  if (sc.p >= sc.end || *sc.p != '{') return RS_JSON;
  ++sc.p;
  sc.ws();
  if (sc.p < sc.end && *sc.p == '}') {
    ++sc.p;
  } else {
    int n_fields = 0;
    for (;;) {
      sc.ws();
      P key; int key_len; bool esc = false;
      if (!sc.string(&key, &key_len, &esc)) return RS_JSON;
      sc.ws();
      if (sc.p >= sc.end || *sc.p != ':') return RS_JSON;
      ++sc.p;
      sc.ws();
      if (sc.p >= sc.end) return RS_JSON;
      FoundT<P> cur;
      cur.s = sc.p; cur.len = 0; cur.present = true; cur.is_num = cur.is_str = cur.escaped = false;
      if (*sc.p == '"') {
        if (!sc.string(&cur.s, &cur.len, &cur.escaped)) return RS_JSON;
        cur.is_str = true;
      } else if (*sc.p == '-' || (*sc.p >= '0' && *sc.p <= '9')) {
        if (!sc.number(&cur.s, &cur.len)) return RS_JSON;
        cur.is_num = true;
      } else if (!sc.skip_value()) {
        return RS_JSON;
      }
      if (!esc && n_fields < 24) {  // the host decoder remembers 24 fields and ignores names with escapes
        ++n_fields;
        if (t->name_len[0] && name_is(t->names[0], t->name_len[0], key, key_len)) disc = cur;
        if (cur.is_num) {
#pragma unroll
          for (int j = 0; j < kEvjTrack; ++j)
            if (j + 1 < n_names && num_len[j] < 0 && name_is(t->names[j + 1], t->name_len[j + 1], key, key_len)) { num_s[j] = cur.s; num_len[j] = cur.len; }
        }
      }
      sc.ws();
      if (sc.p < sc.end && *sc.p == ',') { ++sc.p; continue; }
      if (sc.p < sc.end && *sc.p == '}') { ++sc.p; break; }
      return RS_JSON;
    }
  }
  sc.ws();
  if (sc.p != sc.end) return RS_JSON;
  int ty = -1;
  if (t->name_len[0] == 0) {
    ty = 0;
  } else {
    if (!disc.present || !disc.is_str) return RS_FIELD;
    for (uint32_t i = 0; i < t->n_types && ty < 0; ++i)
      if (!disc.escaped && name_is(t->type_name[i], t->type_name_len[i], disc.s, disc.len)) ty = (int)i;
    if (ty < 0) return RS_TYPE;
  }
  auto pick = [&](int name_idx, P* s, int* l) {  // the tracked field of that name (name_idx >= 1)
    *l = -1;
#pragma unroll
    for (int j = 0; j < kEvjTrack; ++j)
      if (j + 1 == name_idx) { *s = num_s[j]; *l = num_len[j]; }
  };
  int32_t seq = 0;
  if (t->seq_name[ty] != 0xff) {
    P s = v; int l;
    pick(t->seq_name[ty], &s, &l);
    if (l < 0 || parse_i32(s, l, &seq) != 0) return RS_FIELD;
  }
  uint64_t raw = 0;
  uint32_t status = RS_OK;
  if (t->arg_kind[ty] != SURGE_EVJ_ARG_NONE) {
    P s = v; int l;
    pick(t->arg_name[ty], &s, &l);
    if (l < 0) return RS_FIELD;
    if (t->arg_kind[ty] == SURGE_EVJ_ARG_I32) {
      int32_t a = 0;
      if (parse_i32(s, l, &a) != 0) return RS_FIELD;
      raw = (uint64_t)(uint32_t)a;
    } else {
      const int rc = surge::f64_parse_json_number((const uint8_t*)s, l, ptab, &raw);  // (a flat pointer: the parser is shared with the host)
      if (rc == surge::F64_PARSE_MALFORMED) return RS_FIELD;
      if (rc == surge::F64_PARSE_AMBIGUOUS) status = RS_F64_HOST;  // type and seq are final; the host fills in the payload
    }
  }
  *out = make_uint4(t->event_type[ty], (uint32_t)seq, (uint32_t)raw, (uint32_t)(raw >> 32));
  return status;
}

// (Tried in round 4 and dropped: the same grammar as a table-driven automaton — one loop over the bytes in lockstep, next
// state looked up by (state, byte) in LDS, field names recognised by length + hash and verified afterwards, this walk as
// the fallback.  It traded the walk's scalar exec-mask bookkeeping for twice the vector instructions: section_kernel
// alone 0.70 -> 0.77 ms per 1 M records.  The counters say where this kernel's scalar work really is: chaining and
// parsing the varint-framed records, not the JSON.)
struct JsonCtx {  // what the value decoder needs besides the value
  const EvjDevice* tmpl;  // nullptr: 16-byte fixed events
  const surge::F64ParseTable* ptab;
  int32_t dbg = 0;  // SURGE_DBG_DECODE (timing experiments only, results are NOT the topic's): 1 = sections are staged, nothing else; 2 = staged and chained, no record decoded
#ifdef SURGE_EXPERIMENTS
  unsigned long long* ticks = nullptr;  // per workgroup: wall clock (10 ns) at start, after staging, after the chain, at the end
#endif
  int32_t wave_chain = 1;  // a staged batch's records are found by its first wave, every lane in its own chunk (chain_records_parallel); 0: walked by one lane (SURGE_INGEST_CHAIN=lane)
};

// a record value -> event16 + status
template <typename P>
__device__ __forceinline__ uint32_t decode_value(const JsonCtx& jc, P vp, int val_len, uint4* e) {
  if (jc.tmpl) return decode_json_event(jc.tmpl, jc.ptab, vp, val_len, e);
  if (val_len != 16) return RS_SIZE;
  uint32_t w[4];
  for (int q = 0; q < 4; ++q) w[q] = (uint32_t)vp[4 * q] | ((uint32_t)vp[4 * q + 1] << 8) | ((uint32_t)vp[4 * q + 2] << 16) | ((uint32_t)vp[4 * q + 3] << 24);
  *e = make_uint4(w[0], w[1], w[2], w[3]);
  return RS_OK;
}

// One record body [body, end) of a section that starts at `base` (absolute offset sec_off in the staged bytes); `valid` =
// this lane has a record.
template <typename P>
__device__ __forceinline__ void decode_record(P base, int64_t sec_off, int64_t base_offset, bool valid, int32_t body, int32_t end, int64_t gi, uint64_t seed,
                                              const JsonCtx& jc, RecMeta* __restrict__ meta, uint4* __restrict__ ev_tmp,
                                              uint32_t* __restrict__ f64_host_list, ErrorCell* err) {
  RecMeta m;
  m.key_off = m.val_off = m.offset = 0; m.hash = 0; m.key_len = m.val_len = 0; m.slot = 0; m.status = RS_MALFORMED;
  uint4 e = make_uint4(0, 0, 0, 0);
  bool has_value = false;
  P val = base;
  int vlen32 = 0;
  if (valid && body >= 0) {
    ReaderT<P> q{base + body, base + end, true};
    if (q.p < q.end) ++q.p; else q.ok = false;  // attributes
    (void)q.varlong();                            // timestampDelta
    const int64_t offset_delta = q.varlong();
    const int64_t klen = q.varlong();
    P key = q.p;
    if (q.ok && klen > 0) { if (q.end - q.p >= klen) q.p += klen; else q.ok = false; }
    const int64_t vlen = q.ok ? q.varlong() : 0;
    val = q.p;
    if (q.ok && vlen > 0) { if (q.end - q.p >= vlen) q.p += vlen; else q.ok = false; }
    // (headers follow; the chain already knows where the record ends)
    if (q.ok && klen >= -1 && vlen >= -1 && klen < (1ll << 31) && vlen < (1ll << 31)) {
      m.offset = base_offset + offset_delta;
      if (klen == 0 && vlen == 0) {
        m.status = RS_SKIP;  // KafkaProducerActorImpl.scala:322-329
      } else if (klen < 0 || vlen < 0) {
        m.status = RS_NULL;
      } else {
        int n = 0;
        uint64_t hk = hash_key_begin(seed);
        for (bool colon = false; n < (int)klen && !colon;) {  // PartitionStringUpToColon (KafkaPartitioner.scala:38-42), hashed on the way
          uint32_t w4 = load4(key + n);
          const int take = (int)klen - n < 4 ? (int)klen - n : 4;
          for (int b = 0; b < take; ++b, w4 >>= 8) {
            const uint32_t c = w4 & 0xffu;
            if (c == (uint32_t)':') { colon = true; break; }
            hk = (hk ^ c) * 0x100000001B3ull;
            ++n;
          }
        }
        m.key_off = sec_off + (key - base);
        m.key_len = n;
        m.val_off = sec_off + (val - base);
        m.val_len = (int32_t)vlen;
        m.hash = hash_key_end(hk, n, seed);
        has_value = true;
        vlen32 = (int32_t)vlen;
      }
    }
  }
  if (has_value) {
    uint32_t st = decode_value(jc, val, vlen32, &e);
    if (st == RS_F64_HOST) {
      f64_host_list[atomicAdd(&err->n_f64_host, 1u)] = (uint32_t)gi;
      st = RS_OK;
    }
    m.status = st;
  }
  if (!valid) return;
  if (m.status >= RS_NULL) report(err, gi, m.status);
  meta[gi] = m;
  ev_tmp[gi] = e;
}

// Workgroup sizes of section_kernel.  The decode walks are latency-bound: what counts is how many batches a CU has in
// flight, and a workgroup's lanes beyond its batch's records only take registers away from other batches.  A 16 KiB batch of
// the reference's publisher holds ~140 play-json events: 192 lanes decode it in one round where 256 idled 45 % of theirs — and
// at 90 VGPRs a CU held 5 workgroups of four waves; three-wave workgroups at <= 80 VGPRs make it the 8 the LDS allows.  A
// push picks the smallest size that takes its largest batch's records in one round (64 for small publisher flushes).
constexpr int kSecRecs = 256;  // records chained per round (LDS: two int32 per record)

// the length varints of up to kSecRecs records from relative position *pos on; false = unreadable from record `bad` on
// (This walk is the one sequential step of a batch — one lane works, 255 wait: per record ONE load (the four bytes at the
// record's start hold its whole length varint unless the record is 256 KiB or longer) and a dozen 32-bit instructions.  The
// first version went through ReaderT's general varlong: two byte loads, 64-bit pointer arithmetic and a loop the compiler
// could not drop — 0.19 of section_kernel's 0.55 ms per 10^6 records, measured with the decode switched off.)
template <typename P>
__device__ bool chain_records(P base, int32_t len, int32_t* pos, int32_t cnt, int32_t* rec_body, int32_t* rec_end, int32_t* bad) {
  int32_t at = *pos;
  for (int32_t i = 0; i < cnt; ++i) {
    const int32_t room = len - at;
    const uint32_t w = room > 0 ? load4(base + at) : 0x80808080u;
    uint32_t v;
    int32_t n;
    if (!(w & 0x80u)) { v = w & 0x7fu; n = 1; }
    else if (!(w & 0x8000u)) { v = (w & 0x7fu) | ((w >> 1) & 0x3f80u); n = 2; }
    else if (!(w & 0x800000u)) { v = (w & 0x7fu) | ((w >> 1) & 0x3f80u) | ((w >> 2) & 0x1fc000u); n = 3; }
    else {  // four bytes or more (or nothing left): the general reader decides
      ReaderT<P> r{base + at, base + len, true};
      const int64_t l = r.varlong();
      const int32_t body = (int32_t)(r.p - base);
      if (!r.ok || l < 0 || r.end - r.p < l) { *bad = i; return false; }
      rec_body[i] = body;
      rec_end[i] = body + (int32_t)l;
      at = body + (int32_t)l;
      continue;
    }
    const int32_t l = (int32_t)(v >> 1) ^ -(int32_t)(v & 1u);
    const int32_t body = at + n;
    if (n > room || l < 0 || l > len - body) { *bad = i; return false; }
    rec_body[i] = body;
    rec_end[i] = body + l;
    at = body + l;
  }
  *pos = at;
  return true;
}

// The same walk by the batch's first WAVE, every lane on its own 1/64th of the section (the walk above is
// the one sequential step of a batch — half of section_kernel's time alone, profiles/r06_section_chain_ab.txt).  A record's
// start cannot be computed without the records before it, but it can be RECOGNISED and the recognition verified:
//   * find: lane k looks in its chunk [k C, (k + 1) C) for the first position that reads as a record start — a length
//     varint of one to three bytes, the attributes byte 0 behind it (kafka-clients writes 0; zero bytes are found four at a
//     time), the record ending inside the section, and the same again at the position where it ends (lane 0: position 0);
//   * walk: from there the lane follows the lengths to the first start at or behind its chunk's end, counting its records;
//   * link: the find is a guess, the walk is exact GIVEN a true start — so if every lane's find is the position the nearest
//     lane in front of it walked to (lane 0's is 0, which is a start by definition), every lane stood on the true chain, by
//     induction; the last walk has to end at the section's end and the counts have to add up to the batch's record count.
//   * write: an exclusive prefix sum of the counts gives each lane its records' indices; it walks its chunk once more and
//     writes their bounds.
// Anything else — a find that does not link (binary values full of zero bytes can fake a start), a four-byte varint, a length
// that leaves the section, a count that differs — makes the function return false with nothing used: the one-lane walk then
// does what it always did, and reports what it always reported.  Called by the 64 lanes of a workgroup's first wave.
__device__ bool chain_records_parallel(lds_ptr_t base, int32_t len, int32_t cnt, int32_t* rec_body, int32_t* rec_end, int32_t* link) {
  const int lane = (int)(threadIdx.x & 63u);
  int32_t C = (((len + 63) >> 6) + 3) & ~3;  // a chunk: a multiple of four bytes ...
  C += (C & 4) ? 0 : 4;                      // ... and an odd number of dwords: lane l's w-th word then lies in bank (l C / 4 + w) % 64, a different one
                                             // for every lane (a 16 KiB section's 256-byte chunks put all 64 lanes on ONE bank: 64-way conflicts, measured 2 x slower)
  if (C > 284) return false;                       // (the find keeps a chunk's zero bytes in 288 bits: sections up to 17.75 KiB, a 16 KiB batch with room to spare)
  const int32_t lo = lane * C, hi = lo + C < len ? lo + C : len;
  // one step of the walk at p (< len): false = not a record start this walk follows
  auto step = [&](int32_t p, int32_t& body, int32_t& end) -> bool {
    const uint32_t w = load4(base + p);
    uint32_t v;
    int32_t n;
    if (!(w & 0x80u)) { v = w & 0x7fu; n = 1; }
    else if (!(w & 0x8000u)) { v = (w & 0x7fu) | ((w >> 1) & 0x3f80u); n = 2; }
    else if (!(w & 0x800000u)) { v = (w & 0x7fu) | ((w >> 1) & 0x3f80u) | ((w >> 2) & 0x1fc000u); n = 3; }
    else return false;
    if (v & 1u) return false;  // (zig-zag: a negative length)
    body = p + n;
    end = body + (int32_t)(v >> 1);
    return end <= len;  // (n bytes of varint fit as well: end >= body)
  };
  // find, part one (the same instructions in every lane: a wave pays for the union of its lanes' paths): the zero bytes of
  // positions [lo, lo + 288) — a record that starts in my chunk has its attributes byte there — as nine 32-bit masks
  uint32_t zm[9];
#pragma unroll
  for (int g = 0; g < 9; ++g) {
    uint32_t m = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int32_t q = lo + 32 * g + 4 * k;
      const uint32_t w = q < len ? load4(base + q) : 0xffffffffu;
      const uint32_t z = ~(((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w | 0x7f7f7f7fu);  // 0x80 exactly where a byte is zero
      m |= (((z >> 7) * 0x10204080u) >> 28) << (4 * k);                             // ... as four bits
    }
    const int32_t left = len - lo - 32 * g;  // positions of this mask inside the section
    zm[g] = left >= 32 ? m : left <= 0 ? 0u : m & ((1u << left) - 1u);
  }
  // find, part two: my zero bytes in ascending order, for each the starts one to three bytes in front of it, longest varint
  // first (= ascending position); a start has to read as a record — varint, attributes 0, at least the six one-byte fields,
  // inside the section — and so has the position where it ends.  At most 32 candidates: binary values are left to the one-lane walk
  int32_t s = -1;
  if (lane == 0) {
    s = len > 0 ? 0 : -1;
  } else {
    int g = 0, tests = 0;
    uint32_t cur = zm[0];
    while (s < 0 && tests < 32) {
      while (cur == 0u && g < 8) {
        ++g;
        cur = g == 1 ? zm[1] : g == 2 ? zm[2] : g == 3 ? zm[3] : g == 4 ? zm[4] : g == 5 ? zm[5] : g == 6 ? zm[6] : g == 7 ? zm[7] : zm[8];
      }
      if (cur == 0u) break;
      const int32_t at = lo + 32 * g + __builtin_ctz(cur);
      cur &= cur - 1u;
      if (at <= lo) continue;
      const uint32_t f = load4(base + at - 3);  // (at >= 5: lane >= 1, C >= 4)
      const uint32_t b3 = f & 0xffu, b2 = (f >> 8) & 0xffu, b1 = (f >> 16) & 0xffu;
      if (b1 & 0x80u) continue;
      for (int n = (b2 & 0x80u) ? ((b3 & 0x80u) ? 3 : 2) : 1; n >= 1 && s < 0; --n) {
        const int32_t p = at - n;
        if (p < lo || p >= hi) continue;
        ++tests;
        const uint32_t v = n == 1 ? b1 : n == 2 ? ((b2 & 0x7fu) | (b1 << 7)) : ((b3 & 0x7fu) | ((b2 & 0x7fu) << 7) | (b1 << 14));
        const int32_t l = (int32_t)(v >> 1), end = at + l;
        if ((v & 1u) || l < 6 || end > len) continue;
        if (end < len) {  // the record behind it
          const uint32_t w2 = load4(base + end);
          const int n2 = !(w2 & 0x80u) ? 1 : !(w2 & 0x8000u) ? 2 : !(w2 & 0x800000u) ? 3 : 0;
          if (n2 == 0) continue;
          const uint32_t v2 = n2 == 1 ? (w2 & 0x7fu) : n2 == 2 ? ((w2 & 0x7fu) | ((w2 >> 1) & 0x3f80u)) : ((w2 & 0x7fu) | ((w2 >> 1) & 0x3f80u) | ((w2 >> 2) & 0x1fc000u));
          const uint32_t attr2 = n2 == 3 ? (w2 >> 24) : (w2 >> (8 * n2)) & 0xffu;
          if ((v2 & 1u) || (v2 >> 1) < 6u || attr2 != 0u || end + n2 + (int32_t)(v2 >> 1) > len) continue;
        }
        s = p;
      }
    }
  }
  // walk my chunk: count
  int32_t c = 0, e = s;
  bool ok = true;
  if (s >= 0) {
    int32_t p = s;
    while (p < lo + C && p < len) {
      int32_t body, end;
      if (!step(p, body, end)) { ok = false; break; }
      ++c;
      p = end;
    }
    e = p;
  }
  link[lane] = e;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const unsigned long long have = __ballot(s >= 0);
  if (s >= 0 && lane > 0) {
    const unsigned long long front = have & ((1ull << lane) - 1ull);
    ok = ok && front != 0ull && link[63 - __builtin_clzll(front)] == s;
  }
  const int last = 63 - __builtin_clzll(have | 1ull);
  if (lane == last) ok = ok && e == len;
  // the counts: inclusive scan over the lanes
  int32_t incl = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int32_t up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  const int32_t total = __shfl(incl, 63, 64);
  if (__ballot(!ok) != 0ull || total != cnt || (have & 1ull) == 0ull) return false;
  // write
  if (s >= 0) {
    int32_t idx = incl - c, p = s;
    while (p < lo + C && p < len) {
      int32_t body, end;
      (void)step(p, body, end);
      rec_body[idx] = body;
      rec_end[idx] = end;
      ++idx;
      p = end;
    }
  }
  return true;
}

// A workgroup takes its section when lo_excl < byte_len and (byte_len <= cap or take_rest); byte_len <= cap is staged.
template <int kSecThreads>
__global__ void __launch_bounds__(kSecThreads) __attribute__((amdgpu_waves_per_eu(kSecThreads == 256 ? 5 : 6)))  // (four waves at 80 VGPRs spill)
section_kernel(const uint8_t* __restrict__ bytes, const Section* __restrict__ sections, int64_t n_sections, int64_t lo_excl, int64_t cap, int32_t take_rest,
               uint64_t seed, JsonCtx jc, RecMeta* __restrict__ meta, uint4* __restrict__ ev_tmp, uint32_t* __restrict__ f64_host_list, ErrorCell* err) {
  extern __shared__ __attribute__((aligned(16))) uint8_t sec_smem[];
  __shared__ int32_t s_pos, s_bad;
  __shared__ int32_t s_link[64];
  const Section sec = sections[blockIdx.x];
  if (sec.n_records <= 0) return;
  const int64_t len = sec.byte_len;
  if (!(len > lo_excl && (len <= cap || take_rest))) return;
  const bool staged = len <= cap && len < (1ll << 31);
#ifdef SURGE_EXPERIMENTS
#define SURGE_TICK(k) do { if (jc.ticks && threadIdx.x == 0) jc.ticks[4ull * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define SURGE_TICK(k) do { } while (0)
#endif
  SURGE_TICK(0);
  int32_t* const rec_body = (int32_t*)(sec_smem + ((cap + 47) & ~15ll));
  int32_t* const rec_end = rec_body + kSecRecs;
  const int64_t a0 = sec.byte_off & ~15ll;
  const int32_t skew = (int32_t)(sec.byte_off - a0);
  const int n16 = staged ? (int)((skew + len + 15) >> 4) : 0;
  if (staged) {  // (the staged bytes buffer ends 16 bytes after its last byte)
    for (int c = threadIdx.x; c < n16; c += kSecThreads) ((uint4*)sec_smem)[c] = *(const uint4*)(bytes + a0 + 16ll * c);
  }
  if (threadIdx.x == 0) { s_pos = 0; s_bad = -1; }
  __syncthreads();
  SURGE_TICK(1);
  if (jc.dbg) {  // timing experiments: every record of the batch is "skipped"
    RecMeta m;
    m.key_off = m.val_off = m.offset = 0; m.hash = 0; m.key_len = m.val_len = 0; m.slot = 0; m.status = RS_SKIP;
    for (int32_t i = threadIdx.x; i < sec.n_records; i += kSecThreads) { meta[sec.rec_first + i] = m; ev_tmp[sec.rec_first + i] = make_uint4(0, 0, 0, 0); }
    if (jc.dbg == 1) return;
  }
  if (len >= (1ll << 31)) {  // a section of 2 GiB: nothing writes one (a batch's length is an int32)
    for (int32_t i0 = 0; i0 < sec.n_records; i0 += kSecThreads)
      decode_record(bytes, 0, 0, i0 + (int32_t)threadIdx.x < sec.n_records, -1, -1, sec.rec_first + i0 + threadIdx.x, seed, jc, meta, ev_tmp, f64_host_list, err);
    return;
  }
  const lds_ptr_t lbase = (lds_ptr_t)sec_smem + skew;
  const uint8_t* gbase = bytes + sec.byte_off;
  for (int32_t r0 = 0; r0 < sec.n_records; r0 += kSecRecs) {
    const int32_t cnt = sec.n_records - r0 < kSecRecs ? sec.n_records - r0 : kSecRecs;
    int32_t chained = 0;
    if (staged && jc.wave_chain && sec.n_records <= kSecRecs && threadIdx.x < 64u) {  // (the first wave, all of it; a batch of one round)
      if (chain_records_parallel(lbase, (int32_t)len, cnt, rec_body, rec_end, s_link)) chained = cnt;
      else if (threadIdx.x == 0) atomicAdd(&err->reserved, 1u);
    }
    if (threadIdx.x == 0) {
      if (s_bad < 0) {
        if (chained < cnt) {  // not chained by the wave, or left by it at a record it does not read: the one-lane walk from there
          int32_t pos = s_pos, bad = 0;
          const bool ok = staged ? chain_records(lbase, (int32_t)len, &pos, cnt - chained, rec_body + chained, rec_end + chained, &bad)
                                 : chain_records(gbase, (int32_t)len, &pos, cnt - chained, rec_body + chained, rec_end + chained, &bad);
          s_pos = pos;
          if (!ok) {
            // everything from here to the end of the batch is unreadable: the rest is marked, the first is reported
            bad += chained;
            s_bad = r0 + bad;
            for (int32_t k = bad; k < cnt; ++k) rec_body[k] = -1;
            report(err, sec.rec_first + r0 + bad, RS_MALFORMED);
          }
        }
      } else {
        for (int32_t k = 0; k < cnt; ++k) rec_body[k] = -1;
      }
    }
    __syncthreads();
    SURGE_TICK(2);
    for (int32_t i0 = 0; i0 < (jc.dbg ? 0 : cnt); i0 += kSecThreads) {
      const int32_t i = i0 + (int32_t)threadIdx.x;
      const bool valid = i < cnt;
      const int32_t body = valid ? rec_body[i] : -1, end = body >= 0 ? rec_end[i] : -1;
      if (staged)
        decode_record(lbase, sec.byte_off, sec.base_offset, valid, body, end, sec.rec_first + r0 + i, seed, jc, meta, ev_tmp, f64_host_list, err);
      else
        decode_record(gbase, sec.byte_off, sec.base_offset, valid, body, end, sec.rec_first + r0 + i, seed, jc, meta, ev_tmp, f64_host_list, err);
    }
    __syncthreads();
    SURGE_TICK(3);
  }
}

// records that arrive already framed (a JVM's ConsumerRecords: key bytes, value bytes, offset per record): the bytes buffer
// holds the keys first, the values from `val_base` on
__global__ void __launch_bounds__(256) records_kernel(const uint8_t* __restrict__ bytes, const int64_t* __restrict__ key_off, const int64_t* __restrict__ val_off,
                                                      const int64_t* __restrict__ offsets, int64_t val_base, int64_t n_rec, uint64_t seed, JsonCtx jc,
                                                      RecMeta* __restrict__ meta, uint4* __restrict__ ev_tmp, uint32_t* __restrict__ f64_host_list, ErrorCell* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n_rec;
  RecMeta m;
  m.key_off = m.val_off = m.offset = 0; m.hash = 0; m.key_len = m.val_len = 0; m.slot = 0; m.status = RS_MALFORMED;
  uint4 e = make_uint4(0, 0, 0, 0);
  bool has_value = false;
  if (valid) {
    const int64_t k0 = key_off[i], k1 = key_off[i + 1], v0 = val_off[i], v1 = val_off[i + 1];
    m.key_off = k0; m.val_off = val_base + v0; m.offset = offsets ? offsets[i] : i;
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
      m.hash = hash_key(key, n, seed);
      has_value = true;
    }
  }
  if (!valid) return;
  if (has_value) {
    uint32_t st = decode_value(jc, bytes + m.val_off, m.val_len, &e);
    if (st == RS_F64_HOST) {
      f64_host_list[atomicAdd(&err->n_f64_host, 1u)] = (uint32_t)i;
      st = RS_OK;
    }
    m.status = st;
  }
  if (m.status >= RS_NULL) report(err, i, m.status);
  meta[i] = m;
  ev_tmp[i] = e;
}

// ---- interning the aggregate ids -----------------------------------------------------------------------------------------
// One 16-byte slot per entry (round 5; three parallel arrays before): a probe, the flag pass and the final gather each touch
// ONE random line of the 2^25-slot table per record instead of two or three.
struct TableSlot {
  unsigned long long hash;  // 0 = empty
  uint32_t key_id;          // 0xffffffff = not assigned yet (inserted by the push in flight)
  uint32_t first_rec;       // of a slot inserted by the push in flight: its first record (0xffffffff otherwise)
};
static_assert(sizeof(TableSlot) == 16, "one slot = one 16-byte store");
struct Table {
  TableSlot* s;
  uint64_t mask;
};

__global__ void table_clear_kernel(TableSlot* __restrict__ s, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ((uint4*)s)[i] = make_uint4(0u, 0u, 0xffffffffu, 0xffffffffu);
}

// n bytes at a == n bytes at b?  Sixteen bytes a round, every load of a round in flight at once (the first version compared
// byte by byte and stopped at the first difference: two dependent single-byte loads per byte — 110 us per 10^6 records on
// 13-byte ids, 550 us on 36-byte UUIDs, profiles/r06_e2e_*_depth1_kernel_stats.csv).  Reads at most 7 bytes past either end
// (the staged bytes and the key arena both end 16 bytes after their last byte).
__device__ __forceinline__ bool keys_equal(const uint8_t* a, const uint8_t* b, int n) {
  uint32_t diff = 0u;
  for (int i = 0; i < n; i += 16) {
    uint32_t x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int left = n - i - 4 * k;
      x[k] = left > 0 ? load4(a + i + 4 * k) ^ load4(b + i + 4 * k) : 0u;
      if (left < 4 && left > 0) x[k] &= (1u << (8 * left)) - 1u;
    }
    diff |= x[0] | x[1] | x[2] | x[3];
  }
  return diff == 0u;
}

// insert-or-find by hash; a slot this push inserts remembers its first record
__global__ void probe_kernel(RecMeta* __restrict__ meta, int64_t n_rec, Table t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  if (meta[i].status != RS_OK) return;
  const unsigned long long h = meta[i].hash;
  uint64_t s = h & t.mask;
  while (true) {
    const unsigned long long old = atomicCAS(&t.s[s].hash, 0ull, h);
    if (old == 0ull || old == h) break;
    s = (s + 1) & t.mask;
  }
  meta[i].slot = (uint32_t)s;
  if (t.s[s].key_id == 0xffffffffu) atomicMin(&t.s[s].first_rec, (uint32_t)i);
}

// Per record: is it the first record of a key this push discovers (those get the next ids, in record order: the host
// decoder's first-delivered numbering — an exclusive scan of the flags, no sort); does its key equal, byte for byte, the
// key its slot stands for (the key arena for a known key, the slot's first record for a new one): a 64-bit hash
// collision is detected here, before anything of the push is committed.  first[i] = flag << 40 | key length (scanned:
// id rank and arena offset in one pass); keep[i] = the record is delivered.
__global__ void flag_kernel(const RecMeta* __restrict__ meta, int64_t n_rec, const uint8_t* __restrict__ bytes, Table t, const uint8_t* __restrict__ arena,
                            const int64_t* __restrict__ key_off, unsigned long long* __restrict__ first, uint32_t* __restrict__ keep, ErrorCell* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_rec) return;
  unsigned long long f = 0ull;
  uint32_t k = 0u;
  if (i < n_rec) {
    const RecMeta m = meta[i];
    if (m.status == RS_OK) {
      const uint32_t id = t.s[m.slot].key_id;
      const uint8_t* kp = bytes + m.key_off;
      bool same;
      if (id != 0xffffffffu) {
        const int64_t a0 = key_off[id], a1 = key_off[id + 1];
        same = a1 - a0 == m.key_len && keys_equal(arena + a0, kp, m.key_len);
      } else {
        const uint32_t fr = t.s[m.slot].first_rec;
        if ((int64_t)fr == i) {
          same = true;
          f = (1ull << 40) | (unsigned long long)(uint32_t)m.key_len;
        } else {
          const RecMeta o = meta[fr];
          const uint8_t* op = bytes + o.key_off;
          same = o.key_len == m.key_len && keys_equal(op, kp, m.key_len);
        }
      }
      if (same) k = 1u; else report(err, i, RS_COLLISION);
    }
  }
  first[i] = f;  // (entry n_rec = 0: the scans' totals land there)
  keep[i] = k;
}

// the keys this push discovered: id = n_keys + rank, bytes to the arena; their slots stop being "new"
__global__ void assign_kernel(const RecMeta* __restrict__ meta, int64_t n_rec, const uint8_t* __restrict__ bytes, const unsigned long long* __restrict__ first,
                              const unsigned long long* __restrict__ first_scan, int64_t n_keys, int64_t arena_base, Table t, uint8_t* __restrict__ arena,
                              int64_t* __restrict__ key_off, unsigned long long* __restrict__ key_hash) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec || first[i] == 0ull) return;
  const RecMeta m = meta[i];
  const unsigned long long sc = first_scan[i];
  const int64_t id = n_keys + (int64_t)(sc >> 40);
  const int64_t dst = arena_base + (int64_t)(sc & ((1ull << 40) - 1));
  for (int b = 0; b < m.key_len; ++b) arena[dst + b] = bytes[m.key_off + b];
  key_off[id + 1] = dst + m.key_len;
  key_hash[id] = m.hash;
  t.s[m.slot].key_id = (uint32_t)id;
  t.s[m.slot].first_rec = 0xffffffffu;
}

// delivered records -> the result arrays (aggregate index from the record's slot)
__global__ void finalize_kernel(const RecMeta* __restrict__ meta, int64_t n_rec, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos, Table t,
                                const uint4* __restrict__ ev_tmp, int64_t out_base, int64_t* __restrict__ agg_out, uint4* __restrict__ ev_out,
                                int64_t* __restrict__ off_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec || !keep[i]) return;
  const int64_t o = out_base + pos[i];
  agg_out[o] = (int64_t)t.s[meta[i].slot].key_id;
  ev_out[o] = ev_tmp[i];
  off_out[o] = meta[i].offset;
}

// a push that fails after its keys were probed takes them out again: slots it inserted go back to empty (they only ever
// occupied slots that were empty before, so the table is what it was)
__global__ void rollback_kernel(const RecMeta* __restrict__ meta, int64_t n_rec, Table t) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec || meta[i].status != RS_OK) return;
  const uint32_t s = meta[i].slot;
  if (t.s[s].key_id == 0xffffffffu) {
    t.s[s].hash = 0ull;
    t.s[s].first_rec = 0xffffffffu;
  }
}

__global__ void rehash_kernel(const unsigned long long* __restrict__ key_hash, int64_t n_keys, Table t) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_keys) return;
  const unsigned long long h = key_hash[id];
  uint64_t s = h & t.mask;
  // (two known keys that collide under a new seed share a hash and get two slots: lookups of the second then find the
  // first, flag_kernel reports the mismatch and the table is re-seeded once more)
  while (atomicCAS(&t.s[s].hash, 0ull, h) != 0ull) s = (s + 1) & t.mask;
  t.s[s].key_id = (uint32_t)id;
}

// after a re-seed: every known key's hash from its bytes in the arena, every record's from its key in the staged bytes
__global__ void rekey_keys_kernel(const uint8_t* __restrict__ arena, const int64_t* __restrict__ key_off, int64_t n_keys, uint64_t seed,
                                  unsigned long long* __restrict__ key_hash) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_keys) return;
  key_hash[id] = hash_key(arena + key_off[id], (int)(key_off[id + 1] - key_off[id]), seed);
}
__global__ void rekey_records_kernel(RecMeta* __restrict__ meta, int64_t n_rec, const uint8_t* __restrict__ bytes, uint64_t seed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec || meta[i].status != RS_OK) return;
  meta[i].hash = hash_key(bytes + meta[i].key_off, meta[i].key_len, seed);
}

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes, bool keep, hipStream_t stream) {
    if (bytes <= cap) return hipSuccess;
    // room to spare: a fetch is a few per cent larger or smaller than the one before it, and a buffer that grows is freed —
    // hipFree waits for the whole device (1 - 3 ms with four pushes in flight: the spikes of round 4's per-fetch times)
    const size_t roomy = bytes + bytes / 4 + 4096;
    size_t want = cap * 2 > roomy ? cap * 2 : roomy;
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

// Everything one push needs until it is finished.  A push has two halves: stage 1 does not touch the key table — copy to
// the device, LZ4, chain / parse / decode — and runs on the slot's own stream; stage 2 — interning, compaction, append —
// runs on the decoder's stream, one push after the other.  With several slots the host enqueues stage 1 of the next
// fetches (surge_device_decoder_push_async) while stage 2 and the fold of the current one run: the copy engine, the
// latency-bound LZ4 kernels and the compute-bound decode of different fetches overlap on the chip.
struct PushSlot {
  Buf lz4_blocks, lz4_sizes, lz4_nseq, lz4_seq, lz4_cls;
  Buf d_bytes, d_sections, rec_a, rec_b, rec_c, meta, ev_tmp, f64_list, d_err;
  void* pinned = nullptr;
  size_t pinned_cap = 0;
  std::vector<Section> h_secs;      // (sources of asynchronous copies: they live as long as the slot is busy)
  std::vector<Lz4Block> h_blocks;
  std::vector<CrcSpan> h_crc;       // sections whose CRC-32C this push finishes on the device
  Buf crc_spans;
  ErrorCell h_err;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;      // recorded on `stream` behind stage 1
  hipEvent_t released = nullptr;  // recorded on the decoder's stream behind stage 2: the slot's buffers may be written again
  bool released_valid = false;
  int64_t n_rec = 0;
  uint64_t seed = 0;   // the hash function stage 1 hashed the keys with
  bool busy = false, wire = false;
};
constexpr int kSlots = 5;

}  // namespace

struct surge_device_decoder {
  int device = 0;
  hipStream_t stream = nullptr;
  bool json = false;
  // Two host threads may drive one decoder: one enqueues stage 1 (push_async / push_parts_async), the other finishes pushes
  // (push_finish*, result, clear, append_decoded*).  `mu` covers what both touch: the slot queue and the error text.
  std::mutex mu;
  std::atomic<bool> poisoned{false};  // a device error left the tables in an unknown state: every later push is refused
  std::string err;
  Buf d_tmpl, d_ptab;
  surge_event_json_template h_tmpl;  // (host copy: Doubles the device cannot decide are re-parsed with it)
  PushSlot slots[kSlots];
  // Stage 1 streams.  Consecutive pushes take consecutive streams (PushSlot::stream is set when a push claims its slot): with four
  // pushes in flight three are in stage 1 at any time, so THREE streams are all the concurrency there is — and with the
  // stream stage 2 and the fold run on that makes four, the number of hardware queues the HIP runtime creates by default
  // (GPU_MAX_HW_QUEUES).  Round 4 gave each of the five slots a stream of its own: the fifth and sixth stream of the process
  // shared a hardware queue with another one, and every fifth fetch waited for a neighbour's stage 1 in front of its stage 2
  // (profiles/r05_e2e_k512_per_fetch_trace.txt: finish 1.1 - 1.5 ms instead of 0.4 on fetches 8, 13, 18, 23, 28).
  hipStream_t push_streams[kSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int n_push_streams = 0;
  int n_push_active = 0;       // of them in rotation (<= n_push_streams): one fewer once a consumer folds on a stream of its own (hardware queues, below)
  bool push_streams_pinned = false;  // SURGE_INGEST_PUSH_STREAMS said how many: never adjusted
  uint64_t push_seq = 0;
  int head = 0;
  std::atomic<int> n_pending{0};  // slots [head, head + n_pending) hold pushes whose stage 1 is enqueued (changes under `mu`)
  // stage 2 scratch
  Buf first, first_scan, keep, keep_pos, temp;
  // hash table + key table
  Buf t_slots, arena, key_off, key_hash;
  uint64_t t_cap = 0;
  std::atomic<uint64_t> seed{0};  // (stage 1 reads it once per push; stage 2 of an earlier push may move it on: PushSlot::seed)
  int64_t n_keys = 0, arena_bytes = 0;
  // result
  Buf r_agg, r_ev, r_off;
  int64_t n_records = 0;
  // hand-over of the result arrays to a consumer on another stream (surge_replay_append_decoded_async): `consumed` is
  // recorded on the consumer's stream behind its last read, the next stage 2 waits for it before it writes the arrays
  hipEvent_t ready = nullptr, consumed = nullptr;
  void* crc_slice = nullptr;     // the CRC-32C slicing tables (4 KB) on the device: made at the first push that needs them
  hipEvent_t sleeper = nullptr;  // hipEventBlockingSync: host waits of the consumer thread sleep on it instead of spinning
  bool block_waits = true;
  int64_t poll_ns = 0;  // > 0 (SURGE_INGEST_WAIT=poll): the waits query the event between naps of this length instead
  bool consumed_valid = false;
  int64_t counters[4] = {0, 0, 0, 0};  // records seen, delivered, flush records skipped, f64 values re-parsed on the host
  int64_t chain_fallbacks = 0;         // batches whose records the one-lane walk chained after the parallel recognition declined (SURGE_EXPERIMENTS builds print it)
  int64_t reseeds = 0, pushes = 0;
  bool slots_sized = false;  // the first wire push has sized every slot's buffers like its own
};

namespace {

int32_t dfail(surge_device_decoder* d, int32_t code, const std::string& m) {
  if (d) {
    std::lock_guard<std::mutex> lk(d->mu);
    d->err = m;
  }
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
  t.s = (TableSlot*)d->t_slots.p;
  t.mask = d->t_cap - 1;
  return t;
}

// capacity for n_keys + extra more keys at a load factor of at most 1/2 (rebuilt from the key hashes when it grows)
int32_t ensure_table(surge_device_decoder* d, int64_t extra) {
  uint64_t need = 1024;
  while (need < (uint64_t)(d->n_keys + extra) * 2) need *= 2;
  if (need <= d->t_cap) return OK;
  if (need > (1ull << 32)) return dfail(d, E_UNSUPPORTED, "more than 2^31 aggregate ids");
  Buf fresh;
  DCHK(d, fresh.reserve(need * sizeof(TableSlot), false, d->stream));
  hipLaunchKernelGGL(table_clear_kernel, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, d->stream, (TableSlot*)fresh.p, need);
  d->t_slots.release();
  d->t_slots = fresh;
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

int32_t surge_ingest_group_use_pinned_slabs(surge_ingest_group* g) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return dfail(nullptr, E_DEVICE, "no usable HIP device: the slabs stay in pageable memory");
  return surge_ingest_group_set_allocator(g, pinned_alloc, pinned_release);
}

const char* surge_device_decoder_last_error(const surge_device_decoder* d) {
  if (d) {  // (a copy: the other thread of a two-thread host may be setting its own error text)
    std::lock_guard<std::mutex> lk(const_cast<surge_device_decoder*>(d)->mu);
    g_dec_err = d->err;
  }
  return g_dec_err.c_str();
}

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
  if (const char* v = std::getenv("SURGE_INGEST_DEBUG_WEAK_HASH"))
    if (v[0] == '1') d->seed = 1ull << 63;  // test hook: the table's first hash function keeps 8 bits, so keys collide and the re-seed runs
  int prev = 0;
  (void)hipGetDevice(&prev);
  int32_t rc = OK;
  auto init = [&]() -> int32_t {
    DCHK(d, hipSetDevice(device_id));
    {
      // Stage 1 rotates over THREE streams — with the decoder's own stream that makes four, the hardware queues the runtime
      // maps streams onto (GPU_MAX_HW_QUEUES).  A fifth stream shares a queue with another one, and a queue runs in order:
      // with the fold on a stream of its own every third push's interning sat behind a later push's whole stage 1 (2 ms
      // instead of 0.4: profiles/r06_e2e_consumer_waits_trace.txt).  So the hand-over calls (surge_replay_append_decoded_async,
      // surge_replay_stage_decoded) take one stream out of the rotation when the handle folds on another stream than the
      // decoder's.  Measured, fold on the decoder's stream: 3 streams 9.1 - 9.2e8, 2 streams 8.6 - 8.7e8 events/s
      // (profiles/r06_e2e_push_streams.txt); fold on its own stream: 3 streams 6.9 - 7.3e8, 2 streams 8.3 - 8.7e8.  A fourth
      // (low-priority) push stream: 6.8 - 7.0e8 with the 2.5 ms stalls back (profiles/r06_e2e_push_priority.txt).
      int n = 3;
      if (const char* v = std::getenv("SURGE_INGEST_PUSH_STREAMS")) {  // experiments: 5 = a stream per slot (round 4)
        n = std::atoi(v);
        d->push_streams_pinned = true;
      }
      n = n < 1 ? 1 : (n > kSlots ? kSlots : n);
      for (int i = 0; i < n; ++i) {
        // ... and at LOW priority: the runtime keeps a pool of hardware queues per priority, so the stage-1 streams cannot land on
        // the queue of the decoder's own stream or of the fold's whatever other streams the process has created (in bench.py's
        // default line, behind the legs that ran before it, one of three normal-priority push streams did: 7.4 instead of 9.1e8);
        // and stage 1 is the bulk work — interning and fold, the dependent chain, should win a tie (SURGE_INGEST_PUSH_PRIORITY=normal)
        int least = 0, greatest = 0;
        static const bool normal_prio = [] { const char* v = std::getenv("SURGE_INGEST_PUSH_PRIORITY"); return v && std::strcmp(v, "normal") == 0; }();
        if (!normal_prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
          DCHK(d, hipStreamCreateWithPriority(&d->push_streams[i], hipStreamNonBlocking, least));
        else
          DCHK(d, hipStreamCreateWithFlags(&d->push_streams[i], hipStreamNonBlocking));
        d->n_push_streams = d->n_push_active = i + 1;
      }
    }
    for (PushSlot& s : d->slots) {
      DCHK(d, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
      DCHK(d, hipEventCreateWithFlags(&s.released, hipEventDisableTiming));
      DCHK(d, s.d_err.reserve(sizeof(ErrorCell), false, d->stream));
    }
    DCHK(d, hipEventCreateWithFlags(&d->ready, hipEventDisableTiming));
    DCHK(d, hipEventCreateWithFlags(&d->consumed, hipEventDisableTiming));
    // A consumer thread that waits for the device a few times per fetch spins a core away in hipStreamSynchronize (0.4 - 1.2 ms of
    // CPU per 10^6-record fetch, and — on a host whose CPUs the framing threads need — 5 - 15 % of the bytes -> states rate:
    // profiles/r06_e2e_host_budget.jsonl).  The runtime's blocking wait (hipEventBlockingSync) still spins for its first few
    // hundred microseconds — most of a wait here — before it sleeps on the interrupt.  So the waits NAP: the event is queried
    // between nanosleeps of SURGE_INGEST_POLL_US (10) microseconds — the consumer thread's CPU 1.28 -> 0.81 ms per fetch at the
    // same rate (profiles/r06_e2e_wait_modes.txt).  SURGE_INGEST_WAIT=block sleeps on the blocking event, =spin keeps
    // hipStreamSynchronize.
    d->block_waits = true;
    d->poll_ns = 10000;
    if (const char* v = std::getenv("SURGE_INGEST_WAIT")) {
      d->block_waits = std::strcmp(v, "spin") != 0;
      if (std::strcmp(v, "poll") != 0) d->poll_ns = 0;
    }
    if (d->poll_ns > 0)
      if (const char* u = std::getenv("SURGE_INGEST_POLL_US")) d->poll_ns = std::atoll(u) > 0 ? std::atoll(u) * 1000 : d->poll_ns;
    if (d->block_waits) DCHK(d, hipEventCreateWithFlags(&d->sleeper, hipEventDisableTiming | hipEventBlockingSync));
    DCHK(d, d->key_off.reserve(8, false, d->stream));
    DCHK(d, hipMemset(d->key_off.p, 0, 8));
    if (tmpl) {
      d->h_tmpl = *tmpl;
      // the distinct field names, each once, and per type which of them it reads
      static EvjDevice ev;  // (a few KB of names: off the stack; create is not a hot path)
      static std::mutex ev_mu;
      std::lock_guard<std::mutex> lk(ev_mu);
      std::memset(&ev, 0, sizeof(ev));
      ev.n_types = tmpl->n_types;
      auto intern = [&](const char* name) -> uint8_t {
        if (!name[0]) return 0xff;
        for (uint32_t j = 1; j < ev.n_names; ++j)
          if (std::strncmp(ev.names[j], name, SURGE_EVJ_NAME) == 0) return (uint8_t)j;
        std::memcpy(ev.names[ev.n_names], name, SURGE_EVJ_NAME);
        ev.name_len[ev.n_names] = (uint8_t)strnlen(name, SURGE_EVJ_NAME);
        return (uint8_t)ev.n_names++;
      };
      std::memcpy(ev.names[0], tmpl->discriminator, SURGE_EVJ_NAME);
      ev.name_len[0] = (uint8_t)strnlen(tmpl->discriminator, SURGE_EVJ_NAME);
      ev.n_names = 1;
      for (uint32_t i = 0; i < tmpl->n_types; ++i) {
        const surge_event_json_type& ty = tmpl->types[i];
        std::memcpy(ev.type_name[i], ty.name, SURGE_EVJ_NAME);
        ev.type_name_len[i] = (uint8_t)strnlen(ty.name, SURGE_EVJ_NAME);
        ev.seq_name[i] = intern(ty.seq_field);
        ev.arg_name[i] = intern(ty.arg_field);
        ev.event_type[i] = ty.event_type;
        ev.arg_kind[i] = ty.arg_kind;
      }
      if (ev.n_names - 1 > (uint32_t)kEvjTrack)
        return dfail(d, E_UNSUPPORTED, "the event template names more than 8 distinct sequence / argument fields (decode this topic with surge_ingest_drain_json)");
      DCHK(d, d->d_tmpl.reserve(sizeof(ev), false, d->stream));
      DCHK(d, hipMemcpy(d->d_tmpl.p, &ev, sizeof(ev), hipMemcpyHostToDevice));
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
#ifdef SURGE_EXPERIMENTS
  std::fprintf(stderr, "[surge experiments] decoder: %lld pushes, %lld batches chained by the one-lane walk after the parallel recognition declined\n", (long long)d->pushes,
               (long long)d->chain_fallbacks);
#endif
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(d->device);
  (void)hipStreamSynchronize(d->stream);
  for (int i = 0; i < d->n_push_streams; ++i) {
    (void)hipStreamSynchronize(d->push_streams[i]);
    (void)hipStreamDestroy(d->push_streams[i]);
  }
  for (PushSlot& s : d->slots) {
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.released) (void)hipEventDestroy(s.released);
    Buf* sb[] = {&s.lz4_blocks, &s.lz4_sizes, &s.lz4_nseq, &s.lz4_seq, &s.lz4_cls, &s.d_bytes, &s.d_sections, &s.rec_a, &s.rec_b, &s.rec_c, &s.meta, &s.ev_tmp,
                 &s.f64_list, &s.d_err, &s.crc_spans};
    for (Buf* b : sb) b->release();
    if (s.pinned) (void)hipHostFree(s.pinned);
  }
  Buf* bufs[] = {&d->d_tmpl, &d->d_ptab, &d->first, &d->first_scan, &d->keep, &d->keep_pos, &d->temp, &d->t_slots,
                 &d->arena, &d->key_off, &d->key_hash, &d->r_agg, &d->r_ev, &d->r_off};
  for (Buf* b : bufs) b->release();
  if (d->ready) (void)hipEventDestroy(d->ready);
  if (d->consumed) (void)hipEventDestroy(d->consumed);
  if (d->sleeper) (void)hipEventDestroy(d->sleeper);
  if (d->crc_slice) (void)hipFree(d->crc_slice);
  (void)hipSetDevice(prev);
  delete d;
  return OK;
}

}  // extern "C" (reopened below)

namespace {

const char* why_bad(uint32_t status) {
  static const char* why[] = {"", "", "has a null key or value (not an event)", "is malformed (a length runs past its record or batch)",
                              "is not the JSON object the event template describes", "names an event type the template does not know",
                              "lacks a field the template names, or the field is not the number it should be", "is not a 16-byte fixed event", "",
                              "collides with another key on its 64-bit hash"};
  return status < 10 ? why[status] : "is bad";
}

struct DeviceScope {  // the calling thread's device, restored on the way out
  int prev = 0;
  explicit DeviceScope(int dev) { (void)hipGetDevice(&prev); (void)hipSetDevice(dev); }
  ~DeviceScope() { (void)hipSetDevice(prev); }
};

// the slot a new push's stage 1 goes into (nullptr with the error set: every slot holds an unfinished push)
PushSlot* claim_slot(surge_device_decoder* d, int32_t* rc) {
  if (d->poisoned) {
    *rc = dfail(d, SURGE_E_STATE, "an earlier push failed on the device half way: destroy this decoder and create a new one");
    return nullptr;
  }
  PushSlot* s = nullptr;
  {
    std::lock_guard<std::mutex> lk(d->mu);
    if (d->n_pending < kSlots) s = &d->slots[(d->head + d->n_pending) % kSlots];  // (a finish on the other thread moves head and n_pending together: the same slot)
    if (s) s->stream = d->push_streams[d->push_seq++ % (uint64_t)d->n_push_active];
  }
  if (!s) *rc = dfail(d, SURGE_E_STATE, "every push slot holds an unfinished push: call surge_device_decoder_push_finish first");
  return s;
}

// the slot's last push was finished without waiting for the device (push_finish_async): its buffers are free once the
// decoder's stream has passed the end of that stage 2
int32_t await_release(surge_device_decoder* d, PushSlot& s) {
  if (!s.released_valid) return OK;
  DCHK(d, hipStreamWaitEvent(s.stream, s.released, 0));
  s.released_valid = false;
  return OK;
}

int32_t slot_scratch(surge_device_decoder* d, PushSlot& s, int64_t n_rec) {
  const size_t R = (size_t)n_rec;
  DCHK(d, s.meta.reserve(R * sizeof(RecMeta), false, s.stream));
  DCHK(d, s.ev_tmp.reserve(R * 16, false, s.stream));
  DCHK(d, s.f64_list.reserve(R * 4, false, s.stream));
  static const ErrorCell kZero{~0ull, 0u, 0u, ~0u, ~0u};  // (the source of an asynchronous copy: it must outlive the call)
  DCHK(d, hipMemcpyAsync(s.d_err.p, &kZero, sizeof(kZero), hipMemcpyHostToDevice, s.stream));
  return OK;
}

int32_t slot_pinned(surge_device_decoder* d, PushSlot& s, size_t bytes) {
  if (bytes <= s.pinned_cap) return OK;
  if (s.pinned) (void)hipHostFree(s.pinned);
  s.pinned = nullptr;
  s.pinned_cap = 0;
  bytes += bytes / 4 + 65536;  // (room to spare: the next fetch's tables are a few per cent larger or smaller)
  DCHK(d, hipHostMalloc(&s.pinned, bytes, hipHostMallocDefault));
  s.pinned_cap = bytes;
  return OK;
}

// SURGE_DBG_DECODE=<lz4 mode><section mode> (two digits; timing experiments only — see Lz4Work::dbg / JsonCtx::dbg; a
// lz4 mode needs a section mode, since the sections' bytes are then not the topic's)
int32_t dbg_decode() {
#ifdef SURGE_EXPERIMENTS  // this hook makes the decoder deliver records that are NOT the topic's: experiment builds of the library only
  static const int32_t v = [] {
    const char* e = std::getenv("SURGE_DBG_DECODE");
    const int32_t x = e ? std::atoi(e) : 0;
    return (x >= 10 && x % 10 == 0) ? x + 1 : x;
  }();
  return v;
#else
  return 0;
#endif
}

// Stage 1 of a wire push: the parts' records sections to the device, LZ4 blocks decoded, every record chained, parsed and
// its value decoded.  Nothing here reads or writes the key table.
int32_t stage1_wire(surge_device_decoder* d, PushSlot& s, int32_t n_parts, const uint8_t* const* bytes, const surge_batch_section* const* sections,
                    const int64_t* n_sections) {
  static const bool dbg_t = std::getenv("SURGE_DBG_TIMING") != nullptr;
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = dbg_t ? now_us() : 0.0;
  auto lap = [&](const char* what) {
    if (!dbg_t) return;
    const double t = now_us();
    std::fprintf(stderr, "[surge dbg] stage1 %-14s %8.1f us\n", what, t - t_mark);
    t_mark = t;
  };
  const uint64_t seed = d->seed.load();
  int64_t total_sections = 0;
  for (int32_t p = 0; p < n_parts; ++p) {
    if (n_sections[p] < 0 || (n_sections[p] > 0 && (!bytes[p] || !sections[p]))) return dfail(d, E_INVALID, "bad argument");
    total_sections += n_sections[p];
  }
  s.n_rec = 0;
  s.wire = true;
  if (total_sections == 0) return OK;
  if (total_sections >= (1ll << 31)) return dfail(d, E_UNSUPPORTED, "more than 2^31 batches in one push");
  std::vector<Section>& secs = s.h_secs;
  std::vector<Lz4Block>& blocks = s.h_blocks;
  std::vector<uint8_t> extra;
  std::vector<int64_t> part_lo((size_t)n_parts, 0), part_dev((size_t)n_parts, 0), part_len((size_t)n_parts, 0);
  int64_t n_rec = 0, n_raw = 0;
  int32_t max_recs = 0;  // of one batch of this push (section_kernel's workgroup size)
  try {
    secs.resize((size_t)total_sections);
    blocks.clear();
    s.h_crc.clear();
    // the span of each part's arena this push needs, laid out one after the other (16-byte aligned) on the device
    int64_t at = 0;
    for (int32_t p = 0; p < n_parts; ++p) {
      int64_t lo = INT64_MAX, hi = 0;
      for (int64_t i = 0; i < n_sections[p]; ++i) {
        const surge_batch_section& in = sections[p][i];
        if (in.byte_off < 0 || in.byte_len < 0 || in.n_records < 0) return dfail(d, E_INVALID, "negative section field");
        const int64_t crc_prefix = (in.codec & SURGE_SECTION_CRC_PENDING) ? 8 : (in.codec & SURGE_SECTION_CRC_WIRE) ? 44 : 0;
        if (crc_prefix && (in.byte_off < crc_prefix || in.byte_len >= (1ll << 31) - 64)) return dfail(d, E_INVALID, "a CRC-pending section without the bytes in front of it");
        lo = in.byte_off - crc_prefix < lo ? in.byte_off - crc_prefix : lo;
        hi = in.byte_off + in.byte_len > hi ? in.byte_off + in.byte_len : hi;
      }
      if (n_sections[p] == 0) lo = hi = 0;
      part_lo[(size_t)p] = lo;
      part_len[(size_t)p] = hi - lo;
      part_dev[(size_t)p] = n_raw;
      n_raw = (n_raw + (hi - lo) + 15) & ~15ll;
      for (int64_t i = 0; i < n_sections[p]; ++i, ++at) {
        const surge_batch_section& in = sections[p][i];
        secs[(size_t)at] = Section{part_dev[(size_t)p] + (in.byte_off - lo), in.byte_len, in.base_offset, in.n_records, 0, n_rec};
        n_rec += in.n_records;
        max_recs = in.n_records > max_recs ? in.n_records : max_recs;
      }
    }
  } catch (const std::bad_alloc&) {
    return dfail(d, E_NOMEM, "out of host memory");
  }
  if (n_rec == 0) return OK;
  if (n_rec >= (1ll << 32) - 1) return dfail(d, E_UNSUPPORTED, "more than 2^32 - 2 records in one push: push fewer sections at a time");
  // LZ4 sections (codec 3: the batch's records section is still one LZ4 frame): the host reads the frame header and the
  // block size words, the device decodes the blocks.  Frames it cannot take block by block (blocks larger than 64 KiB,
  // dependent blocks) are decompressed here, on the host, and travel as plain bytes behind the raw spans.
  int64_t area = 0;           // bytes of the device-side decompressed area handed out so far (multiples of 64 KiB)
  int64_t n_seq_entries = 0;  // sequence-table entries handed out (two-pass decode)
  bool any_one_pass = false;
  static const bool force_one_pass = [] { const char* v = std::getenv("SURGE_INGEST_LZ4_ONEPASS"); return v && v[0] == '1'; }();
  try {
    int64_t at = 0;
    for (int32_t p = 0; p < n_parts; ++p) {
      for (int64_t i = 0; i < n_sections[p]; ++i, ++at) {
        const surge_batch_section& in = sections[p][i];
        // (the frame headers were written by the framing threads, on other cores: one cache miss per batch — 1.2 ms per
        // 7000-batch push — unless they are asked for ahead of time)
        // ... the lines this walk reads of a section: its frame header (and, framed in place, the crc field 44 bytes in front of
        // it), and the frame's last words — the EndMark behind its last block
        if (i + 16 < n_sections[p]) {
          const surge_batch_section& nx = sections[p][i + 16];
          __builtin_prefetch(bytes[p] + nx.byte_off - ((nx.codec & SURGE_SECTION_CRC_WIRE) ? 44 : 0));
          __builtin_prefetch(bytes[p] + nx.byte_off + 16);
          if (nx.byte_len > 64) __builtin_prefetch(bytes[p] + nx.byte_off + nx.byte_len - 8);
        }
        Section& sec = secs[(size_t)at];
        if (in.codec & SURGE_SECTION_CRC_PENDING) {  // {crc, register after the header bytes}, written by the framer in front of the section
          uint32_t pre[2];
          std::memcpy(pre, bytes[p] + in.byte_off - 8, 8);
          s.h_crc.push_back(CrcSpan{sec.byte_off, (int32_t)in.byte_len, (int32_t)at, pre[0], pre[1]});
        } else if (in.codec & SURGE_SECTION_CRC_WIRE) {  // the batch as received: its crc field (big-endian), then the 40 header bytes it covers, then the section
          const uint8_t* c = bytes[p] + in.byte_off - 44;
          const uint32_t expect = ((uint32_t)c[0] << 24) | ((uint32_t)c[1] << 16) | ((uint32_t)c[2] << 8) | c[3];
          s.h_crc.push_back(CrcSpan{sec.byte_off - 40, (int32_t)in.byte_len + 40, (int32_t)at, expect, ~0u});
        }
        if ((in.codec & 0xff) != 3 || in.n_records == 0) continue;
        const uint8_t* f = bytes[p] + in.byte_off;
        const int64_t fl = in.byte_len;
        bool device_ok = fl >= 7 && f[0] == 0x04 && f[1] == 0x22 && f[2] == 0x4D && f[3] == 0x18 && (f[4] >> 6) == 1 && (f[4] & 0x20) &&
                         ((f[5] >> 4) & 7) == 4;
        int64_t q = 6;
        if (device_ok) {
          if (f[4] & 0x08) q += 8;  // content size
          if (f[4] & 0x01) q += 4;  // dictionary id
          device_ok = q < fl && f[q] == (uint8_t)(surge_xxh32(f + 4, q - 4, 0) >> 8);
          ++q;
        }
        const size_t first_block = blocks.size();
        const int64_t seq_mark = n_seq_entries;
        int32_t k = 0;
        while (device_ok) {
          if (fl - q < 4) { device_ok = false; break; }
          const uint32_t bs = (uint32_t)f[q] | ((uint32_t)f[q + 1] << 8) | ((uint32_t)f[q + 2] << 16) | ((uint32_t)f[q + 3] << 24);
          q += 4;
          if (bs == 0) break;  // EndMark
          const uint32_t size = bs & 0x7fffffffu;
          if ((int64_t)size > fl - q || size > (1u << 30)) { device_ok = false; break; }
          Lz4Block b;
          b.src_off = sec.byte_off + q;
          b.dst_off = area + (int64_t)k * kLz4BlockMax;
          b.src_len = (int32_t)size | (int32_t)(bs & 0x80000000u);
          b.section = (int32_t)at;
          b.last = 0;
          b.index = k++;
          // a sequence with a match takes at least 3 bytes of the block, the closing literal run at least 1
          if (!force_one_pass && size < (uint32_t)kLz4BlockMax) {
            b.seq_off = n_seq_entries;
            n_seq_entries += (bs & 0x80000000u) ? 0 : (int64_t)size / 3 + 2;
          } else {
            b.seq_off = -1;
            any_one_pass = true;
          }
          blocks.push_back(b);
          q += size;
          if (f[4] & 0x10) q += 4;  // block checksum (not verified: the batch CRC already covers these bytes)
        }
        if (device_ok && k > 0) {
          blocks.back().last = 1;
          sec.byte_off = -1 - area;  // resolved below, once the raw spans' final size is known
          sec.byte_len = 0;          // set by the kernel that decodes the frame's last block
          area += (int64_t)k * kLz4BlockMax;
        } else {
          blocks.resize(first_block);
          n_seq_entries = seq_mark;
          int64_t cap = fl * 8 + 1024, got;
          const size_t ex = extra.size();
          while (true) {
            extra.resize(ex + (size_t)cap);
            got = surge_lz4_frame_decompress(f, fl, extra.data() + ex, cap);
            if (got != -6) break;
            cap *= 4;
            if (cap > (1ll << 31)) return dfail(d, SURGE_E_CORRUPT, "LZ4 batch expands beyond 2 GiB");
          }
          if (got < 0) return dfail(d, SURGE_E_CORRUPT, "bad LZ4 frame in the section at base offset " + std::to_string(in.base_offset));
          extra.resize(ex + (size_t)got);
          sec.byte_off = n_raw + (int64_t)ex;
          sec.byte_len = got;
        }
      }
    }
    any_one_pass = false;
    for (const Lz4Block& b : blocks) any_one_pass = any_one_pass || b.seq_off < 0;
  } catch (const std::bad_alloc&) {
    return dfail(d, E_NOMEM, "out of host memory");
  }
  lap("walk sections");
  const int64_t n_bytes = n_raw + (int64_t)extra.size();              // what is staged and copied
  const int64_t area_base = (n_bytes + 15) & ~15ll;                    // where the device-decompressed frames start
  for (Section& sc : secs)
    if (sc.byte_off < 0) sc.byte_off = area_base + (-1 - sc.byte_off);
  hipStream_t st = s.stream;
  const size_t sec_bytes = sizeof(Section) * (size_t)total_sections, blk_bytes = sizeof(Lz4Block) * blocks.size();
  // The device buffers of this push.  The FIRST wire push of a decoder sizes every slot like its own: the fetches of a
  // recovery are alike, and a slot that sizes its buffers when its turn comes does so in the middle of the pipeline (an
  // allocation per buffer, and a hipFree — a device-wide wait — for every one that grows).
  auto size_slot = [&](PushSlot& t) -> int32_t {
    DCHK(d, t.d_bytes.reserve((size_t)(area_base + area) + 64, false, t.stream));
    DCHK(d, t.d_sections.reserve(sec_bytes, false, t.stream));
    DCHK(d, t.meta.reserve((size_t)n_rec * sizeof(RecMeta), false, t.stream));
    DCHK(d, t.ev_tmp.reserve((size_t)n_rec * 16, false, t.stream));
    DCHK(d, t.f64_list.reserve((size_t)n_rec * 4, false, t.stream));
    if (!blocks.empty()) {
      const size_t nb = blocks.size();
      DCHK(d, t.lz4_blocks.reserve(blk_bytes, false, t.stream));
      DCHK(d, t.lz4_sizes.reserve(nb * 4, false, t.stream));
      if (!force_one_pass) {
        DCHK(d, t.lz4_nseq.reserve(nb * 4, false, t.stream));
        DCHK(d, t.lz4_seq.reserve((size_t)(n_seq_entries + 1) * 8, false, t.stream));
        DCHK(d, t.lz4_cls.reserve((size_t)(kLz4Classes + 1) * (nb + 1) * 4, false, t.stream));
      }
    }
    return OK;
  };
  if (!d->slots_sized) {
    d->slots_sized = true;
    // (only while nothing else is in flight — the decoder's first push as a rule: no other slot is then being read by a push or by
    // the thread that finishes pushes; a slot whose last push was finished without a wait is left alone too)
    bool idle;
    {
      std::lock_guard<std::mutex> lk(d->mu);
      idle = d->n_pending == 0;
    }
    if (idle)
      for (PushSlot& t : d->slots)
        if (&t != &s && !t.busy && !t.released_valid && size_slot(t) != OK) (void)hipGetLastError();  // (best effort: a slot that could not be sized now reports it when its turn comes)
  }
  {
    int32_t rc = size_slot(s);
    if (rc == OK) rc = slot_scratch(d, s, n_rec);
    if (rc != OK) return rc;
  }
  // H2D: a part goes straight out of the caller's arena when that is page-locked (surge_ingest_use_pinned_arena), else
  // through the slot's own pinned staging (one extra host copy)
  lap("reserve");
  // the section and block tables travel through page-locked staging too: a copy from pageable memory waits for the stream
  // (1.4 ms of host time per push went there)
  size_t need_stage = ((extra.size() + 15) & ~(size_t)15) + sec_bytes + blk_bytes + ((s.h_crc.size() * sizeof(CrcSpan) + 15) & ~(size_t)15) + 64;
  std::vector<char> in_place((size_t)n_parts, 0);
  for (int32_t p = 0; p < n_parts; ++p) {
    if (part_len[(size_t)p] == 0) continue;
    hipPointerAttribute_t attr;
    in_place[(size_t)p] = hipPointerGetAttributes(&attr, bytes[p] + part_lo[(size_t)p]) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();  // a pageable pointer makes hipPointerGetAttributes fail: not an error of this call
    if (!in_place[(size_t)p]) need_stage += ((size_t)part_len[(size_t)p] + 15) & ~(size_t)15;
  }
  {
    const bool first_staging = s.pinned_cap == 0;
    const int32_t rc = slot_pinned(d, s, need_stage);
    if (rc != OK) return rc;
    if (first_staging) {  // (page-locking memory takes milliseconds: every slot's staging while the pipeline is still empty)
      bool idle;
      {
        std::lock_guard<std::mutex> lk(d->mu);
        idle = d->n_pending == 0;
      }
      if (idle)
        for (PushSlot& t : d->slots)
          if (&t != &s && !t.busy && !t.released_valid && t.pinned_cap == 0) (void)slot_pinned(d, t, need_stage);
    }
  }
  size_t staged = 0;
  auto stage = [&](const void* src, size_t len) -> const uint8_t* {
    uint8_t* at = (uint8_t*)s.pinned + staged;
    std::memcpy(at, src, len);
    staged += (len + 15) & ~(size_t)15;
    return at;
  };
  for (int32_t p = 0; p < n_parts; ++p) {
    const size_t len = (size_t)part_len[(size_t)p];
    if (len == 0) continue;
    const uint8_t* src = bytes[p] + part_lo[(size_t)p];
    if (!in_place[(size_t)p]) src = stage(src, len);
    DCHK(d, hipMemcpyAsync((uint8_t*)s.d_bytes.p + part_dev[(size_t)p], src, len, hipMemcpyHostToDevice, st));
  }
  if (!extra.empty()) DCHK(d, hipMemcpyAsync((uint8_t*)s.d_bytes.p + n_raw, stage(extra.data(), extra.size()), extra.size(), hipMemcpyHostToDevice, st));
  DCHK(d, hipMemcpyAsync(s.d_sections.p, stage(secs.data(), sec_bytes), sec_bytes, hipMemcpyHostToDevice, st));
  if (!s.h_crc.empty()) {
    // the batches' CRC-32C, finished where their bytes now are (the host ran it over the 40 header bytes only)
    if (!d->crc_slice) {
      DCHK(d, hipMalloc(&d->crc_slice, sizeof(CrcTables)));
      DCHK(d, hipMemcpy(d->crc_slice, &crc_tables(), sizeof(CrcTables), hipMemcpyHostToDevice));
    }
    const size_t crc_bytes = s.h_crc.size() * sizeof(CrcSpan);
    DCHK(d, s.crc_spans.reserve(crc_bytes, false, st));
    DCHK(d, hipMemcpyAsync(s.crc_spans.p, stage(s.h_crc.data(), crc_bytes), crc_bytes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(crc_kernel, dim3((unsigned)((s.h_crc.size() + kCrcWaves - 1) / kCrcWaves)), dim3(64 * kCrcWaves), 0, st, (const uint8_t*)s.d_bytes.p,
                       (const CrcSpan*)s.crc_spans.p, (int32_t)s.h_crc.size(), (const CrcTables*)d->crc_slice, (ErrorCell*)s.d_err.p);
  }
  lap("copies");
  const uint8_t* dby = (const uint8_t*)s.d_bytes.p;
  Section* dsec = (Section*)s.d_sections.p;
  ErrorCell* derr = (ErrorCell*)s.d_err.p;
  RecMeta* dmeta = (RecMeta*)s.meta.p;
  if (!blocks.empty()) {
    DCHK(d, s.lz4_blocks.reserve(blk_bytes, false, st));
    DCHK(d, hipMemcpyAsync(s.lz4_blocks.p, stage(blocks.data(), blk_bytes), blk_bytes, hipMemcpyHostToDevice, st));
    DCHK(d, s.lz4_sizes.reserve(blocks.size() * 4, false, st));
    const int64_t nb = (int64_t)blocks.size();
    if (nb >= (1ll << 31)) return dfail(d, E_UNSUPPORTED, "more than 2^31 LZ4 blocks in one push: push fewer sections at a time");
    if (!force_one_pass) {
      // two passes: the sequence headers by one lane per block, then the copies by one wave per block in a launch with
      // the LDS the block's size needs (the first pass knows it)
      Lz4Work w;
      w.dbg = dbg_decode() / 10;
      static const int32_t lz4_pad = [] { const char* v = std::getenv("SURGE_INGEST_LZ4_PAD"); return v ? (std::atoi(v) & ~15) : 0; }();
      w.pad = lz4_pad < 0 ? 0 : (lz4_pad > 4096 ? 4096 : lz4_pad);
      DCHK(d, s.lz4_nseq.reserve((size_t)nb * 4, false, st));
      DCHK(d, s.lz4_seq.reserve((size_t)(n_seq_entries + 1) * 8, false, st));
      DCHK(d, s.lz4_cls.reserve((size_t)(kLz4Classes + 1) * ((size_t)nb + 1) * 4, false, st));
      w.state = (int32_t*)s.lz4_sizes.p;
      w.n_seq = (int32_t*)s.lz4_nseq.p;
      w.seq = (uint2*)s.lz4_seq.p;
      w.cls_count = (int32_t*)s.lz4_cls.p;
      w.cls_list = w.cls_count + (kLz4Classes + 1);
      DCHK(d, hipMemsetAsync(w.cls_count, 0, (kLz4Classes + 1) * 4, st));
      {
        // LDS of the launches of the first pass.  A block of the 16 KiB producer compresses to 3 - 5.5 KiB: 6.5 KiB of LDS is what
        // 24 waves per CU — all the kernel's 78 VGPRs allow — leave each other (8 KiB: 20 waves); a batch that compresses badly
        // takes the 16 KiB class (9 waves per CU) instead of sharing a CU with one other wave in the 64 KiB one.
        static const bool two_launches = [] { const char* v = std::getenv("SURGE_INGEST_LZ4_PARSE_CLASSES"); return v && v[0] == '2'; }();  // experiments: round 4's {8 KiB, 64 KiB}
        const int32_t caps3[3] = {6656, 16384 + 64, kLz4BlockMax + 64}, caps2[2] = {8192, kLz4BlockMax + 64};
        const int32_t* caps = two_launches ? caps2 : caps3;
        const int n_caps = two_launches ? 2 : 3;
        for (int c = 0; c < n_caps; ++c)
          hipLaunchKernelGGL(lz4_parse_kernel, dim3((unsigned)nb), dim3(64), (size_t)caps[c], st, dby, (uint8_t*)s.d_bytes.p + area_base, (const Lz4Block*)s.lz4_blocks.p,
                             (int32_t)nb, w, dsec, derr, c == 0 ? -1 : caps[c - 1], caps[c]);
      }
      for (int c = 0; c <= kLz4Classes; ++c) {
        const int32_t cap = c < kLz4Classes ? kLz4ClassCapHost[c] : 0;
        const size_t lds = cap ? (size_t)cap + (size_t)w.pad + (size_t)kLz4Map * 2 : 0;
        const int64_t resident = 256ll * (lds ? (int64_t)(160 * 1024 / lds) : 16);  // waves the chip holds at this LDS size
        const unsigned grid = (unsigned)(nb < resident ? nb : resident);
        static const bool one_window = [] { const char* v = std::getenv("SURGE_INGEST_LZ4_WINDOWS"); return v && v[0] == '1'; }();
        if (one_window)
          hipLaunchKernelGGL(lz4_exec_kernel<false>, dim3(grid), dim3(64), lds, st, dby, (uint8_t*)s.d_bytes.p + area_base, (const Lz4Block*)s.lz4_blocks.p, (int32_t)nb, w, c, cap);
        else
          hipLaunchKernelGGL(lz4_exec_kernel<true>, dim3(grid), dim3(64), lds, st, dby, (uint8_t*)s.d_bytes.p + area_base, (const Lz4Block*)s.lz4_blocks.p, (int32_t)nb, w, c, cap);
      }
    }
    if (any_one_pass) {
      const unsigned grid = (unsigned)(nb < 8192 ? nb : 8192);
      const int32_t caps[3] = {16384, 32768, kLz4BlockMax};  // LDS per wave of the three launches
      for (int c = 0; c < 3; ++c)
        hipLaunchKernelGGL(lz4_block_kernel, dim3(grid), dim3(64), (size_t)caps[c], st, dby, (uint8_t*)s.d_bytes.p + area_base,
                           (const Lz4Block*)s.lz4_blocks.p, nb, (int32_t*)s.lz4_sizes.p, c, caps[c], dsec, derr);
    }
    // (a frame that does not decode zeroes its section and raises the error cell: the push fails in stage 2's first
    // synchronisation, before anything is committed)
  }
  lap("lz4 launches");
  // chain + parse + decode, one workgroup per batch: sections up to 16.25 KiB (the reference producer closes a batch at 16 KiB)
  // out of 18.3 KiB of LDS (8 workgroups per CU), the rest out of 66 KiB or, beyond 64 KiB, in place
  // (round 5: a third, 8 KiB class in front — a publisher that flushes every 64 events writes 7 KiB batches, and 10.4 KiB of LDS
  // instead of 18.3 lets a CU hold 15 of them; the workgroup is as wide as the push's largest batch needs)
  {
    JsonCtx jc{d->json ? (const EvjDevice*)d->d_tmpl.p : nullptr, (const surge::F64ParseTable*)d->d_ptab.p};
    jc.dbg = dbg_decode() % 10;
    static const bool lane_chain = [] { const char* v = std::getenv("SURGE_INGEST_CHAIN"); return v && v[0] == 'l'; }();  // A/B: "lane" = the one-lane walk of rounds 4 / 5
    jc.wave_chain = lane_chain ? 0 : 1;
    static const int force_threads = [] { const char* v = std::getenv("SURGE_INGEST_SEC_THREADS"); return v ? std::atoi(v) : 0; }();  // experiments: 64 / 128 / 192 / 256
    static const bool two_classes = [] { const char* v = std::getenv("SURGE_INGEST_SEC_CLASSES"); return v && v[0] == '2'; }();      // experiments: round 4's two launches
    const int threads = force_threads ? force_threads : (max_recs <= 64 ? 64 : max_recs <= 128 ? 128 : max_recs <= 192 ? 192 : 256);
    const int64_t caps3[3] = {8320, 16640, 65536};
    const int64_t* caps = two_classes ? caps3 + 1 : caps3;
    const int n_caps = two_classes ? 2 : 3;
#ifdef SURGE_EXPERIMENTS
    static unsigned long long* tick_buf = nullptr;
    static const bool want_ticks = std::getenv("SURGE_SECTION_TICKS") != nullptr;
    if (want_ticks && !tick_buf) (void)hipMalloc(&tick_buf, 4ull * 8 * 65536);
#endif
    for (int c = 0; c < n_caps; ++c) {
#ifdef SURGE_EXPERIMENTS
      if (want_ticks && total_sections <= 65536 && c == 1) {
        (void)hipMemsetAsync(tick_buf, 0, 4ull * 8 * 65536, st);
        jc.ticks = tick_buf;
      } else {
        jc.ticks = nullptr;
      }
#endif
      const size_t lds = (size_t)((caps[c] + 47) & ~15ll) + 2 * (size_t)kSecRecs * 4;
      const int64_t lo_excl = c == 0 ? -1 : caps[c - 1];
      const int32_t rest = c == n_caps - 1 ? 1 : 0;
#define SURGE_SECTION_LAUNCH(T)                                                                                                                       \
  hipLaunchKernelGGL(section_kernel<T>, dim3((unsigned)total_sections), dim3(T), lds, st, dby, (const Section*)dsec, total_sections, lo_excl, caps[c], rest, seed, \
                     jc, dmeta, (uint4*)s.ev_tmp.p, (uint32_t*)s.f64_list.p, derr)
      if (threads <= 64) SURGE_SECTION_LAUNCH(64);
      else if (threads <= 128) SURGE_SECTION_LAUNCH(128);
      else if (threads <= 192) SURGE_SECTION_LAUNCH(192);
      else SURGE_SECTION_LAUNCH(256);
#undef SURGE_SECTION_LAUNCH
#ifdef SURGE_EXPERIMENTS
      if (jc.ticks) {  // (blocks: experiment runs only)
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> t(4ull * (size_t)total_sections);
        (void)hipMemcpy(t.data(), tick_buf, t.size() * 8, hipMemcpyDeviceToHost);
        double sum[3] = {0, 0, 0};
        unsigned long long first = ~0ull, last = 0;
        size_t n = 0;
        for (size_t b = 0; b < (size_t)total_sections; ++b) {
          const unsigned long long* q = &t[4 * b];
          if (!q[0] || !q[3]) continue;
          ++n;
          for (int k = 0; k < 3; ++k) sum[k] += (double)(q[k + 1] - q[k]);
          first = q[0] < first ? q[0] : first;
          last = q[3] > last ? q[3] : last;
        }
        if (n) std::fprintf(stderr, "[surge experiments] section launch (class 2): %zu workgroups with work, mean us: staging %.2f, chain %.2f, decode %.2f; first start to last end %.1f us\n", n,
                            sum[0] / n / 100.0, sum[1] / n / 100.0, sum[2] / n / 100.0, (double)(last - first) / 100.0);
      }
#endif
    }
    DCHK(d, hipGetLastError());
  }
  lap("section launches");
  s.n_rec = n_rec;
  s.seed = seed;
  return OK;
}

// Stage 1 of a push of records that are already framed
int32_t stage1_records(surge_device_decoder* d, PushSlot& s, const uint8_t* keys, const int64_t* key_off, const uint8_t* values, const int64_t* value_off,
                       const int64_t* offsets, int64_t n) {
  s.n_rec = 0;
  s.wire = false;
  const uint64_t seed = d->seed.load();
  if (n == 0) return OK;
  if (n >= (1ll << 32) - 1) return dfail(d, E_UNSUPPORTED, "more than 2^32 - 2 records in one push");
  const int64_t kb = key_off[n] - key_off[0], vb = value_off[n] - value_off[0];
  if (kb < 0 || vb < 0 || (kb > 0 && !keys) || (vb > 0 && !values)) return dfail(d, E_INVALID, "bad key / value spans");
  for (int64_t i = 0; i < n; ++i)  // (the kernels index the staged bytes with these: no launch on offsets that run backwards)
    if (key_off[i + 1] < key_off[i] || value_off[i + 1] < value_off[i]) return dfail(d, E_INVALID, "key_off / value_off must not decrease (record " + std::to_string(i) + ")");
  hipStream_t st = s.stream;
  const size_t n_bytes = (size_t)(kb + vb), off_bytes = (size_t)(n + 1) * 8;
  const size_t stage = n_bytes + 2 * off_bytes + (offsets ? (size_t)n * 8 : 0) + 64;
  DCHK(d, s.d_bytes.reserve(n_bytes + 16, false, st));
  DCHK(d, s.rec_a.reserve(off_bytes, false, st));  // the device copies of key_off / value_off / offsets
  DCHK(d, s.rec_b.reserve(off_bytes, false, st));
  DCHK(d, s.rec_c.reserve((size_t)n * 8 + 8, false, st));
  {
    int32_t rc = slot_scratch(d, s, n);
    if (rc == OK) rc = slot_pinned(d, s, stage);
    if (rc != OK) return rc;
  }
  // pinned staging: [keys][values][key_off (rebased)][value_off (rebased)][offsets]
  uint8_t* pin = (uint8_t*)s.pinned;
  if (kb) std::memcpy(pin, keys + key_off[0], (size_t)kb);
  if (vb) std::memcpy(pin + kb, values + value_off[0], (size_t)vb);
  int64_t* p_ko = (int64_t*)(pin + ((n_bytes + 7) & ~(size_t)7));
  int64_t* p_vo = p_ko + (n + 1);
  int64_t* p_of = p_vo + (n + 1);
  for (int64_t i = 0; i <= n; ++i) { p_ko[i] = key_off[i] - key_off[0]; p_vo[i] = value_off[i] - value_off[0]; }
  if (offsets) std::memcpy(p_of, offsets, (size_t)n * 8);
  if (n_bytes) DCHK(d, hipMemcpyAsync(s.d_bytes.p, pin, n_bytes, hipMemcpyHostToDevice, st));
  DCHK(d, hipMemcpyAsync(s.rec_a.p, p_ko, off_bytes, hipMemcpyHostToDevice, st));
  DCHK(d, hipMemcpyAsync(s.rec_b.p, p_vo, off_bytes, hipMemcpyHostToDevice, st));
  if (offsets) DCHK(d, hipMemcpyAsync(s.rec_c.p, p_of, (size_t)n * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(records_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t*)s.d_bytes.p, (const int64_t*)s.rec_a.p,
                     (const int64_t*)s.rec_b.p, offsets ? (const int64_t*)s.rec_c.p : nullptr, kb, n, seed,
                     JsonCtx{d->json ? (const EvjDevice*)d->d_tmpl.p : nullptr, (const surge::F64ParseTable*)d->d_ptab.p},
                     (RecMeta*)s.meta.p, (uint4*)s.ev_tmp.p, (uint32_t*)s.f64_list.p, (ErrorCell*)s.d_err.p);
  DCHK(d, hipGetLastError());
  s.n_rec = n;
  s.seed = seed;
  return OK;
}

// a consumer that folds on another stream than the decoder's adds a stream: stage 1 then rotates over one fewer (surge_device_decoder_create)
void fold_stream_seen(surge_device_decoder* d, hipStream_t fold_stream) {
  if (d->push_streams_pinned || fold_stream == d->stream) return;
  std::lock_guard<std::mutex> lk(d->mu);
  if (d->n_push_active == d->n_push_streams && d->n_push_active > 2) d->n_push_active = d->n_push_streams - 1;
}

// the consumer thread's wait for `st`: naps between event queries (surge_device_decoder_create; SURGE_INGEST_WAIT=block: asleep on
// the blocking event, =spin: hipStreamSynchronize)
hipError_t wait_stream(surge_device_decoder* d, hipStream_t st) {
  if (!d->block_waits) return hipStreamSynchronize(st);
  hipError_t e = hipEventRecord(d->sleeper, st);
  if (e != hipSuccess || d->poll_ns <= 0) return e != hipSuccess ? e : hipEventSynchronize(d->sleeper);
  // naps of poll_ns between queries; the thread's timer slack (50 us by default: it would triple a 25 us nap) is 1 us meanwhile
  const int slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
  if (slack > 1000) (void)prctl(PR_SET_TIMERSLACK, 1000ul, 0, 0, 0);
  const timespec nap{0, (long)d->poll_ns};
  while ((e = hipEventQuery(d->sleeper)) == hipErrorNotReady) (void)nanosleep(&nap, nullptr);
  (void)hipGetLastError();  // (hipErrorNotReady is remembered as the thread's last error: not one)
  if (slack > 1000) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)slack, 0, 0, 0);
  return e;
}

// Stage 2: everything behind the per-record metadata and decoded values — interning, compaction, append — on the
// decoder's stream.  Two synchronisations: one in the middle (what the push discovered: errors, new keys, their bytes,
// delivered records — everything the allocations behind it need), one at the end.  Nothing is committed before the
// first: a push that fails takes the keys it probed out of the table again (rollback_kernel), so a failed push leaves
// the decoder exactly as it was.  wait = false leaves the second synchronisation out: the results are complete in the
// order of the decoder's stream (surge_device_decoder_push_finish_async).
int32_t stage2(surge_device_decoder* d, PushSlot& s, bool wait) {
  const int64_t n_rec = s.n_rec;
  if (n_rec == 0) return OK;
  hipStream_t st = d->stream;
  const size_t R = (size_t)n_rec;
  const uint8_t* dby = (const uint8_t*)s.d_bytes.p;
  ErrorCell* derr = (ErrorCell*)s.d_err.p;
  RecMeta* dmeta = (RecMeta*)s.meta.p;
  const unsigned rb = (unsigned)((n_rec + 255) / 256), rb1 = (unsigned)((n_rec + 256) / 256);
  auto poison = [&](int32_t rc) { d->poisoned = true; return rc; };
#define PCHK(call)                                                                                                      \
  do {                                                                                                                  \
    hipError_t e_ = (call);                                                                                             \
    if (e_ != hipSuccess) return poison(dfail(d, e_ == hipErrorOutOfMemory ? E_NOMEM : E_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_))); \
  } while (0)
  // scratch (nothing of the push is in the table yet: an allocation failure here needs no rollback)
  DCHK(d, d->first.reserve((R + 1) * 8, false, st));
  DCHK(d, d->first_scan.reserve((R + 1) * 8, false, st));
  DCHK(d, d->keep.reserve((R + 1) * 4, false, st));
  DCHK(d, d->keep_pos.reserve((R + 1) * 4, false, st));
  {
    size_t tb_a = 0, tb_b = 0;
    DCHK(d, rocprim::exclusive_scan(nullptr, tb_a, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, 0ull, R + 1, rocprim::plus<unsigned long long>(), st));
    DCHK(d, rocprim::exclusive_scan(nullptr, tb_b, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, R + 1, rocprim::plus<uint32_t>(), st));
    DCHK(d, d->temp.reserve(tb_a > tb_b ? tb_a : tb_b, false, st));
    const int32_t rc = ensure_table(d, n_rec);
    if (rc != OK) return rc;
  }
  PCHK(hipStreamWaitEvent(st, s.done, 0));
  if (s.seed != d->seed)  // the table was re-seeded after this push's stage 1 hashed its keys
    hipLaunchKernelGGL(rekey_records_kernel, dim3(rb), dim3(256), 0, st, dmeta, n_rec, dby, d->seed);
  ErrorCell ec;
  unsigned long long first_total = 0;
  uint32_t kept = 0;
  for (int attempt = 0;; ++attempt) {
    hipLaunchKernelGGL(probe_kernel, dim3(rb), dim3(256), 0, st, dmeta, n_rec, table_of(d));
    hipLaunchKernelGGL(flag_kernel, dim3(rb1), dim3(256), 0, st, (const RecMeta*)dmeta, n_rec, dby, table_of(d), (const uint8_t*)d->arena.p, (const int64_t*)d->key_off.p,
                       (unsigned long long*)d->first.p, (uint32_t*)d->keep.p, derr);
    size_t tb = d->temp.cap;
    PCHK(rocprim::exclusive_scan(d->temp.p, tb, (const unsigned long long*)d->first.p, (unsigned long long*)d->first_scan.p, 0ull, R + 1,
                                 rocprim::plus<unsigned long long>(), st));
    tb = d->temp.cap;
    PCHK(rocprim::exclusive_scan(d->temp.p, tb, (const uint32_t*)d->keep.p, (uint32_t*)d->keep_pos.p, 0u, R + 1, rocprim::plus<uint32_t>(), st));
    PCHK(hipMemcpyAsync(&ec, derr, sizeof(ec), hipMemcpyDeviceToHost, st));
    PCHK(hipMemcpyAsync(&first_total, (unsigned long long*)d->first_scan.p + R, 8, hipMemcpyDeviceToHost, st));
    PCHK(hipMemcpyAsync(&kept, (uint32_t*)d->keep_pos.p + R, 4, hipMemcpyDeviceToHost, st));
    PCHK(wait_stream(d, st));
    if (ec.lz4_bad == ~0u && ec.crc_bad == ~0u && ec.first_bad != ~0ull && (uint32_t)(ec.first_bad & 0xff) == RS_COLLISION && attempt < 3) {
      // Two different keys share a 64-bit hash (about 3 in a million pushes at 10^7 keys): the table gets another hash
      // function — every known key re-hashed from its bytes in the arena, the push's records from theirs — and the push
      // goes through again.  Nothing of it was committed.
      hipLaunchKernelGGL(rollback_kernel, dim3(rb), dim3(256), 0, st, (const RecMeta*)dmeta, n_rec, table_of(d));
      d->seed = (d->seed & ~(1ull << 63)) + 1;
      ++d->reseeds;
      hipLaunchKernelGGL(table_clear_kernel, dim3((unsigned)((d->t_cap + 255) / 256)), dim3(256), 0, st, (TableSlot*)d->t_slots.p, d->t_cap);
      if (d->n_keys > 0) {
        const unsigned kb = (unsigned)((d->n_keys + 255) / 256);
        hipLaunchKernelGGL(rekey_keys_kernel, dim3(kb), dim3(256), 0, st, (const uint8_t*)d->arena.p, (const int64_t*)d->key_off.p, d->n_keys, d->seed,
                           (unsigned long long*)d->key_hash.p);
        hipLaunchKernelGGL(rehash_kernel, dim3(kb), dim3(256), 0, st, (const unsigned long long*)d->key_hash.p, d->n_keys, table_of(d));
      }
      hipLaunchKernelGGL(rekey_records_kernel, dim3(rb), dim3(256), 0, st, dmeta, n_rec, dby, d->seed);
      s.h_err = ErrorCell{~0ull, 0u, ec.n_f64_host, ~0u, ~0u};
      PCHK(hipMemcpyAsync(derr, &s.h_err, sizeof(s.h_err), hipMemcpyHostToDevice, st));
      continue;
    }
    break;
  }
  d->counters[0] += n_rec;
  auto rollback = [&]() -> int32_t {
    hipLaunchKernelGGL(rollback_kernel, dim3(rb), dim3(256), 0, st, (const RecMeta*)dmeta, n_rec, table_of(d));
    PCHK(hipStreamSynchronize(st));
    return OK;
  };
  if (ec.crc_bad != ~0u) {
    const int32_t rc = rollback();
    if (rc != OK) return rc;
    return dfail(d, SURGE_E_CORRUPT, "record batch CRC-32C mismatch (verified on the device) in the batch at base offset " +
                                         std::to_string(ec.crc_bad < s.h_secs.size() ? s.h_secs[ec.crc_bad].base_offset : -1));
  }
  if (ec.lz4_bad != ~0u) {
    const int32_t rc = rollback();
    if (rc != OK) return rc;
    return dfail(d, SURGE_E_CORRUPT, "bad LZ4 frame in the batch at base offset " + std::to_string(ec.lz4_bad < s.h_secs.size() ? s.h_secs[ec.lz4_bad].base_offset : -1) +
                                     " (malformed sequence, or a block that is not 64 KiB where it must be)");
  }
  if (ec.first_bad != ~0ull) {
    const int64_t rec = (int64_t)(ec.first_bad >> 8);
    const uint32_t status = (uint32_t)(ec.first_bad & 0xff);
    RecMeta m;
    PCHK(hipMemcpy(&m, dmeta + rec, sizeof(m), hipMemcpyDeviceToHost));
    const int32_t rc = rollback();
    if (rc != OK) return rc;
    // nothing of this push is delivered and no key it discovered stays interned
    return dfail(d, status == RS_COLLISION ? E_UNSUPPORTED : SURGE_E_CORRUPT,
                 "record " + std::to_string(rec) + " of the push (offset " + std::to_string(m.offset) + ") " + why_bad(status) +
                     (status == RS_COLLISION ? " under four hash functions in a row" : ""));
  }
  const int64_t n_new = (int64_t)(first_total >> 40), new_bytes = (int64_t)(first_total & ((1ull << 40) - 1));
  {
    // everything that can fail for want of memory, before the first commit
    hipError_t e = hipSuccess;
    if (n_new > 0) {
      e = d->key_off.reserve((size_t)(d->n_keys + n_new + 1) * 8, true, st);
      if (e == hipSuccess) e = d->key_hash.reserve((size_t)(d->n_keys + n_new) * 8, true, st);
      if (e == hipSuccess) e = d->arena.reserve((size_t)(d->arena_bytes + new_bytes) + 16, true, st);
    }
    if (e == hipSuccess) e = d->r_agg.reserve((size_t)(d->n_records + kept) * 8 + 16, true, st);
    if (e == hipSuccess) e = d->r_ev.reserve((size_t)(d->n_records + kept) * 16 + 16, true, st);
    if (e == hipSuccess) e = d->r_off.reserve((size_t)(d->n_records + kept) * 8 + 16, true, st);
    if (e != hipSuccess) {
      const int32_t rc = rollback();
      if (rc != OK) return rc;
      return dfail(d, e == hipErrorOutOfMemory ? E_NOMEM : E_DEVICE, std::string("growing the key table / result arrays: ") + hipGetErrorString(e));
    }
  }
  if (d->consumed_valid) {  // an asynchronous consumer of the last results (surge_replay_append_decoded_async) reads the arrays finalize_kernel writes
    PCHK(hipStreamWaitEvent(st, d->consumed, 0));
    d->consumed_valid = false;
  }
  if (n_new > 0)
    hipLaunchKernelGGL(assign_kernel, dim3(rb), dim3(256), 0, st, (const RecMeta*)dmeta, n_rec, dby, (const unsigned long long*)d->first.p,
                       (const unsigned long long*)d->first_scan.p, d->n_keys, d->arena_bytes, table_of(d), (uint8_t*)d->arena.p, (int64_t*)d->key_off.p,
                       (unsigned long long*)d->key_hash.p);
  hipLaunchKernelGGL(finalize_kernel, dim3(rb), dim3(256), 0, st, (const RecMeta*)dmeta, n_rec, (const uint32_t*)d->keep.p, (const uint32_t*)d->keep_pos.p, table_of(d),
                     (const uint4*)s.ev_tmp.p, d->n_records, (int64_t*)d->r_agg.p, (uint4*)d->r_ev.p, (int64_t*)d->r_off.p);
  PCHK(hipGetLastError());
  d->n_keys += n_new;
  d->arena_bytes += new_bytes;
  if (ec.n_f64_host > 0) {
    PCHK(hipStreamSynchronize(st));  // the results have to be in place before the payloads are patched
    // Doubles the fast parser could not decide (more than 19 digits, or one of Eisel-Lemire's rare ambiguous products):
    // the host parses exactly those values with the library's host decoder and patches the payload in place
    std::vector<uint32_t> list(ec.n_f64_host);
    PCHK(hipMemcpy(list.data(), s.f64_list.p, (size_t)ec.n_f64_host * 4, hipMemcpyDeviceToHost));
    for (uint32_t i : list) {
      RecMeta m;
      uint32_t pos = 0;
      PCHK(hipMemcpy(&m, dmeta + i, sizeof(m), hipMemcpyDeviceToHost));
      PCHK(hipMemcpy(&pos, (uint32_t*)d->keep_pos.p + i, 4, hipMemcpyDeviceToHost));
      uint8_t ev[16];
      std::vector<uint8_t> value((size_t)m.val_len + 1);
      PCHK(hipMemcpy(value.data(), dby + m.val_off, (size_t)m.val_len, hipMemcpyDeviceToHost));  // (an LZ4 section exists decompressed on the device only)
      if (surge_event_json_decode(&d->h_tmpl, value.data(), m.val_len, ev) != 0)  // (the device accepted the number's spelling: cannot happen)
        return poison(dfail(d, SURGE_E_CORRUPT, "record at offset " + std::to_string(m.offset) + ": " + surge_event_json_last_error()));
      PCHK(hipMemcpy((uint8_t*)d->r_ev.p + (size_t)(d->n_records + pos) * 16, ev, 16, hipMemcpyHostToDevice));
    }
    d->counters[3] += ec.n_f64_host;
  }
  d->chain_fallbacks += ec.reserved;
  d->n_records += kept;
  d->counters[1] += kept;
  d->counters[2] += n_rec - kept;
  ++d->pushes;
  if (wait) {
    PCHK(wait_stream(d, st));
  } else {  // the slot's buffers are read until here: its next stage 1 waits for this point of the stream
    PCHK(hipEventRecord(s.released, st));
    s.released_valid = true;
  }
  return OK;
#undef PCHK
}

// stage 1 is enqueued: the slot joins the queue
int32_t commit_slot(surge_device_decoder* d, PushSlot& s) {
  DCHK(d, hipEventRecord(s.done, s.stream));
  s.busy = true;
  std::lock_guard<std::mutex> lk(d->mu);
  ++d->n_pending;
  return OK;
}

int32_t finish_oldest(surge_device_decoder* d, bool wait) {
  PushSlot* sp;
  {
    std::lock_guard<std::mutex> lk(d->mu);
    if (d->n_pending == 0) sp = nullptr;
    else sp = &d->slots[d->head];
  }
  if (!sp) return dfail(d, SURGE_E_STATE, "push_finish without a pending push_async");
  PushSlot& s = *sp;
  const int32_t rc = stage2(d, s, wait);
  if (rc != OK) {
    // the slot's buffers are still being written by its own stream if stage 2 never waited for it, and read by
    // whatever stage 2 launched before it gave up
    (void)hipStreamSynchronize(s.stream);
    (void)hipStreamSynchronize(d->stream);
    s.released_valid = false;
  }
  s.busy = false;
  std::lock_guard<std::mutex> lk(d->mu);
  d->head = (d->head + 1) % kSlots;
  --d->n_pending;
  return rc;
}

}  // namespace

extern "C" {

int32_t surge_device_decoder_push_parts_async(surge_device_decoder* d, int32_t n_parts, const uint8_t* const* bytes, const surge_batch_section* const* sections,
                                              const int64_t* n_sections) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  if (n_parts < 0 || (n_parts > 0 && (!bytes || !sections || !n_sections))) return dfail(d, E_INVALID, "bad argument");
  int32_t rc = OK;
  PushSlot* s = claim_slot(d, &rc);
  if (!s) return rc;
  DeviceScope scope(d->device);
  rc = await_release(d, *s);
  if (rc != OK) return rc;
  rc = stage1_wire(d, *s, n_parts, bytes, sections, n_sections);
  if (rc != OK) {
    (void)hipStreamSynchronize(s->stream);  // whatever was enqueued before the failure reads host memory of this call
    return rc;
  }
  return commit_slot(d, *s);
}

int32_t surge_device_decoder_push_async(surge_device_decoder* d, const uint8_t* bytes, const surge_batch_section* sections, int64_t n_sections) {
  return surge_device_decoder_push_parts_async(d, 1, &bytes, &sections, &n_sections);
}

int32_t surge_device_decoder_push_finish(surge_device_decoder* d) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  DeviceScope scope(d->device);
  return finish_oldest(d, true);
}

int32_t surge_device_decoder_push_finish_async(surge_device_decoder* d) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  DeviceScope scope(d->device);
  return finish_oldest(d, false);
}

int32_t surge_device_decoder_pending(const surge_device_decoder* d) {
  if (!d) return 0;
  std::lock_guard<std::mutex> lk(const_cast<surge_device_decoder*>(d)->mu);
  return d->n_pending;
}

int32_t surge_device_decoder_reserve(surge_device_decoder* d, int64_t n_keys, int64_t key_bytes) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  if (n_keys < 0 || key_bytes < 0) return dfail(d, E_INVALID, "negative capacity");
  if (d->n_pending != 0) return dfail(d, SURGE_E_STATE, "asynchronous pushes are pending");
  DeviceScope scope(d->device);
  hipStream_t st = d->stream;
  const int64_t extra = n_keys > d->n_keys ? n_keys - d->n_keys : 0;
  {
    const int32_t rc = ensure_table(d, extra);
    if (rc != OK) return rc;
  }
  DCHK(d, d->key_off.reserve((size_t)(n_keys + 1) * 8, true, st));
  DCHK(d, d->key_hash.reserve((size_t)n_keys * 8, true, st));
  DCHK(d, d->arena.reserve((size_t)key_bytes + 16, true, st));
  DCHK(d, hipStreamSynchronize(st));
  return OK;
}

int32_t surge_device_decoder_push(surge_device_decoder* d, const uint8_t* bytes, const surge_batch_section* sections, int64_t n_sections) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  if (n_sections < 0 || (n_sections > 0 && (!bytes || !sections))) return dfail(d, E_INVALID, "bad argument");
  if (d->n_pending != 0) return dfail(d, SURGE_E_STATE, "asynchronous pushes are pending: finish them first (results are appended in push order)");
  if (n_sections == 0) return OK;
  const int32_t rc = surge_device_decoder_push_async(d, bytes, sections, n_sections);
  return rc != OK ? rc : surge_device_decoder_push_finish(d);
}

int32_t surge_device_decoder_push_records(surge_device_decoder* d, const uint8_t* keys, const int64_t* key_off, const uint8_t* values,
                                          const int64_t* value_off, const int64_t* offsets, int64_t n) {
  if (!d) return dfail(nullptr, E_INVALID, "decoder is NULL");
  if (n < 0 || (n > 0 && (!key_off || !value_off))) return dfail(d, E_INVALID, "bad argument");
  if (d->n_pending != 0) return dfail(d, SURGE_E_STATE, "asynchronous pushes are pending: finish them first (results are appended in push order)");
  if (n == 0) return OK;
  int32_t rc = OK;
  PushSlot* s = claim_slot(d, &rc);
  if (!s) return rc;
  DeviceScope scope(d->device);
  rc = await_release(d, *s);
  if (rc != OK) return rc;
  rc = stage1_records(d, *s, keys, key_off, values, value_off, offsets, n);
  if (rc != OK) {
    (void)hipStreamSynchronize(s->stream);
    return rc;
  }
  rc = commit_slot(d, *s);
  return rc != OK ? rc : finish_oldest(d, true);
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
    const hipError_t e = wait_stream(d, d->stream);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return dfail(d, E_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    rc = surge_replay_append_events_device(h, (const int64_t*)d->r_agg.p, d->r_ev.p, d->n_records);
    if (rc != OK) return dfail(d, rc, surge_replay_last_error(h));
    if (d->block_waits) {  // (sleep until the fold is through, then let surge_replay_synchronize report what it has to)
      void* hs = nullptr;
      if (surge_replay_get_stream(h, &hs) == OK) {
        (void)hipSetDevice(d->device);
        (void)wait_stream(d, (hipStream_t)hs);
        (void)hipSetDevice(prev);
      }
    }
    rc = surge_replay_synchronize(h);  // the arrays are reused by the next push
    if (rc != OK) return dfail(d, rc, surge_replay_last_error(h));
  }
  d->n_records = 0;
  return OK;
}

// The same hand-over without a host wait on either side: the handle's stream waits (event) for the decoder's stream to
// have written the arrays, the group-by and the fold are enqueued behind that, and the next push_finish's stage 2 waits
// (event) for the group-by's last read before it writes the arrays again.  The host thread only ever waits in the middle
// of stage 2 (what the push discovered), so interning of fetch i + 1 overlaps the fold of fetch i on the device.
int32_t surge_replay_append_decoded_async(surge_replay_handle* h, surge_device_decoder* d, int64_t* n_events_out, int64_t* n_keys_out) {
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
    void* hs = nullptr;
    rc = surge_replay_get_stream(h, &hs);
    if (rc != OK) return dfail(d, rc, surge_replay_last_error(h));
    DeviceScope scope(d->device);
    fold_stream_seen(d, (hipStream_t)hs);
    DCHK(d, hipEventRecord(d->ready, d->stream));
    DCHK(d, hipStreamWaitEvent((hipStream_t)hs, d->ready, 0));
    rc = surge_replay_append_events_device(h, (const int64_t*)d->r_agg.p, d->r_ev.p, d->n_records);
    // (recorded also when the append failed half way: whatever it enqueued reads the arrays)
    const hipError_t e = hipEventRecord(d->consumed, (hipStream_t)hs);
    d->consumed_valid = e == hipSuccess;
    if (rc != OK) return dfail(d, rc, surge_replay_last_error(h));
    if (e != hipSuccess) {
      (void)surge_replay_synchronize(h);
      return dfail(d, E_DEVICE, std::string("hipEventRecord: ") + hipGetErrorString(e));
    }
  }
  d->n_records = 0;
  return OK;
}

int32_t surge_replay_stage_decoded(surge_replay_handle* h, surge_device_decoder* d, int64_t* n_events_out, int64_t* n_keys_out) {
  if (!h || !d) return dfail(d, E_INVALID, "NULL argument");
  if (n_events_out) *n_events_out = d->n_records;
  if (n_keys_out) *n_keys_out = d->n_keys;
  if (d->n_records > 0) {
    void* hs = nullptr;
    int32_t rc = surge_replay_get_stream(h, &hs);
    if (rc != OK) return dfail(d, rc, surge_replay_last_error(h));
    DeviceScope scope(d->device);
    fold_stream_seen(d, (hipStream_t)hs);
    DCHK(d, hipEventRecord(d->ready, d->stream));
    DCHK(d, hipStreamWaitEvent((hipStream_t)hs, d->ready, 0));
    rc = surge_replay_stage_events_device(h, (const int64_t*)d->r_agg.p, d->r_ev.p, d->n_records);
    const hipError_t e = hipEventRecord(d->consumed, (hipStream_t)hs);  // the next push_finish* waits for the copies out of the result arrays
    d->consumed_valid = e == hipSuccess;
    if (rc != OK) return dfail(d, rc, surge_replay_last_error(h));
    if (e != hipSuccess) {
      (void)surge_replay_synchronize(h);
      return dfail(d, E_DEVICE, std::string("hipEventRecord: ") + hipGetErrorString(e));
    }
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

int32_t surge_device_decoder_stats(const surge_device_decoder* d, int64_t out[8]) {
  if (!d || !out) return E_INVALID;
  for (int i = 0; i < 4; ++i) out[i] = d->counters[i];
  out[4] = d->reseeds;
  out[5] = (int64_t)d->t_cap;
  out[6] = d->pushes;
  out[7] = (int64_t)(d->seed & ~(1ull << 63));
  return OK;
}

}  // extern "C"
