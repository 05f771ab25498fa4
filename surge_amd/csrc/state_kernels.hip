// state_kernels.hip — everything on the device that is not the fold itself: the shard map (K4), the state encoders
// (N3), the snapshot delta (N2), the 40-byte wire form of the exchange, bulk point reads, and the HBM stream probes the
// bench calibrates against.  Split from fold_kernels.hip so that the fold kernels' sources (what a committed rocprof
// traffic figure describes) change only when a fold kernel changes.
#include "f64_text.h"
#include "fold_device.h"

namespace surge {
namespace {

// ---- K4: shard map  partitionForKey(s, n) = abs(MurmurHash3.stringHash(s) % n)
// (modules/common/src/main/scala/surge/kafka/KafkaPartitioner.scala:8; scala-library 2.13.8 algorithm); CUT = true
// first applies PartitionStringUpToColon.partitionBy = s.takeWhile(_ != ':') (KafkaPartitioner.scala:38-42)
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

template <bool CUT>
__global__ void partition_hash_kernel(const uint16_t* __restrict__ utf16, const int64_t* __restrict__ str_off, int64_t n,
                                      int32_t n_partitions, int32_t* __restrict__ part_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint16_t* s = utf16 + str_off[i];
  const int64_t full = str_off[i + 1] - str_off[i];
  int64_t len = full;
  if (CUT) {
    len = 0;
    while (len < full && s[len] != (uint16_t)':') ++len;
  }
  uint32_t h = 0xf7ca7fd2u;
  int64_t k = 0;
  for (; k + 1 < len; k += 2) {
    uint32_t d = ((uint32_t)s[k] << 16) + (uint32_t)s[k + 1];
    d *= 0xcc9e2d51u; d = rotl32(d, 15); d *= 0x1b873593u;
    h ^= d; h = rotl32(h, 13); h = h * 5u + 0xe6546b64u;
  }
  if (k < len) {
    uint32_t d = (uint32_t)s[k];
    d *= 0xcc9e2d51u; d = rotl32(d, 15); d *= 0x1b873593u;
    h ^= d;
  }
  h ^= (uint32_t)len;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  const int32_t r = (int32_t)h % n_partitions;  // truncated, like the JVM
  part_out[i] = r < 0 ? -r : r;
}

// ---- HBM read-stream ceiling probes -------------------------------------------------------------
// (a) 16 B/lane register loads, plain or non-temporal; (b) the fold kernels' own transport with the
// arithmetic removed: one wave streams a contiguous range in 16 KiB tiles through global_load_lds nt.
// bench.py reports the fastest as the achievable streaming ceiling beside the 8 TB/s spec.
template <bool NT>
__global__ void __launch_bounds__(256) stream_probe_kernel(const uint4* __restrict__ src, int64_t n_vec,
                                                           uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  auto ld = [&](int64_t k) -> uint4 {
    if (NT) {
      uint4 v;
      v.x = __builtin_nontemporal_load(&src[k].x); v.y = __builtin_nontemporal_load(&src[k].y);
      v.z = __builtin_nontemporal_load(&src[k].z); v.w = __builtin_nontemporal_load(&src[k].w);
      return v;
    }
    return src[k];
  };
  for (; i + 3 * stride < n_vec; i += 4 * stride) {
    const uint4 a = ld(i), b = ld(i + stride), c2 = ld(i + 2 * stride), d = ld(i + 3 * stride);
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c2.x ^ c2.y ^ c2.z ^ c2.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n_vec; i += stride) {
    const uint4 a = ld(i);
    acc ^= a.x ^ a.y ^ a.z ^ a.w;
  }
  if (acc == 0x9e3779b9u) sink[0] = acc;  // practically never; keeps the loads alive
}

__global__ void __launch_bounds__(kWave) stream_probe_lds_kernel(const uint4* __restrict__ src, int64_t n_vec,
                                                                 int64_t vec_per_wave, uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int64_t v0 = (int64_t)blockIdx.x * vec_per_wave;
  int64_t v1 = v0 + vec_per_wave;
  v1 = v1 < n_vec ? v1 : n_vec;
  uint32_t acc = 0;
  for (int64_t v = v0; v + 1024 <= v1; v += 1024) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint4 x = *(const uint4*)(smem + lane * 256);
    acc ^= x.x ^ x.y ^ x.z ^ x.w;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 16; ++q)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + v + q * 64 + lane), (lptr_t)(smem + q * 1024), 16, 0, kLoadAux);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x9e3779b9u) sink[0] = acc;
}

// one wave streams a contiguous range in steps of 16 KiB held in registers (16 non-temporal dwordx4 loads per lane in flight)
__global__ void __launch_bounds__(kWave) stream_probe_regs_kernel(const uint4* __restrict__ src, int64_t n_vec, int64_t vec_per_wave,
                                                                  uint32_t* __restrict__ sink) {
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x;
  const int64_t v0 = (int64_t)blockIdx.x * vec_per_wave;
  int64_t v1 = v0 + vec_per_wave;
  v1 = v1 < n_vec ? v1 : n_vec;
  uint32_t acc = 0;
  v4u r[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) r[q] = v4u{0u, 0u, 0u, 0u};
  for (int64_t v = v0; v + 1024 <= v1; v += 1024) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 16; ++q) acc ^= r[q].x ^ r[q].w;
#pragma unroll
    for (int q = 0; q < 16; ++q) r[q] = __builtin_nontemporal_load((const v4u*)&src[v + q * 64 + lane]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int q = 0; q < 16; ++q) acc ^= r[q].y;
  if (acc == 0x9e3779b9u) sink[0] = acc;
}

// ---- N3: fixed 64-byte state -> serialized text (two passes around an exclusive scan) -----------------
struct JsonTemplateDev {
  uint32_t n_parts;
  uint32_t envelope;  // 0: the template text itself; 1: protobuf State{aggregateId = 1, payload = 2 (the text)}
  uint32_t kind[SURGE_JSON_MAX_PARTS], field_offset[SURGE_JSON_MAX_PARTS], lit_off[SURGE_JSON_MAX_PARTS],
      lit_len[SURGE_JSON_MAX_PARTS];
  uint8_t literals[256];
  const uint8_t* filter;  // nullable: per-aggregate SURGE_SNAP_* kinds; only SURGE_SNAP_VALUE aggregates are encoded
  JsonSide side;          // power-of-5 tables of the Double text, side string columns, the "not a JSON number" counter
};

// an aggregate whose template names a Double that is NaN / infinite has no JSON text (the reference's writeState throws)
__device__ __forceinline__ bool json_all_finite(const JsonTemplateDev& t, const uint8_t* st) {
  for (uint32_t i = 0; i < t.n_parts; ++i)
    if (t.kind[i] == SURGE_JP_F64 && ((*(const uint64_t*)(st + t.field_offset[i]) >> 52) & 0x7ffull) == 0x7ffull) return false;
  return true;
}

__device__ __forceinline__ int dec_len_u64(uint64_t v) {
  int n = 1;
  while (v >= 10ull) { v /= 10ull; ++n; }
  return n;
}

// Jackson's default JSON string escaping: \" \\ \b \f \n \r \t, other controls as \u00XX, the rest verbatim
__device__ __forceinline__ int json_escaped_len(uint8_t c) {
  if (c == '"' || c == '\\' || c == '\b' || c == '\f' || c == '\n' || c == '\r' || c == '\t') return 2;
  return c < 0x20 ? 6 : 1;
}

__device__ __forceinline__ uint8_t* json_put_escaped(uint8_t* o, uint8_t c) {
  const char* hex = "0123456789ABCDEF";
  switch (c) {
    case '"': *o++ = '\\'; *o++ = '"'; return o;
    case '\\': *o++ = '\\'; *o++ = '\\'; return o;
    case '\b': *o++ = '\\'; *o++ = 'b'; return o;
    case '\f': *o++ = '\\'; *o++ = 'f'; return o;
    case '\n': *o++ = '\\'; *o++ = 'n'; return o;
    case '\r': *o++ = '\\'; *o++ = 'r'; return o;
    case '\t': *o++ = '\\'; *o++ = 't'; return o;
    default:
      if (c < 0x20) {
        *o++ = '\\'; *o++ = 'u'; *o++ = '0'; *o++ = '0'; *o++ = (uint8_t)hex[c >> 4]; *o++ = (uint8_t)hex[c & 15];
        return o;
      }
      *o++ = c;
      return o;
  }
}

__device__ __forceinline__ void json_int_value(const uint8_t* st, uint32_t kind, uint32_t off, bool* neg, uint64_t* mag) {
  if (kind == SURGE_JP_I32) {
    const int32_t v = *(const int32_t*)(st + off);
    *neg = v < 0;
    *mag = v < 0 ? (uint64_t)(-(int64_t)v) : (uint64_t)v;
  } else if (kind == SURGE_JP_U32) {
    *neg = false;
    *mag = *(const uint32_t*)(st + off);
  } else {
    const int64_t v = *(const int64_t*)(st + off);
    *neg = v < 0;
    *mag = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
  }
}

// length of the template text for aggregate a
__device__ __forceinline__ int64_t json_text_len(const JsonTemplateDev& t, const uint8_t* st, const uint8_t* __restrict__ keys,
                                                 const int64_t* __restrict__ key_off, int64_t a) {
  int64_t len = 0;
  for (uint32_t i = 0; i < t.n_parts; ++i) {
    const uint32_t k = t.kind[i];
    if (k == SURGE_JP_LITERAL) {
      len += t.lit_len[i];
    } else if (k == SURGE_JP_KEY) {
      len += 2;
      for (int64_t b = key_off[a]; b < key_off[a + 1]; ++b) len += json_escaped_len(keys[b]);
    } else if (k == SURGE_JP_STR) {
      const uint32_t c = t.field_offset[i];
      len += 2;
      for (int64_t b = t.side.str_off[c][a]; b < t.side.str_off[c][a + 1]; ++b) len += json_escaped_len(t.side.str[c][b]);
    } else if (k == SURGE_JP_F64) {
      len += f64_play_json_text(*(const uint64_t*)(st + t.field_offset[i]), t.side.f64, nullptr);
    } else {
      bool neg; uint64_t mag;
      json_int_value(st, k, t.field_offset[i], &neg, &mag);
      len += dec_len_u64(mag) + (neg ? 1 : 0);
    }
  }
  return len;
}

__device__ __forceinline__ int varint_len(uint64_t v) {
  int n = 1;
  while (v >= 0x80ull) { v >>= 7; ++n; }
  return n;
}

__device__ __forceinline__ uint8_t* put_varint(uint8_t* o, uint64_t v) {
  while (v >= 0x80ull) { *o++ = (uint8_t)(v | 0x80ull); v >>= 7; }
  *o++ = (uint8_t)v;
  return o;
}

// Pass 1 (WRITE = false): the serialized length of every aggregate.  Pass 2 (WRITE = true), after the
// exclusive scan turned lengths into offsets: a block's 256 values are contiguous in the output, so they are
// composed in LDS (placed so that LDS offset == global address mod 16) and then stored as whole 16-byte
// words by the block — per-thread byte stores to global were the bottleneck of the first version.  A block
// whose output does not fit the staging buffer (very long keys) writes straight to global.
constexpr int kJsonBlock = 256;
constexpr int kJsonStageBytes = 32 * 1024;

template <bool WRITE>
__global__ void __launch_bounds__(kJsonBlock) json_encode_kernel(const JsonTemplateDev t, const uint4* __restrict__ states, int64_t n,
                                                                 const uint8_t* __restrict__ keys, const int64_t* __restrict__ key_off,
                                                                 int64_t* __restrict__ len_or_off, uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t json_stage[];
  const int64_t a0 = (int64_t)blockIdx.x * kJsonBlock;
  const int64_t a = a0 + threadIdx.x;
  const bool live = a < n;
  const uint8_t* st = (const uint8_t*)(states + (live ? a : 0) * 4);
  const uint32_t fl = *(const uint32_t*)(st + 36);
  bool emit = live && (fl & FL_PRESENT) && !(fl & FL_POISONED) && (!t.filter || t.filter[a] == SURGE_SNAP_VALUE);
  if (emit && !json_all_finite(t, st)) {
    emit = false;
    if (!WRITE) atomicAdd(t.side.not_a_number, 1ull);
  }
  if (!WRITE) {
    if (!live) return;
    int64_t len = 0;
    if (emit) {
      len = json_text_len(t, st, keys, key_off, a);
      if (t.envelope == 1) {  // proto3: empty fields are not written
        const int64_t idlen = key_off[a + 1] - key_off[a];
        len = (idlen ? 1 + varint_len((uint64_t)idlen) + idlen : 0) + (len ? 1 + varint_len((uint64_t)len) + len : 0);
      }
    }
    len_or_off[a] = len;
    return;
  }
  const int64_t a1 = (a0 + kJsonBlock < n) ? a0 + kJsonBlock : n;
  const int64_t base = len_or_off[a0], end = len_or_off[a1];  // [n] holds the total
  const uint32_t shift = (uint32_t)((uintptr_t)(out + base) & 15u);
  const bool staged = (end - base) + shift <= kJsonStageBytes;
  if (emit) {
    uint8_t* o = staged ? json_stage + shift + (len_or_off[a] - base) : out + len_or_off[a];
    if (t.envelope == 1) {
      const int64_t idlen = key_off[a + 1] - key_off[a];
      if (idlen) {
        *o++ = 0x0A;  // field 1, length-delimited
        o = put_varint(o, (uint64_t)idlen);
        for (int64_t b = key_off[a]; b < key_off[a + 1]; ++b) *o++ = keys[b];
      }
      const int64_t plen = json_text_len(t, st, keys, key_off, a);
      if (plen) {
        *o++ = 0x12;  // field 2, length-delimited
        o = put_varint(o, (uint64_t)plen);
      }
    }
    for (uint32_t i = 0; i < t.n_parts; ++i) {
      const uint32_t k = t.kind[i];
      if (k == SURGE_JP_LITERAL) {
        for (uint32_t b = 0; b < t.lit_len[i]; ++b) *o++ = t.literals[t.lit_off[i] + b];
      } else if (k == SURGE_JP_KEY) {
        *o++ = '"';
        for (int64_t b = key_off[a]; b < key_off[a + 1]; ++b) o = json_put_escaped(o, keys[b]);
        *o++ = '"';
      } else if (k == SURGE_JP_STR) {
        const uint32_t c = t.field_offset[i];
        *o++ = '"';
        for (int64_t b = t.side.str_off[c][a]; b < t.side.str_off[c][a + 1]; ++b) o = json_put_escaped(o, t.side.str[c][b]);
        *o++ = '"';
      } else if (k == SURGE_JP_F64) {
        o += f64_play_json_text(*(const uint64_t*)(st + t.field_offset[i]), t.side.f64, o);
      } else {
        bool neg; uint64_t mag;
        json_int_value(st, k, t.field_offset[i], &neg, &mag);
        if (neg) *o++ = '-';
        const int nd = dec_len_u64(mag);
        for (int d = nd - 1; d >= 0; --d) { o[d] = (uint8_t)('0' + (int)(mag % 10ull)); mag /= 10ull; }
        o += nd;
      }
    }
  }
  if (!staged) return;  // block-uniform
  __syncthreads();
  const uint32_t total = (uint32_t)(end - base);
  uint8_t* g = out + base - shift;                    // 16-byte aligned; LDS offset i <-> g[i]
  const uint32_t lo = shift, hi = shift + total;      // valid span in that frame
  const uint32_t body_lo = (lo + 15u) & ~15u, body_hi = hi & ~15u;
  if (body_lo <= body_hi) {
    for (uint32_t i = lo + threadIdx.x; i < body_lo; i += kJsonBlock) g[i] = json_stage[i];
    for (uint32_t i = body_lo + threadIdx.x * 16u; i < body_hi; i += kJsonBlock * 16u) *(uint4*)(g + i) = *(const uint4*)(json_stage + i);
    for (uint32_t i = body_hi + threadIdx.x; i < hi; i += kJsonBlock) g[i] = json_stage[i];
  } else {
    for (uint32_t i = lo + threadIdx.x; i < hi; i += kJsonBlock) g[i] = json_stage[i];
  }
}

// exclusive scan of n int64 values in place (+ total at [n]): per-block scan, scan of block totals, add
constexpr int kScanBlock = 1024;
__global__ void __launch_bounds__(kScanBlock) scan_block_kernel(int64_t* __restrict__ v, int64_t n, int64_t* __restrict__ totals) {
  __shared__ int64_t s[kScanBlock];
  const int tid = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * kScanBlock + tid;
  const int64_t x = i < n ? v[i] : 0;
  s[tid] = x;
  __syncthreads();
  for (int d = 1; d < kScanBlock; d <<= 1) {
    const int64_t y = tid >= d ? s[tid - d] : 0;
    __syncthreads();
    s[tid] += y;
    __syncthreads();
  }
  if (i < n) v[i] = s[tid] - x;  // exclusive
  if (tid == kScanBlock - 1) totals[blockIdx.x] = s[tid];
}

// single block: exclusive scan of totals[0..nb) in place, grand total appended at [nb]
__global__ void __launch_bounds__(1024) scan_totals_kernel(int64_t* totals, int64_t nb) {
  __shared__ int64_t s_part[1024];
  const int tid = threadIdx.x;
  const int64_t per = (nb + 1023) / 1024;
  const int64_t b0 = tid * per, b1 = (b0 + per < nb) ? b0 + per : nb;
  int64_t sum = 0;
  for (int64_t b = b0; b < b1; ++b) sum += totals[b];
  s_part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int64_t run = 0;
    for (int i = 0; i < 1024; ++i) { const int64_t v = s_part[i]; s_part[i] = run; run += v; }
    totals[nb] = run;
  }
  __syncthreads();
  int64_t run = s_part[tid];
  for (int64_t b = b0; b < b1; ++b) { const int64_t v = totals[b]; totals[b] = run; run += v; }
}

__global__ void scan_add_kernel(int64_t* __restrict__ v, int64_t n, const int64_t* __restrict__ totals_excl) {
  const int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x;
  if (i < n) v[i] += totals_excl[blockIdx.x];
}

// snapshot wire form: 5 of the 8 eight-byte words of a state (the reserved tail is always zero)
__global__ void pack_states_kernel(const uint64_t* __restrict__ in, int64_t n, uint64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 8-byte word of the packed form
  if (i >= n * 5) return;
  const int64_t a = i / 5;
  out[i] = in[a * 8 + (i - a * 5)];
}

__global__ void unpack_states_kernel(const uint64_t* __restrict__ in, int64_t n, uint64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 8-byte word of the 64-byte form
  if (i >= n * 8) return;
  const int64_t a = i >> 3;
  const int k = (int)(i & 7);
  out[i] = k < 5 ? in[a * 5 + k] : 0ull;
}

__global__ void gather_states_kernel(const uint4* __restrict__ states, const int64_t* __restrict__ idx, int64_t n,
                                     uint4* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16 B quarter of a state per thread
  if (i >= n * 4) return;
  out[i] = states[idx[i >> 2] * 4 + (i & 3)];
}

// snapshot delta: kind[a] = what the state topic needs for aggregate a relative to the last committed snapshot
// ("publish only if the state changed", PersistentActor.scala:212,257): unchanged or poisoned -> SKIP, Some -> VALUE,
// Some -> None -> TOMBSTONE.  counts[0] += values, counts[1] += tombstones.
// Four lanes per aggregate, one 16-byte quarter of the state each: every load instruction of a wave covers 1 KiB of
// contiguous memory (one thread per aggregate read its 48 bytes at a 64-byte stride: 1.9 ms for 10 M aggregates where the
// 1.28 GB it compares stream in 0.3 ms), the quartet agrees through one ballot, and the counters get one atomic per wave
// of a grid-stride launch instead of one per 64 aggregates.
template <bool FULL64>
__global__ void __launch_bounds__(256) snapshot_delta_kernel(const uint4* __restrict__ states, const uint4* __restrict__ published, int64_t n,
                                                             uint8_t* __restrict__ kind, unsigned long long* __restrict__ counts) {
  const int lane = threadIdx.x & 63, quad = lane & ~3, part = lane & 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;  // a multiple of 4: a lane keeps its quarter
  uint32_t nv = 0, nt = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * 4; i += stride) {
    const uint4 s = states[i], p = published[i];
    bool differs;
    if (part < 2) {
      differs = s.x != p.x || s.y != p.y || s.z != p.z || s.w != p.w;
    } else if (part == 2) {  // v1: bytes 0..39 carry the state (the tail is always zero); v2 slot schemas use all 64 bytes
      differs = s.x != p.x || s.y != p.y || (FULL64 && (s.z != p.z || s.w != p.w));
    } else {
      differs = FULL64 && (s.x != p.x || s.y != p.y || s.z != p.z || s.w != p.w);
    }
    const unsigned long long d = __ballot(differs);
    const uint32_t flags = (uint32_t)__shfl((int)s.y, quad + 2, 64);  // the flags word lives in the third quarter
    if (part == 0) {
      uint32_t k = SURGE_SNAP_SKIP;
      if (((d >> quad) & 0xFull) != 0ull && !(flags & FL_POISONED)) k = (flags & FL_PRESENT) ? SURGE_SNAP_VALUE : SURGE_SNAP_TOMBSTONE;
      kind[i >> 2] = (uint8_t)k;
      nv += k == SURGE_SNAP_VALUE;
      nt += k == SURGE_SNAP_TOMBSTONE;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    nv += (uint32_t)__shfl_down((int)nv, o, 64);
    nt += (uint32_t)__shfl_down((int)nt, o, 64);
  }
  if (lane == 0) {
    if (nv) atomicAdd(&counts[0], (unsigned long long)nv);
    if (nt) atomicAdd(&counts[1], (unsigned long long)nt);
  }
}

// published[a] := states[a] for every aggregate that was just published (kind != SKIP)
__global__ void snapshot_commit_kernel(const uint4* __restrict__ states, uint4* __restrict__ published, int64_t n,
                                       const uint8_t* __restrict__ kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16 B quarter of a state per thread
  if (i >= n * 4) return;
  if (kind[i >> 2] != SURGE_SNAP_SKIP) published[i] = states[i];
}

// a publish that failed AFTER its baseline was committed: make the reported aggregates differ from anything a fold can
// produce (flags word all ones), so the next delta reports them again
__global__ void snapshot_invalidate_kernel(uint4* __restrict__ published, int64_t n, const uint8_t* __restrict__ kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16 B quarter of a state per thread
  if (i >= n * 4) return;
  if (kind[i >> 2] != SURGE_SNAP_SKIP) published[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
}

__global__ void count_poisoned_kernel(const uint4* __restrict__ states, int64_t n, unsigned long long* count) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool p = s < n && (states[s * 4 + 2].y & FL_POISONED);
  const int c = __popcll(__ballot(p));
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}

}  // namespace

hipError_t launch_partition_hash(const uint16_t* utf16, const int64_t* str_off, int64_t n, int32_t n_partitions,
                                 int32_t* part_out, bool up_to_colon, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (up_to_colon)
    hipLaunchKernelGGL(partition_hash_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, utf16, str_off, n,
                       n_partitions, part_out);
  else
    hipLaunchKernelGGL(partition_hash_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, utf16, str_off, n,
                       n_partitions, part_out);
  return hipGetLastError();
}

// variants: 0 plain / 1 non-temporal register loads, grid-stride; 2 / 3 / 4: the LDS-DMA tile stream (the folds' transport)
// with 9 / 6 / 4 resident waves per CU; 5: 16 KiB of non-temporal register loads per wave step, 4 waves per CU (the two
// best configurations of scripts/experiments/probe2.hip — round 3's probe only ran 0..2 and read 6.3 TB/s where these
// read 6.8 - 7.0 on the same boxes)
hipError_t launch_stream_probe(const uint4* src, int64_t n_vec, uint32_t* sink, int variant, hipStream_t stream) {
  if (variant >= 2) {
    const int64_t per_cu = variant == 2 ? 9 : variant == 3 ? 6 : 4;
    const int64_t waves = 256 * per_cu;
    int64_t per = (n_vec / waves) / 1024 * 1024;
    if (per < 1024) per = 1024;
    const int64_t n_waves = n_vec / per;
    if (n_waves <= 0) return hipSuccess;
    if (variant == 5)
      hipLaunchKernelGGL(stream_probe_regs_kernel, dim3((unsigned)n_waves), dim3(kWave), 0, stream, src, n_vec, per, sink);
    else
      hipLaunchKernelGGL(stream_probe_lds_kernel, dim3((unsigned)n_waves), dim3(kWave), 16384, stream, src, n_vec, per, sink);
    return hipGetLastError();
  }
  if (variant == 1)
    hipLaunchKernelGGL(stream_probe_kernel<true>, dim3(256 * 8), dim3(256), 0, stream, src, n_vec, sink);
  else
    hipLaunchKernelGGL(stream_probe_kernel<false>, dim3(256 * 8), dim3(256), 0, stream, src, n_vec, sink);
  return hipGetLastError();
}

// d_len_off: n + 1 entries; d_totals: ceil(n / 1024) + 1 entries of scratch
hipError_t launch_json_encode(const surge_json_template& tmpl, const uint4* states, int64_t n, const uint8_t* keys,
                              const int64_t* key_off, int64_t* d_len_off, int64_t* d_totals, uint8_t* out, bool write_pass,
                              uint32_t envelope, const uint8_t* filter, const JsonSide& side, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  JsonTemplateDev t;
  t.n_parts = tmpl.n_parts;
  t.envelope = envelope;
  t.filter = filter;
  t.side = side;
  for (uint32_t i = 0; i < SURGE_JSON_MAX_PARTS; ++i) {
    t.kind[i] = tmpl.part[i].kind; t.field_offset[i] = tmpl.part[i].field_offset;
    t.lit_off[i] = tmpl.part[i].lit_off; t.lit_len[i] = tmpl.part[i].lit_len;
  }
  for (int i = 0; i < 256; ++i) t.literals[i] = tmpl.literals[i];
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (!write_pass) {
    hipLaunchKernelGGL(json_encode_kernel<false>, dim3(blocks), dim3(kJsonBlock), 0, stream, t, states, n, keys, key_off, d_len_off, out);
    const int64_t nb = (n + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nb), dim3(kScanBlock), 0, stream, d_len_off, n, d_totals);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(1024), 0, stream, d_totals, nb);  // exclusive scan of block totals, grand total at [nb]
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(kScanBlock), 0, stream, d_len_off, n, d_totals);
  } else {
    hipLaunchKernelGGL(json_encode_kernel<true>, dim3(blocks), dim3(kJsonBlock), kJsonStageBytes + 16, stream, t, states, n, keys, key_off, d_len_off, out);
  }
  return hipGetLastError();
}

hipError_t launch_pack_states(const void* in64, int64_t n, void* out40, bool unpack, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (unpack)
    hipLaunchKernelGGL(unpack_states_kernel, dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, stream, (const uint64_t*)in64, n, (uint64_t*)out40);
  else
    hipLaunchKernelGGL(pack_states_kernel, dim3((unsigned)((n * 5 + 255) / 256)), dim3(256), 0, stream, (const uint64_t*)in64, n, (uint64_t*)out40);
  return hipGetLastError();
}

hipError_t launch_gather_states(const uint4* states, const int64_t* idx, int64_t n, uint4* out, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_states_kernel, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, stream, states, idx, n, out);
  return hipGetLastError();
}

hipError_t launch_snapshot_delta(const uint4* states, uint4* published, int64_t n, uint8_t* kind, unsigned long long* d_counts,
                                 bool commit, bool full64, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_counts, 0, 16, stream);
  if (e != hipSuccess || n <= 0) return e;
  const int64_t want = (n * 4 + 255) / 256;
  const unsigned grid = (unsigned)(want < 8192 ? want : 8192);
  if (full64)
    hipLaunchKernelGGL(snapshot_delta_kernel<true>, dim3(grid), dim3(256), 0, stream, states, published, n, kind, d_counts);
  else
    hipLaunchKernelGGL(snapshot_delta_kernel<false>, dim3(grid), dim3(256), 0, stream, states, published, n, kind, d_counts);
  if (commit)
    hipLaunchKernelGGL(snapshot_commit_kernel, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, stream, states, published, n, kind);
  return hipGetLastError();
}

hipError_t launch_snapshot_invalidate(uint4* published, int64_t n, const uint8_t* kind, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(snapshot_invalidate_kernel, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, stream, published, n, kind);
  return hipGetLastError();
}

hipError_t launch_snapshot_commit(const uint4* states, uint4* published, int64_t n, const uint8_t* kind, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(snapshot_commit_kernel, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, stream, states, published, n, kind);
  return hipGetLastError();
}

hipError_t launch_count_poisoned(const uint4* states, int64_t n, unsigned long long* d_count, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream);
  if (e != hipSuccess || n <= 0) return e;
  hipLaunchKernelGGL(count_poisoned_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, states, n, d_count);
  return hipGetLastError();
}

}  // namespace surge
