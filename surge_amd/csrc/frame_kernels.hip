// frame_kernels.hip — the state-topic record batches of a bulk publish, framed on the GPU.
//
// What it replaces: snapshot_writer.cpp's append loop on the host.  A publish of config C5 (0.95 M changed aggregates of
// 10 M, 85 MB of record batches every 30 micro-batches) spent 40 of its 52 ms there — one host thread per partition
// pushing varints and bytes into vectors — while the states, their JSON text and their keys were already in HBM.  Here
// the GPU writes the records where they go and only what the host is better at stays on the host: the CRC-32C of each
// finished batch (the crc32 instruction: 11 GB/s per thread) over the bytes after they arrived in page-locked memory.
//
// The output is BYTE-IDENTICAL to surge_snapshot_writer_append + flush on the same input (tests/test_frame_gpu.py):
// same record order (per partition, aggregate index order), same batch cuts (a batch closes with the record that makes
// it reach max_records or max_bytes), same header fields.  A record is
//     varint(body) | attributes 0 | varint(timestampDelta = 0) | varint(offsetDelta) | varint(klen) key | varint(vlen | -1) value | varint(0 headers)
// and the only field that depends on where a batch starts is offsetDelta (1 byte below 64, 2 below 8192, 3 below 2^20)
// — and through it the size of the length prefix.  So with v = 1, 2, 3 the record's size is known up to that choice:
//     size_v(i) = varint_size(base_i + v) + base_i + v,        base_i = everything but the offsetDelta
// and three prefix sums C_v over the records of a partition give the bytes of ANY run [s, s + m) in closed form:
//     bytes(s, m) = C_1[s .. s+min(m,64)) + C_2[s+64 .. s+min(m,8192)) + C_3[s+8192 .. s+m)
// which makes the greedy cut a binary search per batch (one thread per partition walks its few batches) and gives every
// record its byte position without any sequential pass over the records.
//
// Steps (all on the framer's stream): select the changed aggregates (rocPRIM select) -> their partition + base size ->
// stable radix sort by partition -> the three scans -> batches per partition (count, then emit at scanned offsets) ->
// one thread per record writes its bytes, one per batch its 61-byte header -> one device -> host copy -> host CRCs.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/surge_ingest.h"  // surge_crc32c
#include "../../include/surge_replay.h"  // SURGE_SNAP_*
#include "../../include/surge_snapshot.h"

namespace {

constexpr int32_t OK = 0, E_INVALID = -1, E_DEVICE = -3, E_NOMEM = -4, E_RANGE = -6;
constexpr int kHeader = 61;
constexpr int64_t kD1 = 64, kD2 = 8192, kD3 = 1 << 20;  // offsetDelta below these takes 1 / 2 / 3 bytes

struct Batch {          // device + host
  int64_t out_off;      // first byte of the batch (its header) in the output
  int64_t first;        // index of its first record in partition-sorted order
  int64_t base_offset;  // Kafka offset of its first record
  int32_t count;
  int32_t partition;
  int64_t records_bytes;
};

__host__ __device__ inline int varint_size(int64_t x) {
  uint64_t z = ((uint64_t)x << 1) ^ (uint64_t)(x >> 63);
  int n = 1;
  while (z >= 0x80) { z >>= 7; ++n; }
  return n;
}

__device__ inline uint8_t* put_varint(uint8_t* p, int64_t x) {
  uint64_t z = ((uint64_t)x << 1) ^ (uint64_t)(x >> 63);
  while (z >= 0x80) {
    *p++ = (uint8_t)(z | 0x80);
    z >>= 7;
  }
  *p++ = (uint8_t)z;
  return p;
}

// a record's key / value, 8 bytes at a time: neither end is aligned, which global memory on gfx9+ does not ask for (the
// byte loop this replaces made the write kernel 1.9 ms of an 85 MB publish)
__device__ inline void copy_bytes(uint8_t* dst, const uint8_t* src, int64_t n) {
  int64_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t v;
    __builtin_memcpy(&v, src + i, 8);
    __builtin_memcpy(dst + i, &v, 8);
  }
  for (; i < n; ++i) dst[i] = src[i];
}

__device__ inline void put_be(uint8_t* p, uint64_t x, int n) {
  for (int i = 0; i < n; ++i) p[i] = (uint8_t)(x >> (8 * (n - 1 - i)));
}

// per selected record r (aggregate sel[r]): its partition and the size of its body without the offsetDelta; bad[0] is
// set when a partition is out of range, a span is negative or a kind is unknown
__global__ void frame_size_kernel(const int64_t* __restrict__ sel, int64_t n_sel, const uint8_t* __restrict__ kind, const int32_t* __restrict__ part,
                                  int32_t n_part, const int64_t* __restrict__ key_off, const int64_t* __restrict__ val_off, bool have_keys,
                                  bool have_values, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ base,
                                  uint32_t* __restrict__ bad) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_sel) return;
  const int64_t a = sel[r];
  const uint8_t k = kind[a];
  const int32_t p = part[a];
  const bool value = k == SURGE_SNAP_VALUE;
  const int64_t klen = key_off[a + 1] - key_off[a];
  const int64_t vlen = value && val_off ? val_off[a + 1] - val_off[a] : -1;
  if ((!value && k != SURGE_SNAP_TOMBSTONE) || p < 0 || p >= n_part || klen < 0 || (value && (!val_off || vlen < 0)) || klen > (1 << 28) ||
      vlen > (1 << 28) || (klen > 0 && !have_keys) || (vlen > 0 && !have_values))
    atomicOr(bad, 1u);
  keys[r] = (uint32_t)(p < 0 ? 0 : p);
  vals[r] = (uint32_t)r;
  base[r] = (uint32_t)(1 + 1 + varint_size(klen) + klen + varint_size(vlen) + (vlen > 0 ? vlen : 0) + 1);
}

// j = position in partition order: c_v[j] = size of record order[j] if its offsetDelta takes v bytes (scanned afterwards)
__global__ void frame_sizes3_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ base, int64_t n_sel, int64_t* __restrict__ c1,
                                    int64_t* __restrict__ c2, int64_t* __restrict__ c3) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > n_sel) return;
  if (j == n_sel) {  // the element the exclusive scans leave the totals in
    c1[j] = c2[j] = c3[j] = 0;
    return;
  }
  const int64_t b = base[order[j]];
  c1[j] = varint_size(b + 1) + b + 1;
  c2[j] = varint_size(b + 2) + b + 2;
  c3[j] = varint_size(b + 3) + b + 3;
}

// first position of every partition in the sorted order (pstart[n_part] = n_sel)
__global__ void frame_pstart_kernel(const uint32_t* __restrict__ sorted_part, int64_t n_sel, int32_t n_part, int64_t* __restrict__ pstart) {
  const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_part) return;
  int64_t lo = 0, hi = n_sel;  // first j with sorted_part[j] >= p
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_part[mid] < (uint32_t)p) lo = mid + 1; else hi = mid;
  }
  pstart[p] = lo;
}

__device__ inline int64_t run_bytes(const int64_t* c1, const int64_t* c2, const int64_t* c3, int64_t s, int64_t m) {
  const int64_t m1 = m < kD1 ? m : kD1;
  int64_t b = c1[s + m1] - c1[s];
  if (m > kD1) b += c2[s + (m < kD2 ? m : kD2)] - c2[s + kD1];
  if (m > kD2) b += c3[s + m] - c3[s + kD2];
  return b;
}

// One thread per partition walks its batches.  emit == false: nb[p] / pbytes[p] only; emit == true: the batches are
// written at batch index nb_off[p] and byte offset pbytes_off[p] (the exclusive scans of the first pass).
__global__ void frame_batches_kernel(const int64_t* __restrict__ pstart, int32_t n_part, const int64_t* __restrict__ c1, const int64_t* __restrict__ c2,
                                     const int64_t* __restrict__ c3, int32_t max_records, int64_t max_bytes, const int64_t* __restrict__ next_offset,
                                     bool emit, int64_t* __restrict__ nb, int64_t* __restrict__ pbytes, Batch* __restrict__ batches) {
  const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_part) return;
  int64_t s = pstart[p];
  const int64_t e = pstart[p + 1];
  int64_t count = 0, bytes = 0;
  int64_t bi = emit ? nb[p] : 0, at = emit ? pbytes[p] : 0;
  while (s < e) {
    int64_t m = e - s < max_records ? e - s : max_records;  // the batch closes at max_records at the latest
    if (run_bytes(c1, c2, c3, s, m) >= max_bytes) {         // ... or with the first record that brings it to max_bytes
      int64_t lo = 1, hi = m;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (run_bytes(c1, c2, c3, s, mid) >= max_bytes) hi = mid; else lo = mid + 1;
      }
      m = lo;
    }
    const int64_t rb = run_bytes(c1, c2, c3, s, m);
    if (emit) {
      Batch b;
      b.out_off = at + bytes;
      b.first = s;
      b.base_offset = next_offset[p] + (s - pstart[p]);
      b.count = (int32_t)m;
      b.partition = p;
      b.records_bytes = rb;
      batches[bi + count] = b;
    }
    bytes += kHeader + rb;
    count += 1;
    s += m;
  }
  if (!emit) {
    nb[p] = count;
    pbytes[p] = bytes;
  }
}

// one thread per record (partition order): find its batch, its place in it, write it
__global__ void frame_write_kernel(const Batch* __restrict__ batches, int64_t n_batches, const uint32_t* __restrict__ order, const int64_t* __restrict__ sel,
                                   int64_t n_sel, const uint8_t* __restrict__ kind, const uint32_t* __restrict__ base, const int64_t* __restrict__ c1,
                                   const int64_t* __restrict__ c2, const int64_t* __restrict__ c3, const uint8_t* __restrict__ keys_utf8,
                                   const int64_t* __restrict__ key_off, const uint8_t* __restrict__ values, const int64_t* __restrict__ val_off,
                                   uint8_t* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_sel) return;
  int64_t lo = 0, hi = n_batches;  // last batch whose first record is <= j
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (batches[mid].first <= j) lo = mid; else hi = mid;
  }
  const Batch b = batches[lo];
  const int64_t d = j - b.first;
  const uint32_t r = order[j];
  const int64_t a = sel[r];
  const int v = d < kD1 ? 1 : (d < kD2 ? 2 : 3);
  const int64_t body = (int64_t)base[r] + v;
  uint8_t* p = out + b.out_off + kHeader + run_bytes(c1, c2, c3, b.first, d);
  p = put_varint(p, body);
  *p++ = 0;  // attributes
  *p++ = 0;  // timestampDelta 0: every record of a publish carries the publish's timestamp
  p = put_varint(p, d);
  const int64_t k0 = key_off[a], klen = key_off[a + 1] - k0;
  p = put_varint(p, klen);
  copy_bytes(p, keys_utf8 + k0, klen);
  p += klen;
  if (kind[a] == SURGE_SNAP_VALUE) {
    const int64_t v0 = val_off[a], vlen = val_off[a + 1] - v0;
    p = put_varint(p, vlen);
    copy_bytes(p, values + v0, vlen);
    p += vlen;
  } else {
    p = put_varint(p, -1);  // null value: a tombstone
  }
  *p = 0;  // no headers
}

__global__ void frame_header_kernel(const Batch* __restrict__ batches, int64_t n_batches, int64_t timestamp_ms, uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_batches) return;
  const Batch b = batches[i];
  uint8_t* o = out + b.out_off;
  put_be(o, (uint64_t)b.base_offset, 8);
  put_be(o + 8, (uint64_t)(kHeader - 12 + b.records_bytes), 4);  // batchLength: everything after this field
  put_be(o + 12, 0, 4);                                           // partitionLeaderEpoch
  o[16] = 2;                                                      // magic
  put_be(o + 17, 0, 4);                                           // crc: the host fills it in
  put_be(o + 21, 0, 2);                                           // attributes: no codec, CreateTime, not transactional
  put_be(o + 23, (uint64_t)(b.count - 1), 4);                     // lastOffsetDelta
  put_be(o + 27, (uint64_t)timestamp_ms, 8);                      // baseTimestamp
  put_be(o + 35, (uint64_t)timestamp_ms, 8);                      // maxTimestamp
  put_be(o + 43, ~0ull, 8);                                       // producerId -1
  put_be(o + 51, 0xffffull, 2);                                   // producerEpoch -1
  put_be(o + 53, 0xffffffffull, 4);                               // baseSequence -1
  put_be(o + 57, (uint64_t)b.count, 4);
}

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    const size_t want = cap * 2 > bytes ? cap * 2 : bytes;
    void* fresh = nullptr;
    const hipError_t e = hipMalloc(&fresh, want);
    if (e != hipSuccess) return e;
    if (p) (void)hipFree(p);
    p = fresh;
    cap = want;
    return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

thread_local std::string g_frame_err;

}  // namespace

struct surge_device_framer {
  int device = 0;
  hipStream_t stream = nullptr;
  int32_t n_part = 0, max_records = 10000;
  int64_t max_bytes = 1 << 20;
  std::string err;
  std::vector<int64_t> next_offset, part_byte_off;
  std::vector<Batch> h_batches;
  Buf sel, n_sel, keys_a, keys_b, vals_a, vals_b, base, c, pstart, nb, pbytes, d_next, batches, out, bad, temp;
  uint8_t* pinned = nullptr;
  size_t pinned_cap = 0;
};

namespace {

int32_t ffail(surge_device_framer* f, int32_t code, const std::string& m) {
  if (f) f->err = m;
  g_frame_err = m;
  return code;
}

#define FCHK(f, call)                                                                                                       \
  do {                                                                                                                      \
    const hipError_t e_ = (call);                                                                                           \
    if (e_ != hipSuccess) return ffail(f, e_ == hipErrorOutOfMemory ? E_NOMEM : E_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct Guard {
  int prev = -1;
  explicit Guard(int dev) { (void)hipGetDevice(&prev); (void)hipSetDevice(dev); }
  ~Guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

inline unsigned grid(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

int32_t surge_device_framer_create(int32_t device_id, void* hip_stream, int32_t n_partitions, int32_t max_records_per_batch, int64_t max_batch_bytes,
                                   surge_device_framer** out) {
  if (!out) return ffail(nullptr, E_INVALID, "out is NULL");
  *out = nullptr;
  if (n_partitions <= 0 || n_partitions > (1 << 20) || max_records_per_batch < 0 || max_records_per_batch > kD3 || max_batch_bytes < 0)
    return ffail(nullptr, E_INVALID, "bad argument (partitions in 1 .. 2^20, at most 2^20 records per batch)");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device_id < 0 || device_id >= n_dev)
    return ffail(nullptr, E_DEVICE, "no usable HIP device (the device framer has no CPU fallback: use surge_snapshot_writer_append)");
  surge_device_framer* f = new (std::nothrow) surge_device_framer();
  if (!f) return ffail(nullptr, E_NOMEM, "out of host memory");
  f->device = device_id;
  f->stream = (hipStream_t)hip_stream;
  f->n_part = n_partitions;
  if (max_records_per_batch > 0) f->max_records = max_records_per_batch;
  if (max_batch_bytes > 0) f->max_bytes = max_batch_bytes;
  try {
    f->next_offset.assign((size_t)n_partitions, 0);
    f->part_byte_off.assign((size_t)n_partitions + 1, 0);
  } catch (const std::bad_alloc&) {
    delete f;
    return ffail(nullptr, E_NOMEM, "out of host memory");
  }
  *out = f;
  return OK;
}

int32_t surge_device_framer_destroy(surge_device_framer* f) {
  if (!f) return OK;
  Guard g(f->device);
  (void)hipStreamSynchronize(f->stream);
  for (Buf* b : {&f->sel, &f->n_sel, &f->keys_a, &f->keys_b, &f->vals_a, &f->vals_b, &f->base, &f->c, &f->pstart, &f->nb, &f->pbytes, &f->d_next,
                 &f->batches, &f->out, &f->bad, &f->temp})
    b->release();
  if (f->pinned) (void)hipHostFree(f->pinned);
  delete f;
  return OK;
}

const char* surge_device_framer_last_error(const surge_device_framer* f) { return f ? f->err.c_str() : g_frame_err.c_str(); }

int32_t surge_device_framer_next_offsets(const surge_device_framer* f, int64_t* out) {
  if (!f || !out) return ffail(nullptr, E_INVALID, "bad argument");
  std::memcpy(out, f->next_offset.data(), f->next_offset.size() * 8);
  return OK;
}

int32_t surge_device_framer_frame(surge_device_framer* f, int64_t n_aggregates, const uint8_t* d_kind, const int32_t* d_partition,
                                  const uint8_t* d_keys_utf8, const int64_t* d_key_off, const uint8_t* d_values, const int64_t* d_val_off,
                                  int64_t timestamp_ms, const uint8_t** bytes_out, const int64_t** part_byte_off_out, int64_t* n_records_out,
                                  int64_t* n_batches_out) {
  if (!f) return ffail(nullptr, E_INVALID, "framer is NULL");
  if (n_aggregates < 0 || n_aggregates > (1ll << 32) - 2 || !bytes_out || !part_byte_off_out)
    return ffail(f, E_INVALID, "bad argument");
  if (n_aggregates > 0 && (!d_kind || !d_partition || !d_key_off)) return ffail(f, E_INVALID, "NULL device buffer");
  Guard g(f->device);
  hipStream_t st = f->stream;
  const int32_t P = f->n_part;
  const int64_t n = n_aggregates;
  std::fill(f->part_byte_off.begin(), f->part_byte_off.end(), 0);
  *bytes_out = f->pinned;
  *part_byte_off_out = f->part_byte_off.data();
  if (n_records_out) *n_records_out = 0;
  if (n_batches_out) *n_batches_out = 0;
  if (n == 0) return OK;

  // 1. the changed aggregates, in index order
  FCHK(f, f->sel.reserve((size_t)n * 8));
  FCHK(f, f->n_sel.reserve(16));
  FCHK(f, f->bad.reserve(16));
  size_t tb_select = 0, tb_sort = 0, tb_scan = 0;
  FCHK(f, rocprim::select(nullptr, tb_select, rocprim::counting_iterator<int64_t>(0), d_kind, (int64_t*)f->sel.p, (int64_t*)f->n_sel.p, (size_t)n, st));
  FCHK(f, f->temp.reserve(tb_select));
  FCHK(f, hipMemsetAsync(f->bad.p, 0, 4, st));
  FCHK(f, rocprim::select(f->temp.p, tb_select, rocprim::counting_iterator<int64_t>(0), d_kind, (int64_t*)f->sel.p, (int64_t*)f->n_sel.p, (size_t)n, st));
  int64_t n_sel = 0;
  FCHK(f, hipMemcpyAsync(&n_sel, f->n_sel.p, 8, hipMemcpyDeviceToHost, st));
  FCHK(f, hipStreamSynchronize(st));
  if (n_sel == 0) return OK;
  if (n_sel > 0xfffffff0ll) return ffail(f, E_RANGE, "more than 2^32 records in one publish");

  // 2. partition + base size per record, stable sort by partition
  unsigned bits = 1;
  while ((1ll << bits) < P) ++bits;
  using SortConfig = rocprim::default_config;  // (merge sort up to 2^20 records; a lower limit was measured for K3's group-by and dropped: stream_kernels.hip)
  FCHK(f, rocprim::radix_sort_pairs<SortConfig>(nullptr, tb_sort, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                                (size_t)n_sel, 0u, bits, st));
  FCHK(f, rocprim::exclusive_scan(nullptr, tb_scan, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)n_sel + 1, rocprim::plus<int64_t>(), st));
  FCHK(f, f->temp.reserve(tb_sort > tb_scan ? tb_sort : tb_scan));
  for (Buf* b : {&f->keys_a, &f->keys_b, &f->vals_a, &f->vals_b, &f->base}) FCHK(f, b->reserve((size_t)n_sel * 4));
  FCHK(f, f->c.reserve((size_t)(n_sel + 1) * 8 * 3));
  FCHK(f, f->pstart.reserve((size_t)(P + 1) * 8));
  FCHK(f, f->nb.reserve((size_t)(P + 1) * 8));
  FCHK(f, f->pbytes.reserve((size_t)(P + 1) * 8));
  FCHK(f, f->d_next.reserve((size_t)P * 8));
  hipLaunchKernelGGL(frame_size_kernel, dim3(grid(n_sel)), dim3(256), 0, st, (const int64_t*)f->sel.p, n_sel, d_kind, d_partition, P, d_key_off, d_val_off,
                     d_keys_utf8 != nullptr, d_values != nullptr, (uint32_t*)f->keys_a.p, (uint32_t*)f->vals_a.p, (uint32_t*)f->base.p, (uint32_t*)f->bad.p);
  FCHK(f, rocprim::radix_sort_pairs<SortConfig>(f->temp.p, tb_sort, (const uint32_t*)f->keys_a.p, (uint32_t*)f->keys_b.p, (const uint32_t*)f->vals_a.p,
                                                (uint32_t*)f->vals_b.p, (size_t)n_sel, 0u, bits, st));
  const uint32_t* order = (const uint32_t*)f->vals_b.p;
  int64_t* c1 = (int64_t*)f->c.p;
  int64_t* c2 = c1 + (n_sel + 1);
  int64_t* c3 = c2 + (n_sel + 1);
  hipLaunchKernelGGL(frame_sizes3_kernel, dim3(grid(n_sel + 1)), dim3(256), 0, st, order, (const uint32_t*)f->base.p, n_sel, c1, c2, c3);
  for (int64_t* cv : {c1, c2, c3})
    FCHK(f, rocprim::exclusive_scan(f->temp.p, tb_scan, (const int64_t*)cv, cv, (int64_t)0, (size_t)n_sel + 1, rocprim::plus<int64_t>(), st));
  hipLaunchKernelGGL(frame_pstart_kernel, dim3(grid(P + 1)), dim3(256), 0, st, (const uint32_t*)f->keys_b.p, n_sel, P, (int64_t*)f->pstart.p);

  // 3. the batches of every partition: count, scan, emit
  FCHK(f, hipMemcpyAsync(f->d_next.p, f->next_offset.data(), (size_t)P * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(frame_batches_kernel, dim3(grid(P)), dim3(256), 0, st, (const int64_t*)f->pstart.p, P, c1, c2, c3, f->max_records, f->max_bytes,
                     (const int64_t*)f->d_next.p, false, (int64_t*)f->nb.p, (int64_t*)f->pbytes.p, (Batch*)nullptr);
  FCHK(f, hipMemsetAsync((int64_t*)f->nb.p + P, 0, 8, st));
  FCHK(f, hipMemsetAsync((int64_t*)f->pbytes.p + P, 0, 8, st));
  size_t tb_small = tb_scan;  // P + 1 <= 2^20 + 1 elements: the scratch sized for n_sel + 1 may be smaller
  {
    size_t need = 0;
    FCHK(f, rocprim::exclusive_scan(nullptr, need, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)P + 1, rocprim::plus<int64_t>(), st));
    if (need > f->temp.cap) {
      FCHK(f, hipStreamSynchronize(st));  // the scans above still use the scratch
      FCHK(f, f->temp.reserve(need));
    }
    tb_small = need;
  }
  FCHK(f, rocprim::exclusive_scan(f->temp.p, tb_small, (const int64_t*)f->nb.p, (int64_t*)f->nb.p, (int64_t)0, (size_t)P + 1, rocprim::plus<int64_t>(), st));
  FCHK(f, rocprim::exclusive_scan(f->temp.p, tb_small, (const int64_t*)f->pbytes.p, (int64_t*)f->pbytes.p, (int64_t)0, (size_t)P + 1, rocprim::plus<int64_t>(), st));
  int64_t totals[2] = {0, 0};
  FCHK(f, hipMemcpyAsync(&totals[0], (int64_t*)f->nb.p + P, 8, hipMemcpyDeviceToHost, st));
  FCHK(f, hipMemcpyAsync(&totals[1], (int64_t*)f->pbytes.p + P, 8, hipMemcpyDeviceToHost, st));
  uint32_t bad = 0;
  FCHK(f, hipMemcpyAsync(&bad, f->bad.p, 4, hipMemcpyDeviceToHost, st));
  FCHK(f, hipMemcpyAsync(f->part_byte_off.data(), f->pbytes.p, (size_t)(P + 1) * 8, hipMemcpyDeviceToHost, st));
  FCHK(f, hipStreamSynchronize(st));
  if (bad) {
    std::fill(f->part_byte_off.begin(), f->part_byte_off.end(), 0);
    return ffail(f, E_RANGE, "a changed aggregate has an unknown kind, a partition outside [0, n_partitions), a negative key / value span, or bytes in a table that was passed as NULL");
  }
  const int64_t n_batches = totals[0], n_bytes = totals[1];
  FCHK(f, f->batches.reserve((size_t)n_batches * sizeof(Batch)));
  FCHK(f, f->out.reserve((size_t)n_bytes));
  if ((size_t)n_bytes > f->pinned_cap) {
    if (f->pinned) (void)hipHostFree(f->pinned);
    f->pinned = nullptr;
    f->pinned_cap = 0;
    const size_t want = (size_t)n_bytes + (size_t)n_bytes / 4;
    void* hp = nullptr;
    FCHK(f, hipHostMalloc(&hp, want, hipHostMallocDefault));
    f->pinned = (uint8_t*)hp;
    f->pinned_cap = want;
  }
  try {
    f->h_batches.resize((size_t)n_batches);
  } catch (const std::bad_alloc&) {
    return ffail(f, E_NOMEM, "out of host memory");
  }
  hipLaunchKernelGGL(frame_batches_kernel, dim3(grid(P)), dim3(256), 0, st, (const int64_t*)f->pstart.p, P, c1, c2, c3, f->max_records, f->max_bytes,
                     (const int64_t*)f->d_next.p, true, (int64_t*)f->nb.p, (int64_t*)f->pbytes.p, (Batch*)f->batches.p);

  // 4. the bytes
  hipLaunchKernelGGL(frame_write_kernel, dim3(grid(n_sel)), dim3(256), 0, st, (const Batch*)f->batches.p, n_batches, order, (const int64_t*)f->sel.p, n_sel,
                     d_kind, (const uint32_t*)f->base.p, c1, c2, c3, d_keys_utf8, d_key_off, d_values, d_val_off, (uint8_t*)f->out.p);
  hipLaunchKernelGGL(frame_header_kernel, dim3(grid(n_batches)), dim3(256), 0, st, (const Batch*)f->batches.p, n_batches, timestamp_ms, (uint8_t*)f->out.p);
  FCHK(f, hipGetLastError());
  FCHK(f, hipMemcpyAsync(f->pinned, f->out.p, (size_t)n_bytes, hipMemcpyDeviceToHost, st));
  FCHK(f, hipMemcpyAsync(f->h_batches.data(), f->batches.p, (size_t)n_batches * sizeof(Batch), hipMemcpyDeviceToHost, st));
  FCHK(f, hipStreamSynchronize(st));

  // 5. CRC-32C of every batch (attributes .. end), on the host's crc32 instruction, batches spread over a few threads
  {
    (void)surge_crc32c((const uint8_t*)"", 0);  // initialise the dispatch before threads race for it
    uint8_t* base = f->pinned;
    const std::vector<Batch>& hb = f->h_batches;
    std::atomic<int64_t> next{0};
    auto work = [&]() {
      for (int64_t i = next.fetch_add(1); i < n_batches; i = next.fetch_add(1)) {
        uint8_t* o = base + hb[(size_t)i].out_off;
        const uint32_t crc = surge_crc32c(o + 21, (int64_t)(kHeader - 21) + hb[(size_t)i].records_bytes);
        o[17] = (uint8_t)(crc >> 24); o[18] = (uint8_t)(crc >> 16); o[19] = (uint8_t)(crc >> 8); o[20] = (uint8_t)crc;
      }
    };
    unsigned hw = std::thread::hardware_concurrency();
    int n_threads = (int)(hw ? hw : 1);
    if (n_threads > 8) n_threads = 8;
    if ((int64_t)n_threads > n_batches) n_threads = (int)n_batches;
    if (n_bytes < (4 << 20)) n_threads = 1;
    std::vector<std::thread> th;
    try {
      for (int t = 1; t < n_threads; ++t) th.emplace_back(work);
    } catch (...) {  // the threads that did start, and this one, do the work
    }
    work();
    for (std::thread& t : th) t.join();
  }
  for (const Batch& b : f->h_batches) f->next_offset[(size_t)b.partition] += b.count;
  *bytes_out = f->pinned;
  if (n_records_out) *n_records_out = n_sel;
  if (n_batches_out) *n_batches_out = n_batches;
  return OK;
}

}  // extern "C"
