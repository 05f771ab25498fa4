// index_kernels.hip — the per-log indexes of the lane-per-row kernels, built once when a log is bound (or per micro-batch
// for the slot fold): the length order of SORTED / SLOTS and the chunk table of CHUNKED / TILED.
//
// Round 2 built both with a counting sort whose histogram and scatter used one global atomic per segment: on a Zipf log
// a few thousand length buckets take all of them, the atomics serialise in L2 (1.7 ns each, measured: 3.5 ms per 2 M
// segments per pass) and the "index" of the 10 M-aggregate log cost 35 ms — three times the fold it serves.  A recovery
// folds ONCE, so that is part of what recovery costs.  Now: no contended atomics at all —
//   length order : key = min(length, 65535), value = segment id -> rocPRIM LSD radix sort (descending, 16 key bits, stable);
//   chunk table  : per aggregate {chunks, is-cut, chunks-if-cut} -> three exclusive scans give every aggregate its first
//                  virtual row, its entry in the cut list and its first side slot -> rows are emitted in aggregate order ->
//                  the same radix sort orders them by length -> one gather writes the table the fold kernels read.
// Both orders are now deterministic (stable sort: equal lengths keep aggregate order), which the atomic cursors were not.
// Round 6: rows shorter than 8192 events (every log of the BASELINE configs) are ordered by the hand-written counting sort of
// length_sort.hip and the chunk counts are scanned by its multi-block scan; rocPRIM's radix sort stays for longer rows, in a
// translation unit of its own (index_radix.hip) — its code objects cost 7 ms to load at the first launch of ANY kernel that
// shares a translation unit with them, which a recovery that never needs them should not pay.
#include "fold_chunk_device.h"

namespace surge {
namespace {

constexpr uint32_t kLenKeyMax = 65535u;  // rows longer than this share the first bucket (they are few: each is long)

__global__ void length_keys_kernel(const int64_t* __restrict__ off, int64_t n_seg, uint32_t* __restrict__ keys, int64_t* __restrict__ vals) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seg) return;
  const int64_t len = off[s + 1] - off[s];
  keys[s] = len < (int64_t)kLenKeyMax ? (uint32_t)len : kLenKeyMax;
  vals[s] = s;
}

// chunks of an aggregate whose events span `span` slots from its 128-byte line start
__host__ __device__ __forceinline__ uint32_t chunks_of(int64_t span, uint32_t T) {
  const int64_t c = (span + T - 1) / T;
  return (uint32_t)(c < 1 ? 1 : c);
}

// cnt[0][s] = chunks of aggregate s, cnt[1][s] = 1 if it is cut, cnt[2][s] = its chunks if it is cut (side slots)
// (align: rows are tiled from the 128-byte line that holds their first event — what the CSR kernel wants; the tile-major
// re-layout copies rows to tile boundaries anyway and passes align = false: no pad events)
__global__ void chunk_count_kernel(const int64_t* __restrict__ off, int64_t n_seg, uint32_t T, bool align, int64_t* __restrict__ cnt,
                                   int64_t stride) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seg) return;
  const int64_t st = off[s], len = off[s + 1] - st;
  const uint32_t c = chunks_of(len + (align ? (st & 7) : 0), T);
  cnt[s] = c;
  cnt[stride + s] = c > 1 ? 1 : 0;
  cnt[2 * stride + s] = c > 1 ? c : 0;
}

// one thread per aggregate: its chunks as virtual rows [voff[s], voff[s] + c), in aggregate order; a cut aggregate also
// registers itself for the stitch kernel.  Chunk k = [base + floor8(k span / c), base + floor8((k+1) span / c)):
// boundaries on whole lines, lengths within 8 events of each other.
__global__ void chunk_emit_kernel(const int64_t* __restrict__ off, int64_t n_seg, const int64_t* __restrict__ out_map, uint32_t T, bool align,
                                  const int64_t* __restrict__ cnt, int64_t stride, int64_t* __restrict__ u_start, uint32_t* __restrict__ u_len,
                                  uint32_t* __restrict__ u_info, int64_t* __restrict__ u_dest, uint32_t* __restrict__ keys,
                                  int64_t* __restrict__ vals, int64_t* __restrict__ r_slot0, uint32_t* __restrict__ r_c, int64_t* __restrict__ r_out) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seg) return;
  const int64_t st = off[s], len = off[s + 1] - st;
  const uint32_t pad = align ? (uint32_t)(st & 7) : 0u;
  const int64_t base = st - pad;  // the 128-byte line that holds the first event (align), or the first event itself
  const int64_t span = len + pad, end = st + len;
  const uint32_t c = chunks_of(span, T);
  const int64_t oi = out_map ? out_map[s] : s;
  const int64_t v0 = cnt[s];
  int64_t slot0 = 0;
  if (c > 1) {
    const int64_t r = cnt[stride + s];
    slot0 = cnt[2 * stride + s];
    r_slot0[r] = slot0; r_c[r] = c; r_out[r] = oi;
  }
  for (uint32_t k = 0; k < c; ++k) {
    const int64_t lo = base + (((int64_t)k * span / c) & ~7ll);
    const int64_t hiE = k + 1 == c ? end : base + (((int64_t)(k + 1) * span / c) & ~7ll);
    const bool empty = hiE <= lo;  // cannot happen while T >= 16 (span / c >= 8); kept as a guard
    const int64_t pos = v0 + k;
    const uint32_t vl = empty ? 0u : (uint32_t)(hiE - lo);
    u_start[pos] = empty ? 0 : lo;  // an empty chunk still "reads" (clamped, ignored): keep it in bounds
    u_len[pos] = vl;
    // every chunk of a cut aggregate is walked relative, chunk 0 included: the stitch kernel starts from the
    // aggregate's prior state; an aggregate in one piece is walked concretely and stores its own state
    u_info[pos] = (c > 1 ? (VI_RELATIVE | VI_SIDE) : 0u) | ((k == 0 ? pad : 0u) << VI_PAD_SHIFT);
    u_dest[pos] = c > 1 ? slot0 + k : oi;
    keys[pos] = vl < kLenKeyMax ? vl : kLenKeyMax;
    if (vals) vals[pos] = pos;  // (the counting sort numbers the rows itself)
  }
}

__global__ void chunk_gather_kernel(const int64_t* __restrict__ order, int64_t n, const int64_t* __restrict__ u_start,
                                    const uint32_t* __restrict__ u_len, const uint32_t* __restrict__ u_info, const int64_t* __restrict__ u_dest,
                                    int64_t* __restrict__ v_start, uint32_t* __restrict__ v_len, uint32_t* __restrict__ v_info,
                                    int64_t* __restrict__ v_dest) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t src = order[i];
  v_start[i] = u_start[src]; v_len[i] = u_len[src]; v_info[i] = u_info[src]; v_dest[i] = u_dest[src];
}

}  // namespace

// perm (n_seg int64) := kernel-facing segment ids sorted by length, longest first (stable)
hipError_t launch_sort_by_length(const int64_t* off, int64_t n_seg, const IndexScratch& sc, int64_t* perm, hipStream_t stream) {
  if (n_seg <= 0) return hipSuccess;
  if (sc.counting) return launch_count_sort_desc(nullptr, off, n_seg, sc.max_key, sc.n_cus, sc.temp, perm, stream);
  hipLaunchKernelGGL(length_keys_kernel, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, stream, off, n_seg, sc.keys_a, sc.vals_a);
  return launch_radix_sort_pairs_desc(sc.temp, sc.temp_bytes, sc.keys_a, sc.keys_b, sc.vals_a, perm, n_seg, stream);
}

// Phase 1 of the chunk table: cnt = three arrays of n_seg + 1 int64 (stride n_seg + 1), left as exclusive scans with their
// totals in the last element: {virtual rows, cut aggregates, side slots}.  The host reads the three totals, sizes the
// table and runs phase 2.
hipError_t launch_chunk_count(const int64_t* off, int64_t n_seg, uint32_t T, bool align, int64_t* cnt, void* scan_scratch, hipStream_t stream) {
  if (n_seg <= 0) return hipSuccess;
  const int64_t stride = n_seg + 1;
  hipLaunchKernelGGL(chunk_count_kernel, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, stream, off, n_seg, T, align, cnt, stride);
  for (int k = 0; k < 3; ++k) {
    hipError_t e = hipMemsetAsync(cnt + k * stride + n_seg, 0, 8, stream);  // the extra element: its scan value is the total
    if (e != hipSuccess) return e;
  }
  return launch_exclusive_scans_i64(cnt, stride, stride, 3, scan_scratch, stream);  // (scan_i64_scratch_bytes(stride, 3) bytes)
}

// Phase 2: rows in aggregate order (u_*), sorted by length (longest first, stable), gathered into v_*
hipError_t launch_chunk_table(const int64_t* off, int64_t n_seg, const int64_t* out_map, uint32_t T, bool align, const int64_t* cnt,
                              int64_t n_vrows, const IndexScratch& sc, int64_t* u_start, uint32_t* u_len, uint32_t* u_info, int64_t* u_dest,
                              int64_t* v_start, uint32_t* v_len, uint32_t* v_info, int64_t* v_dest, int64_t* r_slot0, uint32_t* r_c,
                              int64_t* r_out, hipStream_t stream) {
  if (n_seg <= 0 || n_vrows <= 0) return hipSuccess;
  hipLaunchKernelGGL(chunk_emit_kernel, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, stream, off, n_seg, out_map, T, align, cnt,
                     n_seg + 1, u_start, u_len, u_info, u_dest, sc.keys_a, sc.vals_a, r_slot0, r_c, r_out);
  hipError_t e;
  if (sc.counting) {
    e = launch_count_sort_desc(sc.keys_a, nullptr, n_vrows, sc.max_key, sc.n_cus, sc.temp, sc.vals_b, stream);
  } else {
    e = launch_radix_sort_pairs_desc(sc.temp, sc.temp_bytes, sc.keys_a, sc.keys_b, sc.vals_a, sc.vals_b, n_vrows, stream);
  }
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(chunk_gather_kernel, dim3((unsigned)((n_vrows + 255) / 256)), dim3(256), 0, stream, sc.vals_b, n_vrows, u_start, u_len,
                     u_info, u_dest, v_start, v_len, v_info, v_dest);
  return hipGetLastError();
}

}  // namespace surge
