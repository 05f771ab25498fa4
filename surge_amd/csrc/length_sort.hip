// length_sort.hip — the stable descending counting sort behind the per-log indexes (the length order of SORTED / SLOTS, the
// order of the chunk table's virtual rows) and the multi-block exclusive scan of the chunk counts.  Hand-written, in a
// translation unit of its own with no library templates in it: a recovery folds ONCE, so the index is on its critical path,
// and round 5's index — rocPRIM's radix sort — cost 7.5 ms for the 10 M-aggregate log on a fresh process (its code objects
// are loaded at the first launch; 0.28 ms afterwards) against a 12 ms fold.  What the index needs is much less than a
// general sort: keys are event counts, at most a few thousand distinct values (Zipf(1 .. 4096): 4 097), so
//
//   pass 1  histogram   one WAVE per contiguous range of rows; its private histogram lives in LDS (bins x 4 B).  The rows of
//                       one load (64 of them) that share a key are found with one ballot per key bit; the lowest such lane
//                       adds their number — no atomics anywhere, and the order inside a key is the lanes' order
//   scans               per key: exclusive prefix over the waves' histograms (two small kernels over the [waves x bins]
//                       matrix, in slabs of 64 rows), and over the keys in DESCENDING order (one block)
//   pass 2  scatter     the same walk again: every row's position = its wave's cursor for the key + its rank among the
//                       equal keys of its load; cursors live in LDS
//
// Stable (equal lengths keep aggregate order) and deterministic, like the radix sort it replaces (tests compare the two
// permutations element by element).  Keys above kLsMaxBins - 1 fall back to rocPRIM (index_kernels.hip).
#include "replay_internal.h"

namespace surge {
namespace {

constexpr int kLsRowsPerSlab = 64;

__device__ __forceinline__ uint32_t ls_key(const uint32_t* __restrict__ keys, const int64_t* __restrict__ off, int64_t i) {
  if (keys) return keys[i];
  const int64_t len = off[i + 1] - off[i];
  return len < 65535 ? (uint32_t)len : 65535u;
}

// the lanes of this wave (among the valid ones) that hold my key: one ballot per key bit
__device__ __forceinline__ unsigned long long ls_peers(uint32_t key, bool valid, int bits) {
  unsigned long long peers = __ballot(valid);
  for (int b = 0; b < bits; ++b) {
    const bool bit = ((key >> b) & 1u) != 0u;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// one wave per block; dynamic LDS: bins x u32
__global__ void __launch_bounds__(64) ls_hist_kernel(const uint32_t* __restrict__ keys, const int64_t* __restrict__ off, int64_t n, int64_t per,
                                                     int bins, int bits, uint32_t* __restrict__ hm) {
  extern __shared__ uint32_t ls_lds[];
  const int lane = threadIdx.x;
  for (int b = lane; b < bins; b += 64) ls_lds[b] = 0u;
  const int64_t s0 = (int64_t)blockIdx.x * per;
  int64_t s1 = s0 + per;
  s1 = s1 < n ? s1 : n;
  // the next load is in flight while this one is ranked
  int64_t i = s0 + lane;
  uint32_t key_next = i < s1 ? ls_key(keys, off, i) : 0u;
  for (int64_t base = s0; base < s1; base += 64) {
    const bool valid = i < s1;
    const uint32_t key = key_next;
    i += 64;
    key_next = i < s1 ? ls_key(keys, off, i) : 0u;
    const unsigned long long peers = ls_peers(key, valid, bits);
    const bool leader = valid && (__ffsll((long long)peers) - 1 == lane);
    if (leader) ls_lds[key] += (uint32_t)__popcll(peers);
  }
  uint32_t* row = hm + (size_t)blockIdx.x * bins;
  for (int b = lane; b < bins; b += 64) row[b] = ls_lds[b];
}

// slab sums: part[slab][bin] = sum of hm[row][bin] over the slab's rows
__global__ void ls_slab_sum_kernel(const uint32_t* __restrict__ hm, int n_rows, int bins, uint32_t* __restrict__ part) {
  const int bin = blockIdx.x * blockDim.x + threadIdx.x;
  const int slab = blockIdx.y;
  if (bin >= bins) return;
  const int r0 = slab * kLsRowsPerSlab;
  const int r1 = r0 + kLsRowsPerSlab < n_rows ? r0 + kLsRowsPerSlab : n_rows;
  uint32_t sum = 0u;
  for (int r = r0; r < r1; ++r) sum += hm[(size_t)r * bins + bin];
  part[(size_t)slab * bins + bin] = sum;
}

// per bin: exclusive prefix over the slabs (in place), total[bin]
__global__ void ls_slab_scan_kernel(uint32_t* __restrict__ part, int n_slabs, int bins, uint32_t* __restrict__ total) {
  const int bin = blockIdx.x * blockDim.x + threadIdx.x;
  if (bin >= bins) return;
  uint32_t run = 0u;
  for (int s = 0; s < n_slabs; ++s) {
    const uint32_t v = part[(size_t)s * bins + bin];
    part[(size_t)s * bins + bin] = run;
    run += v;
  }
  total[bin] = run;
}

// one block: base[bin] = number of rows with a LARGER key (exclusive scan of total[] from the top bin down), in place
__global__ void __launch_bounds__(1024) ls_bin_scan_kernel(uint32_t* __restrict__ total, int bins) {
  __shared__ uint32_t s_part[1024];
  const int tid = threadIdx.x;
  const int per = (bins + 1023) / 1024;
  // thread t owns the descending run of bins [hi - per + 1 .. hi], hi = bins - 1 - t * per
  const int hi = bins - 1 - tid * per;
  uint32_t sum = 0u;
  for (int k = 0; k < per; ++k) {
    const int b = hi - k;
    if (b >= 0) sum += total[b];
  }
  s_part[tid] = sum;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 1024 partials
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t v = tid >= d ? s_part[tid - d] : 0u;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  uint32_t run = s_part[tid] - sum;  // exclusive
  for (int k = 0; k < per; ++k) {
    const int b = hi - k;
    if (b >= 0) {
      const uint32_t v = total[b];
      total[b] = run;
      run += v;
    }
  }
}

// hm[row][bin] := where wave `row` puts its first row of key `bin`: base[bin] + the slabs before + the rows before it in its slab
__global__ void ls_cursor_kernel(uint32_t* __restrict__ hm, int n_rows, int bins, const uint32_t* __restrict__ part, const uint32_t* __restrict__ base) {
  const int bin = blockIdx.x * blockDim.x + threadIdx.x;
  const int slab = blockIdx.y;
  if (bin >= bins) return;
  const int r0 = slab * kLsRowsPerSlab;
  const int r1 = r0 + kLsRowsPerSlab < n_rows ? r0 + kLsRowsPerSlab : n_rows;
  uint32_t run = base[bin] + part[(size_t)slab * bins + bin];
  for (int r = r0; r < r1; ++r) {
    const uint32_t v = hm[(size_t)r * bins + bin];
    hm[(size_t)r * bins + bin] = run;
    run += v;
  }
}

__global__ void __launch_bounds__(64) ls_scatter_kernel(const uint32_t* __restrict__ keys, const int64_t* __restrict__ off, int64_t n, int64_t per,
                                                        int bins, int bits, const uint32_t* __restrict__ hm, int64_t* __restrict__ perm) {
  extern __shared__ uint32_t ls_lds[];
  const int lane = threadIdx.x;
  const uint32_t* row = hm + (size_t)blockIdx.x * bins;
  for (int b = lane; b < bins; b += 64) ls_lds[b] = row[b];
  const int64_t s0 = (int64_t)blockIdx.x * per;
  int64_t s1 = s0 + per;
  s1 = s1 < n ? s1 : n;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int64_t i = s0 + lane;
  uint32_t key_next = i < s1 ? ls_key(keys, off, i) : 0u;
  for (int64_t base = s0; base < s1; base += 64) {
    const bool valid = i < s1;
    const uint32_t key = key_next;
    const int64_t me = i;
    i += 64;
    key_next = i < s1 ? ls_key(keys, off, i) : 0u;
    const unsigned long long peers = ls_peers(key, valid, bits);
    // every lane of a key reads the cursor, then the key's lowest lane moves it past all of them (LDS operations of one wave
    // execute in program order)
    const uint32_t c0 = valid ? ls_lds[key] : 0u;
    const bool leader = valid && (__ffsll((long long)peers) - 1 == lane);
    if (leader) ls_lds[key] = c0 + (uint32_t)__popcll(peers);
    if (valid) perm[(int64_t)c0 + __popcll(peers & below)] = me;
  }
}

// ---- exclusive scan of K int64 arrays of n elements each (array k at v + k * stride), in place ------------------------------
constexpr int kScanBlock = 256, kScanItems = 8, kScanTile = kScanBlock * kScanItems;

__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t* s_wave, int64_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int64_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kScanBlock / 64; ++w) {
    const int64_t x = s_wave[w];
    if (w < wave) before += x;
    all += x;
  }
  __syncthreads();
  *total = all;
  return before + inc - v;
}

__global__ void __launch_bounds__(kScanBlock) scan_reduce_kernel(const int64_t* __restrict__ v, int64_t n, int64_t stride, int64_t* __restrict__ sums, int64_t nb) {
  __shared__ int64_t s_wave[kScanBlock / 64];
  const int64_t* a = v + (int64_t)blockIdx.y * stride;
  const int64_t t0 = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int64_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (t0 + k < n) sum += a[t0 + k];
  int64_t total;
  (void)block_exclusive_scan(sum, s_wave, &total);
  if (threadIdx.x == 0) sums[(int64_t)blockIdx.y * (nb + 1) + blockIdx.x] = total;
}

// one block per array: block sums -> exclusive offsets, grand total at [nb]
__global__ void __launch_bounds__(kScanBlock) scan_sums_kernel(int64_t* __restrict__ sums, int64_t nb) {
  __shared__ int64_t s_wave[kScanBlock / 64];
  int64_t* a = sums + (int64_t)blockIdx.x * (nb + 1);
  int64_t carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += kScanBlock) {
    const int64_t b = b0 + threadIdx.x;
    const int64_t v = b < nb ? a[b] : 0;
    int64_t total;
    const int64_t ex = block_exclusive_scan(v, s_wave, &total);
    if (b < nb) a[b] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) a[nb] = carry;
}

__global__ void __launch_bounds__(kScanBlock) scan_apply_kernel(int64_t* __restrict__ v, int64_t n, int64_t stride, const int64_t* __restrict__ sums, int64_t nb) {
  __shared__ int64_t s_wave[kScanBlock / 64];
  int64_t* a = v + (int64_t)blockIdx.y * stride;
  const int64_t t0 = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int64_t x[kScanItems];
  int64_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    x[k] = t0 + k < n ? a[t0 + k] : 0;
    sum += x[k];
  }
  int64_t total;
  int64_t run = sums[(int64_t)blockIdx.y * (nb + 1) + blockIdx.x] + block_exclusive_scan(sum, s_wave, &total);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (t0 + k < n) a[t0 + k] = run;
    run += x[k];
  }
}

}  // namespace

// rows a wave of the counting sort owns, and how many waves that makes
static void count_sort_shape(int64_t n, int n_cus, int64_t* per, int64_t* n_waves) {
  int64_t waves = (int64_t)n_cus * 8;  // eight waves per CU: 8 x 16 KB of histograms in a CU's LDS at 4 097 bins
  int64_t p = (n + waves - 1) / waves;
  p = (p + 63) / 64 * 64;
  if (p < 256) p = 256;
  *per = p;
  *n_waves = (n + p - 1) / p;
}

size_t count_sort_scratch_bytes(int64_t n, uint32_t max_key, int n_cus) {
  int64_t per, waves;
  count_sort_shape(n > 0 ? n : 1, n_cus, &per, &waves);
  const size_t bins = (size_t)max_key + 1;
  const size_t slabs = (size_t)((waves + kLsRowsPerSlab - 1) / kLsRowsPerSlab);
  return ((size_t)waves * bins + slabs * bins + bins) * 4 + 256;
}

// perm (n int64) := row ids ordered by key, largest first, equal keys in row order.  key(i) = keys ? keys[i] : min(off[i + 1] - off[i], 65535);
// every key <= max_key < kCountSortMaxBins.  scratch: count_sort_scratch_bytes(n, max_key, n_cus) bytes.
hipError_t launch_count_sort_desc(const uint32_t* keys, const int64_t* off, int64_t n, uint32_t max_key, int n_cus, void* scratch, int64_t* perm,
                                  hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (max_key >= (uint32_t)kCountSortMaxBins) return hipErrorInvalidValue;
  int64_t per, waves;
  count_sort_shape(n, n_cus, &per, &waves);
  const int bins = (int)max_key + 1;
  int bits = 0;
  while ((1u << bits) < (uint32_t)bins) ++bits;
  const int slabs = (int)((waves + kLsRowsPerSlab - 1) / kLsRowsPerSlab);
  uint32_t* hm = (uint32_t*)scratch;
  uint32_t* part = hm + (size_t)waves * bins;
  uint32_t* total = part + (size_t)slabs * bins;
  const unsigned lds = (unsigned)bins * 4u;
  hipLaunchKernelGGL(ls_hist_kernel, dim3((unsigned)waves), dim3(64), lds, stream, keys, off, n, per, bins, bits, hm);
  const dim3 g2((unsigned)((bins + 255) / 256), (unsigned)slabs);
  hipLaunchKernelGGL(ls_slab_sum_kernel, g2, dim3(256), 0, stream, (const uint32_t*)hm, (int)waves, bins, part);
  hipLaunchKernelGGL(ls_slab_scan_kernel, dim3((unsigned)((bins + 255) / 256)), dim3(256), 0, stream, part, slabs, bins, total);
  hipLaunchKernelGGL(ls_bin_scan_kernel, dim3(1), dim3(1024), 0, stream, total, bins);
  hipLaunchKernelGGL(ls_cursor_kernel, g2, dim3(256), 0, stream, hm, (int)waves, bins, (const uint32_t*)part, (const uint32_t*)total);
  hipLaunchKernelGGL(ls_scatter_kernel, dim3((unsigned)waves), dim3(64), lds, stream, keys, off, n, per, bins, bits, (const uint32_t*)hm, perm);
  return hipGetLastError();
}

size_t scan_i64_scratch_bytes(int64_t n, int k_arrays) {
  const int64_t nb = (n + kScanTile - 1) / kScanTile;
  return (size_t)k_arrays * (size_t)(nb + 1) * 8;
}

// k_arrays exclusive scans in place (array k = v + k * stride, n elements each); the totals are NOT appended: callers that
// want a total scan n + 1 elements with a zero in the last one
hipError_t launch_exclusive_scans_i64(int64_t* v, int64_t n, int64_t stride, int k_arrays, void* scratch, hipStream_t stream) {
  if (n <= 0 || k_arrays <= 0) return hipSuccess;
  const int64_t nb = (n + kScanTile - 1) / kScanTile;
  int64_t* sums = (int64_t*)scratch;
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb, (unsigned)k_arrays), dim3(kScanBlock), 0, stream, (const int64_t*)v, n, stride, sums, nb);
  hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned)k_arrays), dim3(kScanBlock), 0, stream, sums, nb);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb, (unsigned)k_arrays), dim3(kScanBlock), 0, stream, v, n, stride, (const int64_t*)sums, nb);
  return hipGetLastError();
}

}  // namespace surge
