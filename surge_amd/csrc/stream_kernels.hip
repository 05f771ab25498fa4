// stream_kernels.hip — K3 micro-batch group-by on the device (SURVEY §8d C5, §8f N1's drain): n events in topic
// (offset) order, event i tagged with its aggregate's dense index, become the grouped batch the flat fold kernel takes:
//   stable sort of (aggregate index, position) pairs   — rocPRIM's LSD radix sort over the bits the index needs
//   -> heads (first event of every aggregate's run)   — one compare per event
//   -> exclusive scan of the head flags               — group id of every run
//   -> scatter {group_agg, group_off} + gather of the 16-byte events into sorted order.
// Order inside an aggregate is the topic's order (the sort is stable and the value is the position), which is all
// foldLeft needs (CommandModels.scala:26).  Round 1 did this on the host (LSD radix sort in engine.hip, 5e7 events/s).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cstdlib>
#include <cstring>

#include "replay_internal.h"

namespace surge {
namespace {

// rocPRIM sorts up to 2^20 items with a block sort + merge passes — for a 10^6-event fetch 21 launches (one block sort, ten
// rounds of two kernels) of ~7 us — and beyond that with the LSD "onesweep" radix sort (a histogram, a scan and one pass per 8
// key bits: five launches for the 24 bits a 10 M-aggregate store's indexes need).  Lowering the limit to 2^15 items was built
// and measured (round 5, SURGE_REPLAY_GROUPBY_SORT=onesweep keeps it selectable): alone on the chip the two cost the same
// (146 against 156 us per 10^6 events: onesweep's three passes take 33 us each), and with four fetches in flight onesweep is
// the slower one (265 us: its blocks wait for their predecessors' prefixes — decoupled look-back — while other kernels hold
// the CUs; bytes -> states 6.6 - 6.7 against 6.9e8 events/s on one box).  rocPRIM's own choice stays the default.
using GroupbyOnesweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32768>;

bool groupby_onesweep() {
  static const bool v = [] { const char* e = std::getenv("SURGE_REPLAY_GROUPBY_SORT"); return e && std::strcmp(e, "onesweep") == 0; }();
  return v;
}

template <class Config>
hipError_t sort_pairs(void* temp, size_t& temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                      unsigned key_bits, hipStream_t stream) {
  return rocprim::radix_sort_pairs<Config>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, key_bits, stream);
}

// keys[i] = low 32 bits of agg_idx[i] (n_agg < 2^32 is checked by the caller), vals[i] = i; bad[0] |= out-of-range index
__global__ void groupby_keys_kernel(const int64_t* __restrict__ agg_idx, uint32_t n, int64_t n_agg, uint32_t* __restrict__ keys,
                                    uint32_t* __restrict__ vals, uint32_t* __restrict__ bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t a = agg_idx[i];
  if (a < 0 || a >= n_agg) atomicOr(bad, 1u);
  keys[i] = (uint32_t)a;
  vals[i] = i;
}

__global__ void groupby_heads_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ head) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// gid = exclusive scan of head: run r starts at the event where head == 1 and gid == r
__global__ void groupby_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                       const uint32_t* __restrict__ head, const uint32_t* __restrict__ gid, uint32_t n,
                                       const uint4* __restrict__ events, uint4* __restrict__ sorted_events,
                                       int64_t* __restrict__ group_agg, int64_t* __restrict__ group_off, uint32_t* __restrict__ n_groups) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sorted_events[i] = events[vals[i]];
  if (head[i]) {
    group_agg[gid[i]] = (int64_t)keys[i];
    group_off[gid[i]] = (int64_t)i;
  }
  if (i == n - 1) {
    const uint32_t g = gid[i] + head[i];
    group_off[g] = (int64_t)n;
    // flags: [0] groups of this batch, [1] "an index of this batch was out of range", [2] batches skipped so far (sticky).
    // A batch with a bad index is SKIPPED on the device (0 groups: the fold that follows touches nothing) — the host does
    // not wait for the flag before it launches the fold; it learns about the skip at its next synchronisation point.
    const bool bad = n_groups[1] != 0u;
    n_groups[0] = bad ? 0u : g;
    if (bad) atomicAdd(&n_groups[2], 1u);
  }
}

// ---- the packer (surge_replay_pack_staged) ---------------------------------------------------------------------------------
__global__ void pack_stage_kernel(const int64_t* __restrict__ agg_idx, uint32_t n, uint32_t* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t a = agg_idx[i];
  keys[i] = (a < 0 || a > 0xfffffffell) ? 0xffffffffu : (uint32_t)a;  // (out of range: caught by the pack's check of the largest key)
}

// vals[i] = i; bad[0] |= a staged index that is not an aggregate of the log being packed
__global__ void pack_iota_kernel(const uint32_t* __restrict__ keys, uint32_t n, int64_t n_agg, uint32_t* __restrict__ vals, uint32_t* __restrict__ bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  vals[i] = i;
  if ((int64_t)keys[i] >= n_agg) atomicOr(bad, 1u);
}

// seg_off[a] = first position of an aggregate index >= a in the sorted keys (a = n_agg: the end)
__global__ void pack_seg_off_kernel(const uint32_t* __restrict__ keys, uint32_t n, int64_t n_agg, int64_t* __restrict__ seg_off) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a > n_agg) return;
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if ((int64_t)keys[mid] < a) lo = mid + 1; else hi = mid;
  }
  seg_off[a] = (int64_t)lo;
}

__global__ void pack_gather_kernel(const uint32_t* __restrict__ pos, uint32_t n, const uint4* __restrict__ staged, uint4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = staged[pos[i]];
}

}  // namespace

hipError_t launch_pack_stage(const int64_t* d_agg_idx, uint32_t n, uint32_t* keys, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_stage_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, d_agg_idx, n, keys);
  return hipGetLastError();
}

hipError_t pack_temp_bytes(uint32_t n, unsigned key_bits, size_t* bytes) {
  size_t a = 0;
  const hipError_t e = sort_pairs<rocprim::default_config>(nullptr, a, nullptr, nullptr, nullptr, nullptr, (size_t)n, key_bits, (hipStream_t) nullptr);
  *bytes = a;
  return e;
}

// keys_a: the staged aggregate indices (n x u32); keys_b / vals_a / vals_b: n x u32 scratch; seg_off: n_agg + 1; out: n x 16 B;
// d_bad: one u32, != 0 afterwards when an index was >= n_agg (the arrays then hold garbage: the caller drops them)
hipError_t launch_pack(const uint32_t* keys_a, const uint4* staged_events, uint32_t n, int64_t n_agg, unsigned key_bits, void* d_temp, size_t temp_bytes,
                       uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int64_t* seg_off, uint4* out, uint32_t* d_bad, hipStream_t stream) {
  const unsigned blocks = (n + 255u) / 256u;
  hipError_t e0 = hipMemsetAsync(d_bad, 0, 4, stream);
  if (e0 != hipSuccess) return e0;
  if (n > 0) {
    hipLaunchKernelGGL(pack_iota_kernel, dim3(blocks), dim3(256), 0, stream, keys_a, n, n_agg, vals_a, d_bad);
    size_t tb = temp_bytes;
    const hipError_t e = sort_pairs<rocprim::default_config>(d_temp, tb, keys_a, keys_b, vals_a, vals_b, (size_t)n, key_bits, stream);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(pack_seg_off_kernel, dim3((unsigned)((n_agg + 1 + 255) / 256)), dim3(256), 0, stream, (const uint32_t*)keys_b, n, n_agg, seg_off);
  if (n > 0) hipLaunchKernelGGL(pack_gather_kernel, dim3(blocks), dim3(256), 0, stream, (const uint32_t*)vals_b, n, staged_events, out);
  return hipGetLastError();
}

// scratch bytes the two rocPRIM primitives need for n events (the larger of the two)
hipError_t groupby_temp_bytes(uint32_t n, unsigned key_bits, size_t* bytes) {
  size_t a = 0, b = 0;
  hipError_t e = sort_pairs<rocprim::default_config>(nullptr, a, nullptr, nullptr, nullptr, nullptr, (size_t)n, key_bits, (hipStream_t) nullptr);
  if (e != hipSuccess) return e;
  {  // (the larger of the two sorts' needs: which one runs is a run-time choice)
    size_t a2 = 0;
    e = sort_pairs<GroupbyOnesweep>(nullptr, a2, nullptr, nullptr, nullptr, nullptr, (size_t)n, key_bits, (hipStream_t) nullptr);
    if (e != hipSuccess) return e;
    a = a2 > a ? a2 : a;
  }
  e = rocprim::exclusive_scan(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n, rocprim::plus<uint32_t>(),
                              (hipStream_t) nullptr);
  if (e != hipSuccess) return e;
  *bytes = a > b ? a : b;
  return hipSuccess;
}

// All buffers device: keys_a/keys_b/vals_a/vals_b/head/gid: n x u32 each; d_flags: {n_groups, bad, skipped batches} (3 x u32;
// the third word is sticky: zeroed by the owner, never here).
hipError_t launch_groupby(const int64_t* d_agg_idx, const uint4* d_events, uint32_t n, int64_t n_agg, unsigned key_bits, void* d_temp,
                          size_t temp_bytes, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, uint32_t* head,
                          uint32_t* gid, uint4* d_sorted_events, int64_t* d_group_agg, int64_t* d_group_off, uint32_t* d_flags,
                          hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (n + 255u) / 256u;
  hipError_t e = hipMemsetAsync(d_flags, 0, 8, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(groupby_keys_kernel, dim3(blocks), dim3(256), 0, stream, d_agg_idx, n, n_agg, keys_a, vals_a, d_flags + 1);
  {
    size_t tb = temp_bytes;
    e = groupby_onesweep() ? sort_pairs<GroupbyOnesweep>(d_temp, tb, keys_a, keys_b, vals_a, vals_b, (size_t)n, key_bits, stream)
                           : sort_pairs<rocprim::default_config>(d_temp, tb, keys_a, keys_b, vals_a, vals_b, (size_t)n, key_bits, stream);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(groupby_heads_kernel, dim3(blocks), dim3(256), 0, stream, keys_b, n, head);
  e = rocprim::exclusive_scan(d_temp, temp_bytes, (const uint32_t*)head, gid, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(groupby_scatter_kernel, dim3(blocks), dim3(256), 0, stream, keys_b, vals_b, head, gid, n, d_events, d_sorted_events,
                     d_group_agg, d_group_off, d_flags);
  return hipGetLastError();
}

}  // namespace surge
