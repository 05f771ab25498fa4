// fold_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the aggregate-replay hot path.
//
// What is computed (reference semantics):
//   state[a] = events[seg_off[a] .. seg_off[a+1]).foldLeft(init[a])(handleEvent)
//   — modules/command-engine/scaladsl/src/main/scala/surge/scaladsl/command/CommandModels.scala:26
// with handleEvent restated per event type by a 32-bit descriptor (include/surge_replay.h).
//
// How (MI355X-first; there is no reference counterpart — the reference folds one event per
// actor message on the JVM):
//   * The event log is streamed exactly once, 16 B/lane, perfectly coalesced, by direct
//     global->LDS loads (global_load_lds_dwordx4).  One wave owns one 16 KiB tile at a time
//     (64 lanes x 16 consecutive events); the source lane permutation rotates each lane's
//     16 events inside its own 256 B LDS row so that the later ds_read_b128 of
//     "event j of lane l" is bank-conflict free.
//   * foldLeft is sequential, the GPU is not: each lane folds its 16 consecutive events into a
//     *state transformer* (per field either "relative to the incoming state": a delta for
//     count/sum64/event_count, a clamp for min/max, unchanged for version/balance — or "absolute"),
//     and transformers compose associatively, so a wave-level segmented scan (segment heads =
//     aggregate boundaries) yields every aggregate's final state in strict event order.  Integer
//     adds wrap mod 2^32 / 2^64 exactly like JVM Int/Long, min/max/set are exact, f64 payloads are
//     only moved — the result is bit-identical to the sequential fold under any association.
//   * The only thing a transformer cannot be relative to is PRESENCE (Some/None decides whether a
//     REQUIRE-class event applies and whether MATERIALIZE resets to defaults).  So presence (and the
//     sticky "an event threw" flag) is resolved first: a cheap pre-pass tracks three booleans per
//     lane, four wave ballots turn them into each lane's incoming (present, poisoned) with a couple
//     of bit scans, and the main walk then runs ONE evaluation path per lane with mask arithmetic
//     from a pre-expanded per-type op table (no per-event decode, no divergent branches).
//   * A wave task is a contiguous range of WHOLE segments (~256 KiB of events) chosen by a tiny
//     plan kernel from the CSR offsets, so waves never exchange carries through memory: the
//     running state of a segment that spans tiles is carried in scalar registers.
//   * No MFMA: this is an HBM-bound fold (16 B in per event, 64 B out per aggregate).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "fold_flat_device.h"
#include "fold_lane_device.h"

namespace surge {
namespace {

template <int MODE, int LE>
__global__ void __launch_bounds__(kWave) fold_kernel(const FoldParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fold_flat_body<MODE, LE>(p, smem);
}

// ---- K1 "rows": uniform fan-in, one lane per aggregate --------------------------------------------
// When every aggregate has the same number of events L (L % 16 == 0) a wave takes 64 consecutive
// aggregates and walks them in lockstep: tile c holds events [LE c, LE c + LE) of each of its 64 rows
// (64 pieces of LE*16 bytes at a row stride of 16 L bytes; the bare load pattern streams 6.1-6.8 TB/s on MI355X, close to
// the linear stream).  Lane l then owns row l outright, so its running state is CONCRETE from the first
// event on: no presence pre-pass, no transformer scan, no cross-lane traffic at all — the walk is the
// same mask arithmetic as the flat kernel and the 64 B results leave as one contiguous 4 KiB store.
template <int LE, bool CONC>
__global__ void __launch_bounds__(kWave) fold_rows_kernel(const FoldParams p) {
  __shared__ __attribute__((aligned(16))) char lds_ev[Geo<LE>::kTileBytes];
  __shared__ __attribute__((aligned(16))) uint32_t lds_tab[kTableLdsDwords];
  fold_rows_body<LE, CONC>(p, lds_ev, lds_tab);
}

// ---- K2 "sorted rows": arbitrary CSR, one lane per aggregate ----------------------------------------
// Segments are counting-sorted by length at load time (perm, longest first).  Persistent waves pull
// groups of 64 consecutive perm entries from an atomic counter; inside a group the lengths are (almost)
// equal, so the 64 lanes walk their own segments in lockstep exactly like the uniform rows kernel:
// concrete running state, no presence pre-pass, no scan.  A lane whose segment ends early pads with the
// null event.  Row pieces are LE*16 bytes at arbitrary 16 B-aligned addresses.

template <int LE>
__global__ void __launch_bounds__(kWave) fold_sorted_kernel(const FoldParams p) {
  using G = Geo<LE>;
  // one dynamic LDS buffer (separate objects as in fold_rows_kernel made no measurable difference here)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_ev = smem;
  int64_t* lds_rs = (int64_t*)(smem + G::kTileBytes);              // 64 row starts (aux area) ...
  uint32_t* lds_len = (uint32_t*)(smem + G::kTileBytes + kWave * 8);  // ... and 64 row lengths
  uint32_t* lds_tab = (uint32_t*)(smem + G::kTileBytes + G::kAuxSorted);
  const int lane = threadIdx.x;
  load_table<LE>(p, lds_tab, lane);
  const uint32_t ev_row = G::ev_row(lane);
  const int64_t n_groups = (p.n_seg + kWave - 1) / kWave;
  const int64_t* perm = p.plan;

  auto grab = [&]() -> int64_t {
    unsigned long long g = 0;
    if (lane == 0) g = atomicAdd(p.counter, 1ull);
    return (int64_t)(((uint64_t)rl((uint32_t)(g >> 32), 0) << 32) | rl((uint32_t)g, 0));
  };
  // A row is tiled from the 128 B line that contains its first event: `start` is rounded down to a multiple
  // of 8 events and the `pad` events in front of the segment (they belong to its predecessor) are treated as
  // null events.  Every LE*16-byte piece is then line-aligned; without this each 512 B piece touched five
  // lines instead of four (FETCH_SIZE was 22 % above the algorithmic bytes).
  struct Meta { int64_t s, start; uint32_t len, pad; };
  auto load_meta = [&](int64_t g) -> Meta {
    Meta m; m.s = -1; m.start = 0; m.len = 0u; m.pad = 0u;
    const int64_t idx = g * kWave + lane;
    if (g < n_groups && idx < p.n_seg) {
      m.s = perm[idx];
      const int64_t st = p.seg_off[m.s];
      m.pad = (uint32_t)(st & 7);
      m.start = st - m.pad;
      m.len = (uint32_t)(p.seg_off[m.s + 1] - st) + m.pad;
    }
    return m;
  };

  int64_t g = grab();
  Meta cur = load_meta(g);
  while (g < n_groups) {
    const int64_t g_next = grab();
    const Meta nxt = load_meta(g_next);  // in flight while this group is walked

    uint32_t maxlen = cur.len, minlen = cur.s >= 0 ? cur.len : 0xffffffffu;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, d, 64));
      minlen = min(minlen, (uint32_t)__shfl_xor((int)minlen, d, 64));
    }
    const int n_tiles = (int)((maxlen + LE - 1) / LE);

    // the rows each load instruction serves (row RPL*q + lane/LE): starts and lengths live in a small LDS
    // table rather than in 3 registers per load instruction — the register budget buys resident waves
    lds_rs[lane] = cur.start;
    lds_len[lane] = cur.len;
    auto issue = [&](int c) {
      if ((uint32_t)(c + 1) * LE <= minlen) {
#pragma unroll
        for (int q = 0; q < G::kLoads; ++q) {
          const int64_t e = lds_rs[G::kRowsPerLoad * q + lane / LE] + (int64_t)c * LE + G::load_j(lane, q % G::kClasses);
          __builtin_amdgcn_global_load_lds((gptr_t)(p.events + e), (lptr_t)(lds_ev + q * 1024), 16, 0, kLoadAux);
        }
      } else {  // some row ends inside this tile: never read past a row's own events
#pragma unroll
        for (int q = 0; q < G::kLoads; ++q) {
          const int r = G::kRowsPerLoad * q + lane / LE;
          const uint32_t rlen = lds_len[r];
          uint32_t j = (uint32_t)c * LE + G::load_j(lane, q % G::kClasses);
          const uint32_t lastj = rlen ? rlen - 1u : 0u;
          j = j < lastj ? j : lastj;
          __builtin_amdgcn_global_load_lds((gptr_t)(p.events + (lds_rs[r] + j)), (lptr_t)(lds_ev + q * 1024), 16, 0, kLoadAux);
        }
      }
    };

    const int64_t oi = cur.s >= 0 ? (p.out_map ? p.out_map[cur.s] : cur.s) : -1;
    Acc a = (p.init && oi >= 0) ? load_state(p.init, oi) : acc_none();
    uint32_t frozenM = (uint32_t)__builtin_amdgcn_sbfe((int32_t)a.fl, 1, 1);
    uint32_t corr = 0u;
    issue(0);
    for (int c = 0; c < n_tiles; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      uint4 ev[LE];
#pragma unroll
      for (int j = 0; j < LE; ++j) ev[j] = *(const uint4*)(lds_ev + (ev_row ^ (uint32_t)(j * 16)));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (c + 1 < n_tiles) issue(c + 1);

      uint32_t tyc[LE];
      if (c > 0 && (uint32_t)(c + 1) * LE <= minlen) {
#pragma unroll
        for (int j = 0; j < LE; ++j) tyc[j] = type_off(ev[j].x);
      } else {
        const int32_t rem = (int32_t)cur.len - c * LE;         // my remaining events (may be <= 0)
        const int32_t skip = c == 0 ? (int32_t)cur.pad : 0;   // events in front of my segment
#pragma unroll
        for (int j = 0; j < LE; ++j)
          tyc[j] = (j >= skip && j < rem) ? type_off(ev[j].x) : kNullEntryOffBytes;
      }
      walk_events<LE, false>(a, frozenM, corr, ev, tyc, 0u, lds_tab, p, [](int) {});
    }
    a.sum = (int64_t)((uint64_t)a.sum + corr);
    if (oi >= 0) store_state(p.out, oi, a);

    g = g_next;
    cur = nxt;
  }
  dispenser_leave(p.counter, lane);
}

// ---- K1s "short rows": any CSR of many short aggregates, one lane per aggregate, no LDS transport, no index ---------------
// Rows r .. r + 63 of a wave are consecutive in the log, so what the wave reads is one contiguous stretch of it (every cache
// line it touches is used whole, by neighbouring lanes); every lane loads its own events — four 16-byte loads in flight — and
// walks a concrete state like the rows kernels do.  A wave takes as many steps as its longest row needs: the kernel is for logs
// whose rows are all short (engine.hip: longest <= 64 events, mean < 16), where the tile kernels' 16-event lanes would walk
// mostly padding and the flat kernel pays a presence pass, a scan and a state store per segment HEAD.  Empty rows are rows
// like any other (their state is the prior one, or None): no compaction, no fill pass.
__global__ void __launch_bounds__(256) fold_short_kernel(const FoldParams p) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_tab[kTableLdsDwords];
  {
    const uint32_t* src = &p.table[0][0];
    for (int i = threadIdx.x; i < kTableEntries * kTableWords; i += 256)
      lds_tab[(i / kTableWords == kTableEntries - 1 ? kNullEntryOff : (i / kTableWords) * kTableStride) + (i % kTableWords)] = src[i];
  }
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool mine = r < p.n_seg;
  const int64_t s0 = mine ? p.seg_off[r] : 0, s1 = mine ? p.seg_off[r + 1] : 0;
  Acc a = (mine && p.init) ? load_state(p.init, r) : acc_none();
  uint32_t frozenM = (uint32_t)__builtin_amdgcn_sbfe((int32_t)a.fl, 1, 1), presentM = (uint32_t)__builtin_amdgcn_sbfe((int32_t)a.fl, 0, 1), corr = 0u;
  begin_concrete(a);
  const int64_t last = p.n_events > 0 ? p.n_events - 1 : 0;
  for (int64_t j = s0; __builtin_amdgcn_ballot_w64(j < s1) != 0ull; j += 4) {  // (wave-uniform trip count: the longest row of the wave)
    uint4 ev[4];
    uint32_t tyc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t at = j + k < s1 ? j + k : (j + k <= last ? j + k : last);  // (a load past my row still hits the log: its event is replaced by the null event)
      typedef unsigned int v4u __attribute__((ext_vector_type(4)));
      const v4u w = __builtin_nontemporal_load((const v4u*)&p.events[at]);
      ev[k] = make_uint4(w.x, w.y, w.z, w.w);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tyc[k] = j + k < s1 ? type_off(ev[k].x) : kNullEntryOffBytes;
    // (rows of one or two events are the rule here: a step that no lane of the wave has an event for is not walked)
    if (__builtin_amdgcn_ballot_w64(j + 2 < s1) != 0ull) {
      walk_events_concrete<4>(a, presentM, frozenM, corr, ev, tyc, lds_tab, p);
    } else if (__builtin_amdgcn_ballot_w64(j + 1 < s1) != 0ull) {
      walk_events_concrete<2>(a, presentM, frozenM, corr, ev, tyc, lds_tab, p);
    } else {
      walk_events_concrete<1>(a, presentM, frozenM, corr, ev, tyc, lds_tab, p);
    }
  }
  finish_concrete(a, presentM, frozenM, corr, p);
  if (mine) store_state(p.out, r, a);
}

// ---- plan: task k owns segments [lower_bound(off, off[0] + k*T), lower_bound(off, off[0] + (k+1)*T)) ----
__global__ void plan_kernel(const int64_t* __restrict__ off, int64_t n_seg, int64_t task_events,
                            int64_t n_tasks, int64_t* __restrict__ plan) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n_tasks) return;
  if (k == n_tasks) { plan[k] = n_seg; return; }
  const int64_t target = off[0] + k * task_events;
  int64_t lo = 0, hi = n_seg;  // first s in [0, n_seg] with off[s] >= target
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (off[mid] < target) lo = mid + 1; else hi = mid;
  }
  plan[k] = lo;
}

// the same plan over a segment count that only the device knows yet (micro-batches: the group count the device group-by
// has just produced — no host round trip between the group-by and the fold)
__global__ void plan_dev_kernel(const int64_t* __restrict__ off, const uint32_t* __restrict__ d_n_seg, int64_t task_events,
                                int64_t n_tasks, int64_t* __restrict__ plan) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n_tasks) return;
  const int64_t n_seg = (int64_t)d_n_seg[0];
  if (k == n_tasks || n_seg == 0) { plan[k] = n_seg; return; }  // no segments: every task is empty
  const int64_t target = off[0] + k * task_events;
  int64_t lo = 0, hi = n_seg;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (off[mid] < target) lo = mid + 1; else hi = mid;
  }
  plan[k] = lo;
}

__global__ void analyze_csr_kernel(const int64_t* __restrict__ off, int64_t n_seg, CsrAnalysis* res) {
  __shared__ unsigned long long s_empty;
  __shared__ long long s_max;
  __shared__ int s_bad, s_nonuni;
  if (threadIdx.x == 0) { s_empty = 0; s_max = 0; s_bad = 0; s_nonuni = 0; }
  __syncthreads();
  const int64_t len0 = n_seg > 0 ? off[1] - off[0] : 0;
  unsigned long long empty = 0;
  long long mx = 0;
  int bad = 0, nonuni = 0;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_seg; s += (int64_t)gridDim.x * blockDim.x) {
    const int64_t len = off[s + 1] - off[s];
    if (len < 0) bad = 1;
    if (len == 0) ++empty;
    if (len != len0) nonuni = 1;
    if (len > mx) mx = len;
  }
  if (empty) atomicAdd(&s_empty, empty);
  if (mx) atomicMax(&s_max, mx);
  if (bad) atomicOr(&s_bad, 1);
  if (nonuni) atomicOr(&s_nonuni, 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_empty) atomicAdd((unsigned long long*)&res->n_empty, s_empty);
    if (s_max) atomicMax((long long*)&res->max_len, s_max);
    if (s_bad) atomicOr(&res->bad, 1);
    if (s_nonuni) atomicOr(&res->nonuniform, 1);
    if (blockIdx.x == 0) {
      res->len0 = len0;
      res->first = off[0];
      res->last = off[n_seg];
    }
  }
}

// Aggregates without events keep their prior snapshot (or stay None).
__global__ void fill_empty_kernel(const int64_t* __restrict__ off, int64_t n_seg, const uint4* __restrict__ init,
                                  uint4* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16 B quarter of a state per thread
  const int64_t s = i >> 2;
  if (s >= n_seg) return;
  if (off[s + 1] != off[s]) return;
  uint4 v;
  v.x = v.y = v.z = v.w = 0u;
  if (init) v = init[i];
  out[i] = v;
}

constexpr int kCompactBlock = 1024;

__global__ void __launch_bounds__(kCompactBlock) compact_count_kernel(const int64_t* __restrict__ off, int64_t n_seg,
                                                                       int64_t* __restrict__ block_counts) {
  __shared__ int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int64_t s = (int64_t)blockIdx.x * kCompactBlock + threadIdx.x;
  const bool nz = s < n_seg && off[s + 1] != off[s];
  const int c = __popcll(__ballot(nz));
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt;
}

// single block: exclusive scan of block_counts[0..nb) in place, total appended at [nb]
__global__ void __launch_bounds__(1024) compact_scan_kernel(int64_t* block_counts, int64_t nb) {
  __shared__ int64_t s_part[1024];
  const int tid = threadIdx.x;
  const int64_t per = (nb + 1023) / 1024;
  const int64_t b0 = tid * per, b1 = (b0 + per < nb) ? b0 + per : nb;
  int64_t sum = 0;
  for (int64_t b = b0; b < b1; ++b) sum += block_counts[b];
  s_part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int64_t run = 0;
    for (int i = 0; i < 1024; ++i) { const int64_t v = s_part[i]; s_part[i] = run; run += v; }
    block_counts[nb] = run;
  }
  __syncthreads();
  int64_t run = s_part[tid];
  for (int64_t b = b0; b < b1; ++b) { const int64_t v = block_counts[b]; block_counts[b] = run; run += v; }
}

__global__ void __launch_bounds__(kCompactBlock) compact_scatter_kernel(const int64_t* __restrict__ off, int64_t n_seg,
                                                                         const int64_t* __restrict__ block_counts,
                                                                         int64_t* __restrict__ nz_off,
                                                                         int64_t* __restrict__ nz_map) {
  __shared__ int s_wave[kCompactBlock / 64];
  const int64_t s = (int64_t)blockIdx.x * kCompactBlock + threadIdx.x;
  const bool nz = s < n_seg && off[s + 1] != off[s];
  const unsigned long long bal = __ballot(nz);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_wave[wave] = __popcll(bal);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  if (nz) {
    const int64_t r = block_counts[blockIdx.x] + base + __popcll(bal & ((1ull << lane) - 1ull));
    nz_off[r] = off[s];
    nz_map[r] = s;
  }
  const int64_t nb = (n_seg + kCompactBlock - 1) / kCompactBlock;
  if (blockIdx.x == 0 && threadIdx.x == 0) nz_off[block_counts[nb]] = off[n_seg];
}

}  // namespace

hipError_t launch_fold_fixed(const FoldParams& p, int64_t n_tasks, int lane_events, hipStream_t stream) {
  if (n_tasks <= 0) return hipSuccess;
  if (lane_events == 8)
    hipLaunchKernelGGL((fold_kernel<MODE_FIXED, 8>), dim3((unsigned)n_tasks), dim3(kWave), Geo<8>::lds_bytes(Geo<8>::kAuxFlat), stream, p);
  else
    hipLaunchKernelGGL((fold_kernel<MODE_FIXED, 16>), dim3((unsigned)n_tasks), dim3(kWave), Geo<16>::lds_bytes(Geo<16>::kAuxFlat), stream, p);
  return hipGetLastError();
}

hipError_t launch_fold_rows(const FoldParams& p, const V1Kernels* spec, int64_t n_tasks, int lane_events, hipStream_t stream) {
  if (n_tasks <= 0) return hipSuccess;
  if (spec && lane_events != 32) return launch_v1_lane(lane_events == 8 ? spec->rows8 : spec->rows16, &p, sizeof(p), n_tasks, 0, stream);
  static const bool conc = [] { const char* e = std::getenv("SURGE_REPLAY_WALK"); return !(e && std::strcmp(e, "transformer") == 0); }();
  // LDS is static in the kernel
  if (lane_events == 8) {
    if (conc) hipLaunchKernelGGL((fold_rows_kernel<8, true>), dim3((unsigned)n_tasks), dim3(kWave), 0, stream, p);
    else hipLaunchKernelGGL((fold_rows_kernel<8, false>), dim3((unsigned)n_tasks), dim3(kWave), 0, stream, p);
  } else if (lane_events == 32) {
    hipLaunchKernelGGL((fold_rows_kernel<32, false>), dim3((unsigned)n_tasks), dim3(kWave), 0, stream, p);
  } else {
    if (conc) hipLaunchKernelGGL((fold_rows_kernel<16, true>), dim3((unsigned)n_tasks), dim3(kWave), 0, stream, p);
    else hipLaunchKernelGGL((fold_rows_kernel<16, false>), dim3((unsigned)n_tasks), dim3(kWave), 0, stream, p);
  }
  return hipGetLastError();
}

hipError_t launch_fold_short(const FoldParams& p, hipStream_t stream) {
  if (p.n_seg <= 0) return hipSuccess;
  hipLaunchKernelGGL(fold_short_kernel, dim3((unsigned)((p.n_seg + 255) / 256)), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_fold_sorted(const FoldParams& p, int64_t n_waves, int lane_events, hipStream_t stream) {
  if (n_waves <= 0) return hipSuccess;
  if (lane_events == 8)
    hipLaunchKernelGGL((fold_sorted_kernel<8>), dim3((unsigned)n_waves), dim3(kWave), Geo<8>::lds_bytes(Geo<8>::kAuxSorted), stream, p);
  else if (lane_events == 32)
    hipLaunchKernelGGL((fold_sorted_kernel<32>), dim3((unsigned)n_waves), dim3(kWave), Geo<32>::lds_bytes(Geo<32>::kAuxSorted), stream, p);
  else
    hipLaunchKernelGGL((fold_sorted_kernel<16>), dim3((unsigned)n_waves), dim3(kWave), Geo<16>::lds_bytes(Geo<16>::kAuxSorted), stream, p);
  return hipGetLastError();
}

// ---- the flat kernel compiled for one v1 schema (hiprtc) ---------------------------------------------------------------
// What fold_slots.hip does for ABI v2, for the op table of ABI v1: the kernel's own device source (fold_flat_device.h),
// compiled once per distinct op table and process with the table's words as compile-time masks (fold_device.h,
// SURGE_V1_SPEC).  The flat kernel is the one of the v1 folds that is bound by its instruction stream (lane transformers:
// ~97 VALU instructions per event against 55 in the lane-per-aggregate kernels); the others run at the transport's pace.
namespace {

struct V1Masks {
  uint32_t word[kTableWords];  // [k] bit e: word k of entry e is non-zero (k < 15)
  uint32_t throws, deletes;    // TW_FLAGS bit 0 / bit 16 per entry
};

// false: a word holds something a bit per entry cannot express (not an op table fill_params writes)
bool v1_masks(const uint32_t (*table)[kTableWords], V1Masks* m) {
  std::memset(m, 0, sizeof(*m));
  for (int e = 0; e < kTableEntries; ++e) {
    for (int k = 0; k < TW_FLAGS; ++k) {
      const uint32_t w = table[e][k];
      if (w != 0u && w != (k == TW_EVC ? 1u : ~0u)) return false;
      if (w) m->word[k] |= 1u << e;
    }
    const uint32_t f = table[e][TW_FLAGS];
    if (f & ~0x10001u) return false;
    if (f & 1u) m->throws |= 1u << e;
    if (f & 0x10000u) m->deletes |= 1u << e;
  }
  return true;
}

std::mutex g_v1_mu;
std::map<std::string, std::unique_ptr<V1Kernels>> g_v1_cache;  // key: kind + device + masks; entries live as long as the process
std::map<std::string, std::string> g_v1_failed;

}  // namespace

// The program handed to hiprtc for one op table: the masks, then either the flat kernel (V1_FLAT: fold_flat_device.h — K3
// appends, AUTO on logs of few long rows) or the lane-per-row kernels (V1_LANES: fold_lane_device.h — SORTED, CHUNKED, ROWS).
// Two programs, not one: a handle that only ever appends micro-batches never pays for the lane kernels' compile and the
// other way round; each is cached on disk by its text (rtc.cpp).
std::string v1_spec_source(const uint32_t (*table)[kTableWords], int kind) {
  V1Masks m;
  if (!v1_masks(table, &m)) return std::string();
  std::string s = "#define SURGE_V1_SPEC 1\n#define SURGE_V1_MASK(k) (";
  char b[64];
  for (int k = 0; k < TW_FLAGS; ++k) {
    std::snprintf(b, sizeof b, "(k) == %d ? 0x%xu : ", k, m.word[k]);
    s += b;
  }
  s += "0u)\n";
  std::snprintf(b, sizeof b, "#define SURGE_V1_THROWS 0x%xu\n", m.throws);
  s += b;
  std::snprintf(b, sizeof b, "#define SURGE_V1_DELETES 0x%xu\n", m.deletes);
  s += b;
#ifdef SURGE_EXPERIMENTS  // experiment builds of the library only (scripts/build_experiments_lib.py): extra #defines for the program, "A=1;B=2"
  if (const char* extra = std::getenv("SURGE_REPLAY_RTC_EXTRA")) {
    std::string e = extra;
    size_t pos = 0;
    while (pos < e.size()) {
      size_t semi = e.find(';', pos);
      if (semi == std::string::npos) semi = e.size();
      std::string d = e.substr(pos, semi - pos);
      const size_t eq = d.find('=');
      if (!d.empty()) s += "#define " + (eq == std::string::npos ? d : d.substr(0, eq) + " " + d.substr(eq + 1)) + "\n";
      pos = semi + 1;
    }
  }
#endif
  if (kind == V1_FLAT) {
    s += R"SRC(
#include "fold_flat_device.h"
using namespace surge;
extern "C" __global__ void __launch_bounds__(64) surge_v1_flat8(const FoldParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fold_flat_body<MODE_FLAT, 8>(p, smem);
}
extern "C" __global__ void __launch_bounds__(64) surge_v1_flat16(const FoldParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fold_flat_body<MODE_FLAT, 16>(p, smem);
}
)SRC";
  } else {
    // (register budgets as in fold_chunked.hip: two waves per SIMD with 16 KiB tiles, three with 8 KiB tiles)
    s += R"SRC(
#include "fold_lane_device.h"
using namespace surge;
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) surge_v1_sorted8(const FoldParams p) {
  ChunkTable t;
  t.v_start = nullptr; t.v_len = nullptr; t.v_info = nullptr; t.v_dest = nullptr; t.n_vrows = 0; t.side = nullptr;
  chunk_walk<8, true, true>(p, t);
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) surge_v1_sorted16(const FoldParams p) {
  ChunkTable t;
  t.v_start = nullptr; t.v_len = nullptr; t.v_info = nullptr; t.v_dest = nullptr; t.n_vrows = 0; t.side = nullptr;
  chunk_walk<16, true, true>(p, t);
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1))) surge_v1_sorted32(const FoldParams p) {
  ChunkTable t;
  t.v_start = nullptr; t.v_len = nullptr; t.v_info = nullptr; t.v_dest = nullptr; t.n_vrows = 0; t.side = nullptr;
  chunk_walk<32, true, true>(p, t);
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) surge_v1_chunked8(const ChunkArgs a) {
  chunk_walk<8, false, true>(a.p, a.t);
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) surge_v1_chunked16(const ChunkArgs a) {
  chunk_walk<16, false, true>(a.p, a.t);
}
extern "C" __global__ void __launch_bounds__(64) surge_v1_rows8(const FoldParams p) {
  __shared__ __attribute__((aligned(16))) char lds_ev[Geo<8>::kTileBytes];
  fold_rows_body<8, true>(p, lds_ev, nullptr);
}
extern "C" __global__ void __launch_bounds__(64) surge_v1_rows16(const FoldParams p) {
  __shared__ __attribute__((aligned(16))) char lds_ev[Geo<16>::kTileBytes];
  fold_rows_body<16, true>(p, lds_ev, nullptr);
}
)SRC";
  }
  return s;
}

void v1_kernels_acquire(const uint32_t (*table)[kTableWords], int device, int kind, V1Kernels** out, double* compile_ms, std::string* why) {
  *out = nullptr;
  *compile_ms = 0.0;
  if (const char* v = std::getenv("SURGE_REPLAY_RTC")) {
    if (std::atoi(v) == 0) {
      *why = "disabled by SURGE_REPLAY_RTC=0";
      return;
    }
  }
  if (kind == V1_LANES) {
    // Opt-in (SURGE_REPLAY_RTC_LANES=1).  Measured in round 6 on the 10 M-aggregate log, one box, A/B/A
    // (profiles/r06_lane_spec_*.jsonl): SORTED takes 12.2 ms whatever its walk costs — compiled for the built-in schema (53 VALU
    // per event), for the Counter fixture's own schema (20), or with the arithmetic REMOVED (an experiment build that only moves
    // the events): 12.19 / 12.22 / 12.20 ms, at 8- and 16-event lanes, from six resident waves per CU up.  The kernel runs at
    // the pace of its transport — 64 row pieces per load instruction, gathered from anywhere in the log — not of its
    // instruction stream; compiling the walk buys nothing there (CHUNKED on one GPU's C4 shard: -1 %; ROWS on C2: +1.5 %).
    // So the default stays the ahead-of-time kernels; the compiled ones are kept, fuzzed against the oracle, for hosts whose
    // schemas or shapes differ.  (Their first version was 13 - 29 % SLOWER: hiprtc's compiler puts an s_waitcnt vmcnt(0) in front
    // of every LDS read that follows an LDS-DMA load, which serialised the sixteen loads of a tile — fold_lane_device.h, issue().)
    const char* v = std::getenv("SURGE_REPLAY_RTC_LANES");
    if (!v || std::atoi(v) == 0) {
      *why = "off (SURGE_REPLAY_RTC_LANES=1 compiles them)";
      return;
    }
  }
  V1Masks m;
  if (!v1_masks(table, &m)) {
    *why = "the op table holds words the specialised build cannot express";
    return;
  }
  std::string key((const char*)&m, sizeof(m));
  key += "@" + std::to_string(device) + "/" + std::to_string(kind);
#ifdef SURGE_EXPERIMENTS
  if (const char* extra = std::getenv("SURGE_REPLAY_RTC_EXTRA")) key += std::string("+") + extra;
#endif
  std::lock_guard<std::mutex> lk(g_v1_mu);
  auto hit = g_v1_cache.find(key);
  if (hit != g_v1_cache.end()) {
    *out = hit->second.get();
    *compile_ms = hit->second->compile_ms;
    return;
  }
  auto miss = g_v1_failed.find(key);
  if (miss != g_v1_failed.end()) {
    *why = miss->second;
    return;
  }
  auto give_up = [&](const std::string& msg) {
    g_v1_failed[key] = msg;
    *why = msg;
  };
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return give_up("hipGetDeviceProperties failed");
  std::string arch = prop.gcnArchName;  // "gfx950:sramecc+:xnack-"
  const size_t colon = arch.find(':');
  if (colon != std::string::npos) arch.resize(colon);
  std::vector<char> code;
  std::string log;
  double ms = 0.0;
  if (!rtc_compile(v1_spec_source(table, kind), arch.c_str(), &code, &log, &ms)) return give_up(log);
  auto k = std::make_unique<V1Kernels>();
  k->device = device;
  k->compile_ms = ms;
  hipError_t e = hipModuleLoadData(&k->module, code.data());
  if (e != hipSuccess) return give_up(std::string("hipModuleLoadData: ") + hipGetErrorString(e));
  struct Fn { hipFunction_t* f; const char* name; };
  std::vector<Fn> fns;
  if (kind == V1_FLAT) fns = {{&k->flat8, "surge_v1_flat8"}, {&k->flat16, "surge_v1_flat16"}};
  else fns = {{&k->sorted8, "surge_v1_sorted8"}, {&k->sorted16, "surge_v1_sorted16"}, {&k->sorted32, "surge_v1_sorted32"}, {&k->chunked8, "surge_v1_chunked8"},
              {&k->chunked16, "surge_v1_chunked16"}, {&k->rows8, "surge_v1_rows8"}, {&k->rows16, "surge_v1_rows16"}};
  for (auto& f : fns) {
    e = hipModuleGetFunction(f.f, k->module, f.name);
    if (e != hipSuccess) {
      (void)hipModuleUnload(k->module);
      return give_up(std::string("hipModuleGetFunction(") + f.name + "): " + hipGetErrorString(e));
    }
  }
  *compile_ms = ms;
  *out = k.get();
  g_v1_cache[key] = std::move(k);
}

// one launch of a kernel of a V1_LANES / V1_FLAT module: the argument block by value
hipError_t launch_v1_lane(hipFunction_t fn, const void* args, size_t args_bytes, int64_t grid, unsigned lds, hipStream_t stream) {
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void*>(args), HIP_LAUNCH_PARAM_BUFFER_SIZE, &args_bytes, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, kWave, 1, 1, lds, stream, nullptr, config);
}

hipFunction_t v1_lane_fn(const V1Kernels* k, int which, int lane_events) {
  if (!k) return nullptr;
  if (which == 0) return lane_events == 8 ? k->sorted8 : k->sorted16;
  if (which == 1) return lane_events == 8 ? k->chunked8 : k->chunked16;
  return lane_events == 8 ? k->rows8 : k->rows16;
}

hipError_t launch_fold_flat(const FoldParams& p, const V1Kernels* spec, int64_t n_tasks, int lane_events, hipStream_t stream) {
  if (n_tasks <= 0) return hipSuccess;
  if (spec) {  // no op table in LDS: the tile and the head bitmask only
    FoldParams args = p;
    size_t args_bytes = sizeof(args);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &args_bytes, HIP_LAUNCH_PARAM_END};
    const unsigned lds = lane_events == 8 ? Geo<8>::kTileBytes + Geo<8>::kAuxFlat : Geo<16>::kTileBytes + Geo<16>::kAuxFlat;
    return hipModuleLaunchKernel(lane_events == 8 ? spec->flat8 : spec->flat16, (unsigned)n_tasks, 1, 1, kWave, 1, 1, lds, stream, nullptr, config);
  }
  if (lane_events == 8)
    hipLaunchKernelGGL((fold_kernel<MODE_FLAT, 8>), dim3((unsigned)n_tasks), dim3(kWave), Geo<8>::lds_bytes(Geo<8>::kAuxFlat), stream, p);
  else
    hipLaunchKernelGGL((fold_kernel<MODE_FLAT, 16>), dim3((unsigned)n_tasks), dim3(kWave), Geo<16>::lds_bytes(Geo<16>::kAuxFlat), stream, p);
  return hipGetLastError();
}

hipError_t launch_plan(const int64_t* off, int64_t n_seg, int64_t task_events, int64_t n_tasks, int64_t* plan,
                       hipStream_t stream) {
  const int64_t n = n_tasks + 1;
  hipLaunchKernelGGL(plan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, off, n_seg, task_events,
                     n_tasks, plan);
  return hipGetLastError();
}

hipError_t launch_plan_dev(const int64_t* off, const uint32_t* d_n_seg, int64_t task_events, int64_t n_tasks, int64_t* plan,
                           hipStream_t stream) {
  const int64_t n = n_tasks + 1;
  hipLaunchKernelGGL(plan_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, off, d_n_seg, task_events, n_tasks, plan);
  return hipGetLastError();
}

hipError_t launch_analyze_csr(const int64_t* off, int64_t n_seg, CsrAnalysis* d_result, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_result, 0, sizeof(CsrAnalysis), stream);
  if (e != hipSuccess) return e;
  int64_t blocks = (n_seg + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(analyze_csr_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, off, n_seg, d_result);
  return hipGetLastError();
}

hipError_t launch_fill_empty(const int64_t* off, int64_t n_seg, const uint4* init, uint4* out, hipStream_t stream) {
  if (n_seg <= 0) return hipSuccess;
  const int64_t n = n_seg * 4;
  hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, off, n_seg, init, out);
  return hipGetLastError();
}

// in-place exclusive scan of v[0..n) with the total appended at v[n] (single block: index-time use only)
hipError_t launch_exclusive_scan_i64(int64_t* v, int64_t n, hipStream_t stream) {
  if (n < 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, stream, v, n);
  return hipGetLastError();
}

hipError_t launch_compact_nonempty(const int64_t* off, int64_t n_seg, int64_t* d_block_counts, int64_t* nz_off,
                                   int64_t* nz_map, hipStream_t stream) {
  if (n_seg <= 0) return hipSuccess;
  const int64_t nb = (n_seg + kCompactBlock - 1) / kCompactBlock;
  hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nb), dim3(kCompactBlock), 0, stream, off, n_seg, d_block_counts);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, stream, d_block_counts, nb);
  hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)nb), dim3(kCompactBlock), 0, stream, off, n_seg,
                     d_block_counts, nz_off, nz_map);
  return hipGetLastError();
}

}  // namespace surge
