// fold_tiled.hip — K2t "tile-major": the fold over a re-laid-out copy of the log in which every tile a wave fetches
// is ONE contiguous run of memory.
//
// Why (measured in round 2): the lane-per-aggregate kernels over the CSR log (rows / sorted / chunked)
// fetch, per tile, 64 separate row pieces of 128–256 bytes from 64 unrelated places of the log and top out at
// 5.7–6.05 TB/s, while a LINEAR stream of the same LDS-DMA loads reaches 7.0–7.2 TB/s on the same chip.  The engine owns
// the layout of a bound log (it builds the length order and the chunk table anyway), so it can also own the order of
// the bytes: here the virtual rows of the chunk table (whole aggregates, or chunks of aggregates longer than T — exactly
// fold_chunked.hip's) are copied ONCE into group-major / tile-major order:
//
//   group g   = 64 virtual rows of (almost) equal length, longest groups first (the chunk table's order)
//   subtile c = events [8c, 8c+8) of each of the group's 64 rows: 64 x 8 x 16 B = 8 KiB, stored as the exact LDS image
//               the walk wants (row of lane l at byte 128 l, its event j at slot j ^ key(l): the XOR swizzle that makes
//               the later ds_read_b128 of "event j of lane l" bank-conflict free is applied by the re-layout, not by
//               the loads)
//   the subtiles of a group are consecutive, the groups are consecutive: a wave that owns group g streams
//   [g_sub0[g], g_sub0[g+1]) x 8 KiB front to back with global_load_lds_dwordx4, every instruction one fully used,
//   contiguous 1 KiB, every byte of the copy read exactly once per fold.
//
// Rows shorter than their group's longest are padded to it with PAD events (type 0xffffffff; never applied: the walk
// masks them by the row length, and an unmasked one would poison its aggregate loudly rather than count as a NOOP).
// Rows are sorted by length, so the padding is the rounding of each row to 8 events: < 1 % of the bytes on the
// Zipf(1..4096) logs.  What a fold reads beyond the algorithmic bytes is that padding; the one-off cost is the copy
// itself (read the CSR log once, write the tiled log once: reported by surge_replay_layout_info, never hidden in a
// fold's time).
//
// Everything else — one lane per virtual row, concrete running state for whole aggregates, the P / S presence split
// and the 80-byte side entries for chunks of cut aggregates, the stitch kernel, the self re-arming group dispenser, the
// mask-arithmetic event walk — is fold_chunked.hip's (fold_chunk_device.h); results are bit-identical to the
// sequential fold for the same reasons.
#include <type_traits>

#include "fold_chunk_device.h"

namespace surge {
namespace {

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// one wave per group: subtiles of the group (from its longest row)
__global__ void __launch_bounds__(256) tile_index_kernel(const uint32_t* __restrict__ v_len, int64_t n_vrows, int64_t n_groups,
                                                         int64_t* __restrict__ g_sub) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n_groups) return;
  const int64_t row = g * kWave + lane;
  uint32_t mx = row < n_vrows ? v_len[row] : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
  if (lane == 0) g_sub[g] = (int64_t)((mx + kSubEvents - 1) / kSubEvents);
}

// The re-layout: 256-thread blocks, each owning a RUN of consecutive subtiles (one search for the run's first group, then
// the groups are walked forward), two 16-byte slots per thread per subtile and two subtiles in flight per thread.
// Output slot `pos` of a subtile belongs to lane l = pos / 8 and holds its event j = (pos % 8) ^ key(l) — Geo<8>'s
// swizzle.  Writes are linear (8 KiB per subtile); reads are 64 row pieces of 128 bytes, each covered by 8 adjacent threads.
// (Round 3, first version: one search of 17 dependent loads per 8 KiB subtile and 9 M short-lived block iterations —
// 4.0 TB/s of combined traffic on the 74 GB log.)
__global__ void __launch_bounds__(256) relayout_kernel(const uint4* __restrict__ events, const int64_t* __restrict__ v_start,
                                                       const uint32_t* __restrict__ v_len, int64_t n_vrows,
                                                       const int64_t* __restrict__ g_sub0, int64_t n_groups, int64_t n_sub_total,
                                                       int64_t subs_per_block, uint4* __restrict__ tiles) {
  const int64_t sub_begin = (int64_t)blockIdx.x * subs_per_block;
  int64_t sub_end = sub_begin + subs_per_block;
  sub_end = sub_end < n_sub_total ? sub_end : n_sub_total;
  if (sub_begin >= sub_end) return;
  // the group that owns the run's first subtile: the last g with g_sub0[g] <= sub (block-uniform search)
  int64_t g;
  {
    int64_t lo = 0, hi = n_groups;
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (g_sub0[mid] <= sub_begin) lo = mid; else hi = mid;
    }
    g = lo;
  }
  auto fetch = [&](int64_t sub, int64_t gg, int k) -> v4u {
    const uint32_t e0 = (uint32_t)(sub - g_sub0[gg]) * kSubEvents;
    const int pos = threadIdx.x + 256 * k;
    const int l = pos >> 3;
    const uint32_t j = (uint32_t)(pos & 7) ^ Geo<8>::key(l);
    const int64_t row = gg * kWave + l;
    v4u v = {kNullEntryOffBytes, 0u, 0u, 0u};  // PAD: the null entry of the op table (identity on every state)
    if (row < n_vrows) {
      const uint32_t len = v_len[row];
      if (e0 + j < len) {
        v = __builtin_nontemporal_load((const v4u*)(events + v_start[row] + e0 + j));
        v.x = type_off(v.x);  // the type word becomes the byte offset of its op-table entry (unknown types: the poison entry)
      }
    }
    return v;
  };
  for (int64_t sub = sub_begin; sub < sub_end; sub += 2) {
    while (g + 1 < n_groups && g_sub0[g + 1] <= sub) ++g;
    int64_t g1 = g;
    const bool two = sub + 1 < sub_end;
    if (two)
      while (g1 + 1 < n_groups && g_sub0[g1 + 1] <= sub + 1) ++g1;
    const v4u a0 = fetch(sub, g, 0), a1 = fetch(sub, g, 1);
    v4u b0 = a0, b1 = a1;
    if (two) { b0 = fetch(sub + 1, g1, 0); b1 = fetch(sub + 1, g1, 1); }
    v4u* o = (v4u*)(tiles + sub * (kSubBytes / 16));
    __builtin_nontemporal_store(a0, o + threadIdx.x);
    __builtin_nontemporal_store(a1, o + threadIdx.x + 256);
    if (two) {
      __builtin_nontemporal_store(b0, o + 512 + threadIdx.x);
      __builtin_nontemporal_store(b1, o + 512 + threadIdx.x + 256);
    }
    g = g1;
  }
}

// SUBS subtiles per step: 1 = 8 events per lane per step (8 KiB in flight per wave, up to 4 waves per SIMD), 2 = 16 events
// (16 KiB, up to 3 waves per SIMD).
template <int SUBS>
__global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(SUBS == 1 ? 3 : 2)))
fold_tiled_kernel(const FoldParams p, const TileTable t) {
  constexpr int LE = kSubEvents * SUBS;
  constexpr int kLoads = SUBS * (kSubBytes / 1024);
  __shared__ __attribute__((aligned(16))) char lds_ev[SUBS * kSubBytes];
  __shared__ __attribute__((aligned(16))) uint32_t lds_tab[kTableLdsDwords];
  const int lane = threadIdx.x;
  load_table<8>(p, lds_tab, lane);
  const uint32_t ev_row = Geo<8>::ev_row(lane);  // my row inside a subtile; event j of the subtile at ev_row ^ (j * 16)
  const int64_t n_groups = (t.n_vrows + kWave - 1) / kWave;

  auto grab = [&]() -> int64_t {
    unsigned long long g = 0;
    if (lane == 0) g = atomicAdd(p.counter, 1ull);
    return (int64_t)(((uint64_t)rl((uint32_t)(g >> 32), 0) << 32) | rl((uint32_t)g, 0));
  };
  struct Meta { int64_t dest; uint32_t len, info; };
  auto load_meta = [&](int64_t g) -> Meta {
    Meta m; m.dest = -1; m.len = 0u; m.info = 0u;
    const int64_t idx = g * kWave + lane;
    if (g < n_groups && idx < t.n_vrows) {
      m.dest = t.v_dest[idx]; m.len = t.v_len[idx]; m.info = t.v_info[idx];
    }
    return m;
  };
  struct Shape { int64_t sub0; int n_sub; };  // wave-uniform
  auto load_shape = [&](int64_t g) -> Shape {  // every lane reads the same words; made scalar for the buffer descriptor
    Shape s; s.sub0 = 0; s.n_sub = 0;
    if (g < n_groups) {
      const int64_t a = t.g_sub0[g], b = t.g_sub0[g + 1];
      s.sub0 = uniform64(a);
      s.n_sub = (int)__builtin_amdgcn_readfirstlane((uint32_t)(b - a));
    }
    return s;
  };
  // Step c of a group = its subtiles [SUBS c, SUBS c + SUBS): one linear run, fetched with buffer_load_dwordx4 ... lds
  // (descriptor base = the step's first byte, lane offset 16 l, instruction q at scalar offset 1024 q; MUBUF rather than
  // global_load_lds for the exact lgkm waits it leaves in the walk: see issue_tile_loads in fold_device.h).
  const int voff = lane * 16;
  auto issue = [&](const Shape& s, int c) {
    const char* base = (const char*)t.tiles + (s.sub0 + (int64_t)c * SUBS) * kSubBytes;  // wave-uniform
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const int n = (SUBS == 2 && c * 2 + 1 >= s.n_sub) ? kLoads / 2 : kLoads;  // an odd last subtile: half a step
#pragma unroll
    for (int q = 0; q < kLoads; ++q)
      if (q < n) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds_ev + q * 1024), 16, voff, q * 1024, 0, kLoadAux);
  };

  int64_t g = grab();
  Meta cur = load_meta(g);
  Shape sh = load_shape(g);
  if (g < n_groups) issue(sh, 0);
  while (g < n_groups) {
    const int64_t g_next = grab();
    const Meta nxt = load_meta(g_next);  // in flight while this group is walked
    const Shape sh_next = load_shape(g_next);
    const int n_steps = (sh.n_sub + SUBS - 1) / SUBS;
    const bool odd_tail = SUBS == 2 && (sh.n_sub & 1);  // the last step holds one subtile: its second half is stale

    const bool whole = (cur.info & VI_RELATIVE) == 0u;
    // an aggregate in one piece starts from its known state, a chunk from "whatever comes in" (relative)
    Acc a = whole ? ((p.init && cur.dest >= 0) ? load_state(p.init, cur.dest) : acc_none()) : acc_identity();
    uint32_t frozenM = whole ? (uint32_t)__builtin_amdgcn_sbfe((int32_t)a.fl, 1, 1) : 0u;
    uint32_t corr = 0u;
    const bool any_relative = __builtin_amdgcn_ballot_w64(!whole) != 0ull;  // wave-uniform

    // one step: wait for it, pull my N events out of LDS, start the next fetch, walk.  The type word of an event in the
    // tile-major log IS the byte offset of its op-table entry (PAD slots: the null entry), so there is no decode and no
    // tail masking here.
    auto fetch_next = [&](int c) {
      if (c + 1 < n_steps) {
        issue(sh, c + 1);
      } else if (g_next < n_groups) {
        issue(sh_next, 0);  // the next group's first step is fetched while this group's last one is walked
      }
    };
    auto read_events = [&](uint4* ev, auto n_tag) {
      constexpr int N = decltype(n_tag)::value;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < N; ++j) ev[j] = *(const uint4*)(lds_ev + (j >> 3) * kSubBytes + (ev_row ^ (uint32_t)((j & 7) * 16)));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    if (!any_relative) {
      // every lane walks a whole aggregate from its known state: the concrete walk (no transformer bookkeeping)
      uint32_t presentM = (uint32_t)__builtin_amdgcn_sbfe((int32_t)a.fl, 0, 1);
      auto step = [&](int c, auto n_tag) {
        constexpr int N = decltype(n_tag)::value;
        uint4 ev[N];
        read_events(ev, n_tag);
        fetch_next(c);
        uint32_t tyc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) tyc[j] = ev[j].x;
        walk_events_concrete<N>(a, presentM, frozenM, corr, ev, tyc, lds_tab, p);
      };
      const int full = odd_tail ? n_steps - 1 : n_steps;
      for (int c = 0; c < full; ++c) step(c, std::integral_constant<int, LE>{});
      if (odd_tail) step(n_steps - 1, std::integral_constant<int, kSubEvents>{});
      a.sum = (int64_t)((uint64_t)a.sum + corr);
      a.fl = (presentM & FL_PRESENT) | (frozenM & FL_POISONED);
      if (cur.dest >= 0) store_state(p.out, cur.dest, a);
    } else {
      Acc P = acc_identity();
      uint32_t undecM = whole ? 0u : ~0u;
      bool watching = true;
      auto step = [&](int c, auto n_tag, auto tracking) {
        constexpr int N = decltype(n_tag)::value;
        uint4 ev[N];
        read_events(ev, n_tag);
        fetch_next(c);
        uint32_t tyc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) tyc[j] = ev[j].x;
        if constexpr (decltype(tracking)::value) {
          walk_events_track<N>(a, P, undecM, frozenM, corr, ev, tyc, lds_tab, p);
          watching = __builtin_amdgcn_ballot_w64(undecM != 0u && frozenM == 0u) != 0ull;
        } else {
          walk_events<N, false>(a, frozenM, corr, ev, tyc, 0u, lds_tab, p, [](int) {});
        }
      };
      // Two loops, not one loop with a branch (fold_chunked.hip): the watching loop — usually the first step only —
      // carries P and the deciding-event test; the plain loop is the transformer walk without it.
      const int full = odd_tail ? n_steps - 1 : n_steps;
      int c = 0;
      for (; c < full && watching; ++c) step(c, std::integral_constant<int, LE>{}, std::true_type{});
      for (; c < full; ++c) step(c, std::integral_constant<int, LE>{}, std::false_type{});
      if (odd_tail) step(n_steps - 1, std::integral_constant<int, kSubEvents>{}, std::true_type{});
      a.sum = (int64_t)((uint64_t)a.sum + corr);

      if (cur.dest >= 0) {
        if (cur.info & VI_SIDE) {
          const bool empty = cur.len == 0u;
          const bool undecided = undecM != 0u || empty;
          const Acc Pw = empty ? acc_identity() : select_acc(undecided, a, P);
          Acc Sw = select_acc(undecided, acc_identity(), a);
          if (!undecided) Sw.fl |= SIDE_DECIDED;
          store_side(t.side, cur.dest, Pw, Sw);
        } else {
          store_state(p.out, cur.dest, a);
        }
      }
    }

    g = g_next;
    cur = nxt;
    sh = sh_next;
  }
  dispenser_leave(p.counter, lane);
}

}  // namespace

// g_sub[g] := subtiles of group g (n_groups entries; the caller scans them into offsets)
hipError_t launch_tile_index(const uint32_t* v_len, int64_t n_vrows, int64_t* g_sub, hipStream_t stream) {
  const int64_t n_groups = (n_vrows + kWave - 1) / kWave;
  if (n_groups <= 0) return hipSuccess;
  hipLaunchKernelGGL(tile_index_kernel, dim3((unsigned)((n_groups + 3) / 4)), dim3(256), 0, stream, v_len, n_vrows, n_groups, g_sub);
  return hipGetLastError();
}

hipError_t launch_relayout(const uint4* events, const int64_t* v_start, const uint32_t* v_len, int64_t n_vrows, const int64_t* g_sub0,
                           int64_t n_sub_total, uint4* tiles, hipStream_t stream) {
  const int64_t n_groups = (n_vrows + kWave - 1) / kWave;
  if (n_groups <= 0 || n_sub_total <= 0) return hipSuccess;
  // runs of 16 subtiles (128 KiB) per block, more when that would exceed a 2^20-block grid
  int64_t per = 16;
  while ((n_sub_total + per - 1) / per > (1ll << 20)) per *= 2;
  const int64_t blocks = (n_sub_total + per - 1) / per;
  hipLaunchKernelGGL(relayout_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, events, v_start, v_len, n_vrows, g_sub0, n_groups,
                     n_sub_total, per, tiles);
  return hipGetLastError();
}

hipError_t launch_fold_tiled(const FoldParams& p, const uint4* tiles, const int64_t* g_sub0, const uint32_t* v_len,
                             const uint32_t* v_info, const int64_t* v_dest, int64_t n_vrows, uint32_t* side, int64_t n_waves, int subs,
                             hipStream_t stream) {
  if (n_waves <= 0 || n_vrows <= 0) return hipSuccess;
  TileTable t;
  t.tiles = tiles; t.g_sub0 = g_sub0; t.v_len = v_len; t.v_info = v_info; t.v_dest = v_dest;
  t.n_vrows = n_vrows; t.side = side;
  if (subs == 1)
    hipLaunchKernelGGL((fold_tiled_kernel<1>), dim3((unsigned)n_waves), dim3(kWave), 0, stream, p, t);
  else
    hipLaunchKernelGGL((fold_tiled_kernel<2>), dim3((unsigned)n_waves), dim3(kWave), 0, stream, p, t);
  return hipGetLastError();
}

}  // namespace surge
