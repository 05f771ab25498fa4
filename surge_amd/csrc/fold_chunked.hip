// fold_chunked.hip — K2c "chunked rows": any CSR, one lane per CHUNK of an aggregate's events.
//
// Why (round-1 profile): in the sorted-rows kernel one lane walks one whole aggregate, so a group
// of 4096-event aggregates is ~1 ms of critical path for ONE wave, whatever the log size: a 1.25 M-aggregate
// Zipf shard (what each GPU of the 8-GPU config holds) ran at 55–62 % of the HBM roofline while the 10 M-aggregate
// log reached 73–77 %, and logs under ~0.5 M aggregates fell to 18–37 %.  Here an aggregate longer than T events
// is cut into c = ceil(len / T) chunks of equal length (boundaries on 128-byte lines).  Every chunk is an
// independent "virtual row": the virtual rows of the whole log are ordered by length like the sorted-rows kernel
// orders aggregates, 64 of them per wave, so no wave ever walks more than ~T events alone.  A chunk of a cut
// aggregate leaves a 80-byte summary in a side buffer; a second, tiny kernel stitches each cut aggregate left to
// right.  (Putting the chunks of one aggregate on ADJACENT lanes of one wave and stitching with shuffles — no side
// buffer, no second kernel — was built and measured first: 4–6 % slower per tile at every size; a wave whose lanes
// stream 64 unrelated rows is what the memory system likes.)
//
// T is chosen per log by the host (engine.hip): about algorithmic_bytes / 6 MB — the longest chunk's walk stays a
// small fraction of the whole kernel — and the engine falls back to the sorted-rows kernel when T >= the longest
// aggregate (nothing to cut).  Measured on MI355X, Zipf(1..4096) logs, % of 8 TB/s (FLAT / SORTED / CHUNKED):
// 0.3 M aggregates 50 / 26 / 57, 0.5 M 52 / 37 / 62, 0.8 M 54 / 48 / 64, 1.25 M 54 / 61 / 68, 2 M 57 / 69 / 70.
//
// foldLeft is sequential; a chunk that does not start its aggregate does not know its incoming state.  It still
// needs only ONE evaluation path per lane (no "both hypotheses" double work):
//   * fields compose as transformers (delta / clamp / set), exactly like the flat kernel's lane transformers;
//   * PRESENCE (Some/None decides whether REQUIRE-class events apply and whether MATERIALIZE resets to the
//     defaults) is split at the chunk's first live event that is NOT of class REQUIRE ("deciding" event):
//       P = the events before it  — all REQUIRE-class: applied iff the incoming state is Some, skipped if None
//       S = the deciding event and everything after it, walked as "state is Some" — which is what it is in
//           BOTH cases: after P when the incoming state was Some, after materialising the defaults when it was
//           None (CREATE / DELETE make S absolute and the incoming state irrelevant)
//     so   result = incoming is Some ? S(P(incoming)) : (decided ? S(defaults) : None).
//   * a throwing event freezes the walk in either case at the same event (throwing does not depend on presence).
// An aggregate in one piece (the vast majority) is walked concretely from its prior state and stores its own
// 64-byte result, exactly as in the sorted-rows kernel.
//
// The event transport (LDS-DMA tiles, XOR swizzle, line-aligned row pieces, LDS row table) and the per-event mask
// arithmetic are the sorted-rows kernel's (fold_device.h).  Integer adds wrap exactly like JVM Int/Long under
// any association and min/max/set are exact, so the result is bit-identical to the sequential fold.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "fold_chunk_device.h"

namespace surge {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// second kernel: one thread per cut aggregate composes its chunk summaries left to right onto the prior state
__global__ void chunk_stitch_kernel(const FoldParams p, const uint32_t* __restrict__ side, const int64_t* __restrict__ r_slot0,
                                    const uint32_t* __restrict__ r_c, const int64_t* __restrict__ r_out, int64_t n_rows) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const int64_t slot0 = r_slot0[r];
  const uint32_t c = r_c[r];
  const int64_t oi = r_out[r];
  Acc x = p.init ? load_state(p.init, oi) : acc_none();  // the aggregate's prior state (or None)
  for (uint32_t k = 0; k < c; ++k) {
    uint32_t w[kSideDwords];
    const uint4* q = (const uint4*)(side + (slot0 + k) * kSideDwords);
#pragma unroll
    for (int i = 0; i < 5; ++i) { const uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
    const Acc P = acc_from_words(w);
    Acc S = acc_from_words(w + 10);
    const bool decided = (S.fl & SIDE_DECIDED) != 0u;
    S.fl &= ~SIDE_DECIDED;
    x = resolve_chunk(x, P, S, decided, p);
  }
  store_state(p.out, oi, x);
}

// The walk of both kernels of this file.  PERM = false: the virtual rows of the chunk table (fold_chunked_kernel).
// PERM = true: whole aggregates in length order straight from the CSR arrays — row i is aggregate perm[i] (p.plan), its
// start and length come from seg_off like in fold_sorted_kernel (fold_kernels.hip); nothing is relative, nothing goes to the
// side buffer, and the deciding-event loop compiles away.  What it has over fold_sorted_kernel is this file's pipeline: the
// next group's first tile is fetched during the current group's last tile (no wait for a cold tile at every group switch).
template <int LE, bool PERM, bool CONC>
__device__ __forceinline__ void chunk_walk(const FoldParams& p, const ChunkTable& t) {
  using G = Geo<LE>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_ev = smem;
  int64_t* lds_rs = (int64_t*)(smem + G::kTileBytes);                  // 64 chunk starts ...
  uint32_t* lds_len = (uint32_t*)(smem + G::kTileBytes + kWave * 8);   // ... and 64 chunk lengths
  uint32_t* lds_tab = (uint32_t*)(smem + G::kTileBytes + G::kAuxSorted);
  const int lane = threadIdx.x;
  load_table<LE>(p, lds_tab, lane);
  const uint32_t ev_row = G::ev_row(lane);
  const int64_t n_rows = PERM ? p.n_seg : t.n_vrows;
  const int64_t n_groups = (n_rows + kWave - 1) / kWave;

  auto grab = [&]() -> int64_t {
    unsigned long long g = 0;
    if (lane == 0) g = atomicAdd(p.counter, 1ull);
    return (int64_t)(((uint64_t)rl((uint32_t)(g >> 32), 0) << 32) | rl((uint32_t)g, 0));
  };
  struct Meta { int64_t dest, start; uint32_t len, info; };
  auto load_meta = [&](int64_t g) -> Meta {
    Meta m; m.dest = -1; m.start = 0; m.len = 0u; m.info = 0u;
    const int64_t idx = g * kWave + lane;
    if (g < n_groups && idx < n_rows) {
      if constexpr (PERM) {
        // a row is tiled from the 128-byte line that holds its first event: the events in front of it (its predecessor's) are
        // walked as null events (fold_sorted_kernel's rule)
        const int64_t sg = p.plan[idx];
        const int64_t st = p.seg_off[sg];
        const uint32_t pad = (uint32_t)(st & 7);
        m.dest = p.out_map ? p.out_map[sg] : sg;
        m.start = st - pad;
        m.len = (uint32_t)(p.seg_off[sg + 1] - st) + pad;
        m.info = pad << VI_PAD_SHIFT;
      } else {
        m.dest = t.v_dest[idx]; m.start = t.v_start[idx]; m.len = t.v_len[idx]; m.info = t.v_info[idx];
      }
    }
    return m;
  };
  // longest / shortest non-empty chunk of a group (empty chunks do not bound the fast path) and its tile count
  struct Shape { uint32_t maxlen, minlen; int n_tiles; };
  auto shape_of = [&](const Meta& m) -> Shape {
    Shape sh;
    sh.maxlen = m.len;
    sh.minlen = m.len ? m.len : 0xffffffffu;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      sh.maxlen = max(sh.maxlen, (uint32_t)__shfl_xor((int)sh.maxlen, d, 64));
      sh.minlen = min(sh.minlen, (uint32_t)__shfl_xor((int)sh.minlen, d, 64));
    }
    sh.n_tiles = (int)((sh.maxlen + LE - 1) / LE);
    return sh;
  };
  // the chunk starts / lengths each load instruction needs (chunk RPL*q + lane/LE) live in a small LDS table
  auto publish = [&](const Meta& m) {
    lds_rs[lane] = m.start;
    lds_len[lane] = m.len;
  };
  // The address of a load is events + 16 * (chunk start + tile offset + the lane's event of the instruction's class): the
  // per-lane part is kept as kClasses 64-bit bases the compiler cannot take apart (three inlined copies of this lambda each
  // hoisted their own variants of it out of the loops — 42 VGPRs of loop invariants, ten of them spilled to scratch).
  uint64_t ebase[G::kClasses];
#pragma unroll
  for (int k = 0; k < G::kClasses; ++k) {
    ebase[k] = (uint64_t)p.events + 16ull * G::load_j(lane, k);
    asm volatile("" : "+v"(ebase[k]));
  }
  auto issue = [&](int c, uint32_t minlen) {
    if ((uint32_t)(c + 1) * LE <= minlen) {
      const uint64_t coff = (uint64_t)(uint32_t)c * (uint32_t)(LE * 16);
#pragma unroll
      for (int q = 0; q < G::kLoads; ++q) {
        const uint64_t a = ebase[q % G::kClasses] + coff + ((uint64_t)lds_rs[G::kRowsPerLoad * q + lane / LE] << 4);
        __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)(lds_ev + q * 1024), 16, 0, kLoadAux);
      }
    } else {  // some chunk ends inside this tile: never read past a chunk's own events
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));  // (this path is rare: its per-lane constants are computed here, not kept in registers)
#pragma unroll
      for (int q = 0; q < G::kLoads; ++q) {
        const int r = G::kRowsPerLoad * q + lane_o / LE;
        const uint32_t rlen = lds_len[r];
        uint32_t j = (uint32_t)c * LE + G::load_j(lane_o, q % G::kClasses);
        const uint32_t lastj = rlen ? rlen - 1u : 0u;
        j = j < lastj ? j : lastj;
        __builtin_amdgcn_global_load_lds((gptr_t)(p.events + (lds_rs[r] + j)), (lptr_t)(lds_ev + q * 1024), 16, 0, kLoadAux);
      }
    }
  };

  int64_t g = grab();
  Meta cur = load_meta(g);
  Shape sh = shape_of(cur);
  publish(cur);
  if (g < n_groups) issue(0, sh.minlen);
  while (g < n_groups) {
    const int64_t g_next = grab();
    const Meta nxt = load_meta(g_next);  // in flight while this group is walked
    Shape sh_next = sh;
    const uint32_t minlen = sh.minlen;
    const int n_tiles = sh.n_tiles;

    const uint32_t pad = (cur.info >> VI_PAD_SHIFT) & 7u;
    const bool whole = PERM || (cur.info & VI_RELATIVE) == 0u;
    // an aggregate in one piece starts from its known state, a chunk from "whatever comes in" (relative)
    Acc a = whole ? ((p.init && cur.dest >= 0) ? load_state(p.init, cur.dest) : acc_none()) : acc_identity();
    Acc P = acc_identity();
    uint32_t undecM = whole ? 0u : ~0u;
    uint32_t frozenM = whole ? (uint32_t)__builtin_amdgcn_sbfe((int32_t)a.fl, 1, 1) : 0u;
    uint32_t corr = 0u;
    // wave-uniform: is any chunk here still waiting for its deciding event?  (usually settled within the first tile)
    const bool any_relative = !PERM && __builtin_amdgcn_ballot_w64(!whole) != 0ull;
    bool watching = any_relative;
    // A group of whole aggregates only (every group of SORTED; of CHUNKED all but the groups that hold chunks of cut aggregates)
    // walks CONCRETE states: presence and "threw" in two mask registers, none of the transformer's absolute / relative
    // bookkeeping — 7 VALU instructions fewer per event (apply_event_concrete; the tile-major fold has walked like this since
    // round 3).  Round 5's counters say why it matters here too: the pipelined kernel keeps the SIMDs' vector pipes busy 61 % of
    // its cycles (profiles/r05_c3_10Magg_sorted_summary.txt) — with two waves per SIMD the walk's instructions and the waits
    // for the next tile overlap only partly, so instructions saved are time saved.
    uint32_t presentM = (uint32_t)__builtin_amdgcn_sbfe((int32_t)a.fl, 0, 1);
    // one tile: wait for it, pull my LE events out of LDS, start the next tile's fetch, walk
    // (tracking: 0 = the transformer walk, 1 = the transformer walk that watches for deciding events, 2 = the concrete walk)
    auto tile_step = [&](int c, auto tracking) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      uint4 ev[LE];
#pragma unroll
      for (int j = 0; j < LE; ++j) ev[j] = *(const uint4*)(lds_ev + (ev_row ^ (uint32_t)(j * 16)));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (c + 1 < n_tiles) {
        issue(c + 1, minlen);
      } else {
        // last tile of this group: the NEXT group's first tile is fetched while this one is walked (its meta loads
        // were issued a whole group ago and, loads returning in order, landed before the tile just waited for)
        sh_next = shape_of(nxt);
        publish(nxt);
        if (g_next < n_groups) issue(0, sh_next.minlen);
      }

      uint32_t tyc[LE];
      if (c > 0 && (uint32_t)(c + 1) * LE <= minlen) {
#pragma unroll
        for (int j = 0; j < LE; ++j) tyc[j] = type_off(ev[j].x);
      } else {
        const int32_t rem = (int32_t)cur.len - c * LE;   // my remaining events (may be <= 0)
        const int32_t skip = c == 0 ? (int32_t)pad : 0;  // events in front of my aggregate (its first chunk only)
#pragma unroll
        for (int j = 0; j < LE; ++j)
          tyc[j] = (j >= skip && j < rem) ? type_off(ev[j].x) : kNullEntryOffBytes;
      }
      if constexpr (decltype(tracking)::value == 1) {
        walk_events_track<LE>(a, P, undecM, frozenM, corr, ev, tyc, lds_tab, p);
        watching = __builtin_amdgcn_ballot_w64(undecM != 0u && frozenM == 0u) != 0ull;
      } else if constexpr (decltype(tracking)::value == 2) {
        walk_events_concrete<LE>(a, presentM, frozenM, corr, ev, tyc, lds_tab, p);
      } else {
        walk_events<LE, false>(a, frozenM, corr, ev, tyc, 0u, lds_tab, p, [](int) {});
      }
    };
    // Two loops, not one loop with a branch: the watching loop (usually just the first tile of a group that holds
    // chunks of cut aggregates) carries P and the deciding-event test; the plain loop is the sorted-rows kernel's walk
    // and gets scheduled like it (one loop with both walks cost +13 % VALU instructions in the plain path).
    int c = 0;
    if (CONC && (PERM || !any_relative)) {
      for (; c < n_tiles; ++c) tile_step(c, std::integral_constant<int, 2>{});
      a.fl = (presentM & FL_PRESENT) | (frozenM & FL_POISONED);
    } else {  // (CONC = false: the transformer walk everywhere, as before round 5 — SURGE_REPLAY_WALK=transformer, for comparisons)
      if constexpr (!PERM)
        for (; c < n_tiles && watching; ++c) tile_step(c, std::integral_constant<int, 1>{});
      for (; c < n_tiles; ++c) tile_step(c, std::integral_constant<int, 0>{});
    }
    a.sum = (int64_t)((uint64_t)a.sum + corr);

    if (cur.dest >= 0) {
      if (!PERM && (cur.info & VI_SIDE)) {
        // a chunk without a deciding event (or an empty one, which walked clamped garbage) is all prefix
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

    g = g_next;
    cur = nxt;
    sh = sh_next;
  }
  dispenser_leave(p.counter, lane);
}

// Register budget = resident waves: 2 per SIMD (<= 256 VGPRs) with 16 KiB tiles, 3 per SIMD (<= 168) with 8 KiB tiles.
template <int LE, bool CONC>
__global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(LE == 8 ? 3 : 2)))
fold_chunked_kernel(const FoldParams p, const ChunkTable t) {
  chunk_walk<LE, false, CONC>(p, t);
}

// K2 "sorted rows", pipelined across groups (round 5; SURGE_ALGO_SORTED's kernel — fold_sorted_kernel stays selectable)
template <int LE, bool CONC>
__global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(LE == 8 ? 3 : 2)))
fold_sorted_pf_kernel(const FoldParams p) {
  ChunkTable t;
  t.v_start = nullptr; t.v_len = nullptr; t.v_info = nullptr; t.v_dest = nullptr; t.n_vrows = 0; t.side = nullptr;
  chunk_walk<LE, true, CONC>(p, t);
}

bool concrete_walk() {
  static const bool v = [] { const char* e = std::getenv("SURGE_REPLAY_WALK"); return !(e && std::strcmp(e, "transformer") == 0); }();
  return v;
}

}  // namespace

hipError_t launch_fold_sorted_pf(const FoldParams& p, int64_t n_waves, int lane_events, hipStream_t stream) {
  if (n_waves <= 0) return hipSuccess;
  const bool conc = concrete_walk();
  if (lane_events == 8) {
    if (conc) hipLaunchKernelGGL((fold_sorted_pf_kernel<8, true>), dim3((unsigned)n_waves), dim3(kWave), Geo<8>::lds_bytes(Geo<8>::kAuxSorted), stream, p);
    else hipLaunchKernelGGL((fold_sorted_pf_kernel<8, false>), dim3((unsigned)n_waves), dim3(kWave), Geo<8>::lds_bytes(Geo<8>::kAuxSorted), stream, p);
  } else {
    if (conc) hipLaunchKernelGGL((fold_sorted_pf_kernel<16, true>), dim3((unsigned)n_waves), dim3(kWave), Geo<16>::lds_bytes(Geo<16>::kAuxSorted), stream, p);
    else hipLaunchKernelGGL((fold_sorted_pf_kernel<16, false>), dim3((unsigned)n_waves), dim3(kWave), Geo<16>::lds_bytes(Geo<16>::kAuxSorted), stream, p);
  }
  return hipGetLastError();
}

// the fold over the chunk table + the stitch of the cut aggregates
hipError_t launch_fold_chunked(const FoldParams& p, const int64_t* v_start, const uint32_t* v_len, const uint32_t* v_info,
                               const int64_t* v_dest, int64_t n_vrows, uint32_t* side, const int64_t* r_slot0, const uint32_t* r_c,
                               const int64_t* r_out, int64_t n_cut, int64_t n_waves, int lane_events, hipStream_t stream) {
  if (n_waves <= 0 || n_vrows <= 0) return hipSuccess;
  ChunkTable t;
  t.v_start = v_start; t.v_len = v_len; t.v_info = v_info; t.v_dest = v_dest; t.n_vrows = n_vrows; t.side = side;
  const bool conc = concrete_walk();
  if (lane_events == 8) {
    if (conc) hipLaunchKernelGGL((fold_chunked_kernel<8, true>), dim3((unsigned)n_waves), dim3(kWave), Geo<8>::lds_bytes(Geo<8>::kAuxSorted), stream, p, t);
    else hipLaunchKernelGGL((fold_chunked_kernel<8, false>), dim3((unsigned)n_waves), dim3(kWave), Geo<8>::lds_bytes(Geo<8>::kAuxSorted), stream, p, t);
  } else {  // 32-event lanes spill (337 VGPRs): 16 is the widest tile of this kernel
    if (conc) hipLaunchKernelGGL((fold_chunked_kernel<16, true>), dim3((unsigned)n_waves), dim3(kWave), Geo<16>::lds_bytes(Geo<16>::kAuxSorted), stream, p, t);
    else hipLaunchKernelGGL((fold_chunked_kernel<16, false>), dim3((unsigned)n_waves), dim3(kWave), Geo<16>::lds_bytes(Geo<16>::kAuxSorted), stream, p, t);
  }
  if (n_cut > 0)
    hipLaunchKernelGGL(chunk_stitch_kernel, dim3((unsigned)((n_cut + 127) / 128)), dim3(128), 0, stream, p, side, r_slot0, r_c, r_out, n_cut);
  return hipGetLastError();
}

// the stitch alone (the tile-major fold, fold_tiled.hip, leaves the same side entries)
hipError_t launch_chunk_stitch(const FoldParams& p, const uint32_t* side, const int64_t* r_slot0, const uint32_t* r_c, const int64_t* r_out,
                               int64_t n_cut, hipStream_t stream) {
  if (n_cut <= 0) return hipSuccess;
  hipLaunchKernelGGL(chunk_stitch_kernel, dim3((unsigned)((n_cut + 127) / 128)), dim3(128), 0, stream, p, side, r_slot0, r_c, r_out, n_cut);
  return hipGetLastError();
}

}  // namespace surge
