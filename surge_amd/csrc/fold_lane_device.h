// fold_lane_device.h — the device code of the lane-per-row folds: the walk of SORTED / CHUNKED (whole aggregates in length
// order, or the chunk table's virtual rows) and the uniform-rows walk of ROWS.  Compiled twice: ahead of time
// (fold_chunked.hip, fold_kernels.hip — the op table is data in LDS) and at run time for one v1 schema (hiprtc,
// fold_kernels.hip writes the program: SURGE_V1_SPEC — the op table is 16 compile-time bit masks, the table reads are gone,
// the arithmetic of fields no event type touches folds away).
#pragma once
#include "fold_chunk_device.h"

namespace surge {
namespace {

template <int V>
struct IntC { static constexpr int value = V; };

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
#ifdef SURGE_EXP_STATIC_TABLE  // (experiment builds: the op table as an LDS object of its own — the compiler then drops the vmcnt(0) it puts in front of the walk's table reads)
  __shared__ __attribute__((aligned(16))) uint32_t lds_tab_static[kTableLdsDwords];
  uint32_t* lds_tab = lds_tab_static;
#else
  uint32_t* lds_tab = (uint32_t*)(smem + G::kTileBytes + G::kAuxSorted);
#endif
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
      // Every row start out of the LDS table FIRST, then the loads back to back.  Interleaved (read a start, issue its load, read
      // the next start ...) is what this was until round 6 — and what hiprtc's compiler, unlike hipcc's, answers with an
      // s_waitcnt vmcnt(0) in front of every one of those LDS reads (an LDS-DMA load in flight "may alias" them): each load of
      // the tile then waits for the one before it, a 16 KiB tile takes 16 memory latencies (4.4 us on an idle chip against 1.4:
      // profiles/r06_lane_spec_ab_c3_per_wave.jsonl) and the kernels compiled at run time lost 25 % to the ones compiled ahead.
      constexpr int kBurst = G::kLoads <= 16 ? G::kLoads : 16;
#pragma unroll
      for (int q0 = 0; q0 < G::kLoads; q0 += kBurst) {
        uint64_t addr[kBurst];
#pragma unroll
        for (int q = 0; q < kBurst; ++q) addr[q] = ebase[(q0 + q) % G::kClasses] + coff + ((uint64_t)lds_rs[G::kRowsPerLoad * (q0 + q) + lane / LE] << 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < kBurst; ++q) __builtin_amdgcn_global_load_lds((gptr_t)addr[q], (lptr_t)(lds_ev + (q0 + q) * 1024), 16, 0, kLoadAux);
        __builtin_amdgcn_sched_barrier(0);
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
#ifdef SURGE_EXP_WAIT_BEFORE_WALK  // (experiment builds: what the ahead-of-time kernels do unasked — see DESIGN §3)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
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
      begin_concrete(a);
      for (; c < n_tiles; ++c) tile_step(c, IntC<2>{});
      finish_concrete(a, presentM, frozenM, corr, p);
    } else {  // (CONC = false: the transformer walk everywhere, as before round 5 — SURGE_REPLAY_WALK=transformer, for comparisons)
      if constexpr (!PERM)
        for (; c < n_tiles && watching; ++c) tile_step(c, IntC<1>{});
      for (; c < n_tiles; ++c) tile_step(c, IntC<0>{});
      a.sum = (int64_t)((uint64_t)a.sum + corr);
    }

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

// ---- K1 "rows": uniform fan-in, one lane per aggregate (the kernel's comment is in fold_kernels.hip) ----------------------
// lds_ev (G::kTileBytes) and lds_tab (kTableLdsDwords dwords; unused in a build for one schema) are two separate LDS objects of
// the kernel, not one carved-up buffer: the compiler orders every LDS read after all outstanding global->LDS loads that MAY
// alias it.  With a single dynamic buffer the op-table reads of the walk "may alias" the tile being fetched, and hipcc put
// an s_waitcnt vmcnt(0) in front of the first table read — i.e. each wave waited for its NEXT tile before walking the current
// one.  Distinct objects let alias analysis drop that wait.
// CONC: the concrete-state walk (a lane owns its row outright); false: the transformer walk of rounds 1-5.
template <int LE, bool CONC>
__device__ __forceinline__ void fold_rows_body(const FoldParams& p, char* lds_ev, uint32_t* lds_tab) {
  using G = Geo<LE>;

  const int lane = threadIdx.x;
  const int64_t S0 = (int64_t)blockIdx.x * p.segs_per_task;
  int64_t S1 = S0 + p.segs_per_task;
  S1 = S1 < p.n_seg ? S1 : p.n_seg;
  if (S0 >= S1) return;
  const uint32_t L = (uint32_t)p.fixed_len;
  const int chunks = (int)(L / LE);
  const int n_groups = (int)((S1 - S0 + kWave - 1) / kWave);
  const int n_tiles = n_groups * chunks;

  load_table<LE>(p, lds_tab, lane);

  // lane offsets of the load-instruction classes: row (m / LE) of the instruction's rows, swizzled slot
  uint32_t voff[G::kClasses];
#pragma unroll
  for (int k = 0; k < G::kClasses; ++k) voff[k] = ((uint32_t)(lane / LE) * L + G::load_j(lane, k)) * 16u;
  const uint32_t ev_row = G::ev_row(lane);

  // buffer_load ... lds (see issue_tile_loads for why not global_load_lds): descriptor base = the group's first row at
  // this tile's column, lane offset = its row (m / LE) and swizzled slot, instruction q at scalar offset RPL q rows
  auto issue = [&](int t) {
    const int g = t / chunks, c = t - g * chunks;
    const int64_t row0 = S0 + (int64_t)g * kWave;
    const char* base = (const char*)(p.events + (row0 * L + (int64_t)c * LE));  // wave-uniform
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    if (row0 + kWave <= p.n_seg) {
#pragma unroll
      for (int q = 0; q < G::kLoads; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds_ev + q * 1024), 16, (int)voff[q % G::kClasses],
                                                 (int)((uint32_t)(G::kRowsPerLoad * q) * L * 16u), 0, kLoadAux);
    } else {  // last group of the log: rows past the end re-read the last row (their lanes are idle)
#pragma unroll
      for (int q = 0; q < G::kLoads; ++q) {
        int64_t row = row0 + G::kRowsPerLoad * q + lane / LE;
        row = row < p.n_seg ? row : p.n_seg - 1;
        const int off = (int)(((row - row0) * L + G::load_j(lane, q % G::kClasses)) * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds_ev + q * 1024), 16, off, 0, 0, kLoadAux);
      }
    }
  };

  issue(0);
  Acc a = acc_none();
  uint32_t frozenM = 0u, corr = 0u, presentM = 0u;
  int c = 0;
  int64_t row = S0 + lane;
  for (int t = 0; t < n_tiles; ++t) {
    if (c == 0) {
      a = (p.init && row < S1) ? load_state(p.init, row) : acc_none();
      frozenM = (uint32_t)__builtin_amdgcn_sbfe((int32_t)a.fl, 1, 1);
      presentM = (uint32_t)__builtin_amdgcn_sbfe((int32_t)a.fl, 0, 1);
      corr = 0u;
      if (CONC) begin_concrete(a);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint4 ev[LE];
#pragma unroll
    for (int j = 0; j < LE; ++j) ev[j] = *(const uint4*)(lds_ev + (ev_row ^ (uint32_t)(j * 16)));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t + 1 < n_tiles) issue(t + 1);

    uint32_t tyc[LE];
#pragma unroll
    for (int j = 0; j < LE; ++j) tyc[j] = type_off(ev[j].x);
    if (CONC) walk_events_concrete<LE>(a, presentM, frozenM, corr, ev, tyc, lds_tab, p);
    else walk_events<LE, false>(a, frozenM, corr, ev, tyc, 0u, lds_tab, p, [](int) {});
    if (++c == chunks) {
      c = 0;
      if (CONC) finish_concrete(a, presentM, frozenM, corr, p);
      else a.sum = (int64_t)((uint64_t)a.sum + corr);
      if (row < S1) store_state(p.out, row, a);
      row += kWave;
    }
  }
}

}  // namespace
}  // namespace surge
