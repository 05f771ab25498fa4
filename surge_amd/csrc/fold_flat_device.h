// fold_flat_device.h — the body of the flat fold kernel (fold_kernels.hip's header comment describes it): one wave task =
// a contiguous range of whole segments, lane transformers + a wave-level segmented scan.  In a header because it is
// compiled twice: ahead of time by hipcc (fold_kernels.hip: the op table is data, read from LDS per event) and, per v1
// schema, at run time by hiprtc (SURGE_V1_SPEC: the op table's words are compile-time bit masks over the event type — see
// fold_device.h — so the table reads become one v_bfe each and the arithmetic of fields no event type touches folds away).
#pragma once
#include "fold_device.h"

namespace surge {
namespace {

enum { MODE_FIXED = 0, MODE_FLAT = 1 };

template <int MODE, int LE>
__device__ __forceinline__ void fold_flat_body(const FoldParams& p, char* smem) {
  using G = Geo<LE>;
  // one dynamic LDS buffer: splitting it into separate objects (which lets hipcc drop its conservative
  // s_waitcnt vmcnt(0) between the tile fetch and the op-table reads, see fold_rows_kernel) measured 3 % SLOWER
  // here — more VGPRs, and the head bookkeeping loads wait on vmcnt anyway
  char* lds_ev = smem;
  uint32_t* lds_hb = (uint32_t*)(smem + G::kTileBytes);
  uint32_t* lds_tab = (uint32_t*)(smem + G::kTileBytes + G::kAuxFlat);

  const int lane = threadIdx.x;
  const int64_t task = blockIdx.x;

  int64_t S0, S1, E0, E1;
  if (MODE == MODE_FIXED) {
    S0 = task * p.segs_per_task;
    S1 = S0 + p.segs_per_task;
    S1 = S1 < p.n_seg ? S1 : p.n_seg;
    E0 = S0 * p.fixed_len;
    E1 = S1 * p.fixed_len;
  } else {
    S0 = uniform64(p.plan[task]);
    S1 = uniform64(p.plan[task + 1]);
    if (S0 >= S1) return;
    E0 = uniform64(p.seg_off[S0]);
    E1 = uniform64(p.seg_off[S1]);
  }
  if (E1 <= E0) return;
  // Tiles are cut from the 128-byte line that holds the task's first event: every 1 KiB load instruction then covers 8
  // whole lines instead of straddling 9 (measured in round 2 on a 0.1 M-aggregate log: FETCH_SIZE 1.17 x the algorithmic
  // bytes).  The `lead` events in front of E0 belong to the previous task's last segment: null events here.
  const int lead = (int)(E0 & 7);
  const int64_t Ea = E0 - lead;

  load_table<LE>(p, lds_tab, lane);
  if (MODE == MODE_FLAT && lane < G::kHeadWords) lds_hb[lane] = 0u;

  const int n_tiles = (int)((E1 - Ea + G::kTile - 1) / G::kTile);
  uint32_t voff[G::kClasses];
#pragma unroll
  for (int k = 0; k < G::kClasses; ++k) voff[k] = ((uint32_t)(lane / LE) * LE + G::load_j(lane, k)) * 16u;
  const uint32_t ev_row = G::ev_row(lane);
  issue_tile_loads<LE>(p, Ea, E1, lds_ev, voff);

  // FLAT: head marking.  next_s = first segment whose start has not been marked yet.
  int64_t next_s = S0;
  auto mark_heads = [&](int64_t te0) {
    const int64_t te1 = (te0 + G::kTile < E1) ? te0 + G::kTile : E1;
    while (true) {
      const int64_t s = next_s + lane;
      const int64_t v = (s < S1) ? p.seg_off[s] : E1;
      const bool in = v < te1;
      if (in) {
        const uint32_t pos = (uint32_t)(v - te0);
        atomicOr(&lds_hb[pos >> 5], 1u << (pos & 31));
      }
      const int cnt = __popcll(__ballot(in));
      next_s += cnt;
      if (cnt < kWave) break;
    }
  };
  if (MODE == MODE_FLAT) mark_heads(Ea);

  // The running segment that enters the next tile; starts as "nothing" (a head, None).
  Acc carry = acc_none();
  carry.fl |= FL_HEAD;
  int64_t c = S0 - 1;  // FLAT: index of the segment open when the tile starts

  const uint64_t below = (1ull << lane) - 1ull;

  for (int tile = 0; tile < n_tiles; ++tile) {
    const int64_t te0 = Ea + (int64_t)tile * G::kTile;

    // tile `tile` has landed in LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint4 ev[LE];
#pragma unroll
    for (int j = 0; j < LE; ++j) ev[j] = *(const uint4*)(lds_ev + (ev_row ^ (uint32_t)(j * 16)));

    uint32_t hb;          // bit j: my event j starts a new segment
    int64_t seg_open;     // segment open when my chunk starts (before a head at j = 0)
    int heads_in_tile = 0;
    if (MODE == MODE_FLAT) {
      hb = (lds_hb[(lane * LE) >> 5] >> ((lane * LE) & 31)) & G::kLaneMask;
      int incl = __popc(hb);
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
      }
      seg_open = c + (incl - __popc(hb));
      heads_in_tile = (int)rl((uint32_t)incl, 63);
    } else {
      const uint32_t L = (uint32_t)p.fixed_len;
      const uint32_t e_rel = (uint32_t)(te0 - E0) + (uint32_t)lane * LE;  // FIXED: L % 16 == 0, so lead == 0
      const uint32_t q = e_rel / L;
      const uint32_t r = e_rel - q * L;
      hb = (r == 0u && te0 + lane * LE < E1) ? 1u : 0u;
      seg_open = S0 - 1 + (int64_t)q + (r != 0u ? 1 : 0);
    }
    // all my reads of the event buffer are done: it can take the next tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == MODE_FLAT && lane < G::kHeadWords) lds_hb[lane] = 0u;
    if (tile + 1 < n_tiles) issue_tile_loads<LE>(p, te0 + G::kTile, E1, lds_ev, voff);

    // LDS dword offset of each event's op-table entry.  Events past the end of the task (last tile
    // only) become the null event [17], an identity on every state, so nothing below needs a validity mask.
    uint32_t tyc[LE];
    if (te0 + G::kTile <= E1 && (tile > 0 || lead == 0)) {
#pragma unroll
      for (int j = 0; j < LE; ++j) tyc[j] = type_off(ev[j].x);
    } else {
      const int64_t rem = E1 - (te0 + (int64_t)lane * LE);
      const int skip = tile == 0 ? lead - lane * LE : 0;  // the lead events of the first tile sit in lane 0 (lead < 8 <= LE)
#pragma unroll
      for (int j = 0; j < LE; ++j)
        tyc[j] = (j >= skip && (int64_t)j < rem) ? type_off(ev[j].x) : kNullEntryOffBytes;
    }

    // ---- pass A: presence / poison only, bit-parallel ---------------------------------------------
    // Three LE-bit masks over my events: P throws, D deletes, M materialises.  The piece that matters to
    // later lanes is the one after my LAST head (or my whole chunk): events before its first throwing
    // event are live; the last live M|D event, if any, forces presence to a constant.
    bool has_head, c_const, c_val, poi;
    {
      uint32_t PD = 0u, Mb = 0u;
#pragma unroll
      for (int j = 0; j < LE; ++j) {
        PD |= flags_word(lds_tab, tyc[j]) << j;                           // bit j: throws ; bit 16+j: deletes
        Mb |= materializes_word(lds_tab, tyc[j]) & (1u << j);
      }
      has_head = hb != 0u;
      const uint32_t lo = has_head ? (31u - (uint32_t)__clz((int)hb)) : 0u;
      uint32_t ifl = 0u;
      if (has_head && p.init) {
        const int64_t sg = seg_open + __popc(hb);
        const int64_t ii = p.out_map ? p.out_map[sg] : sg;
        ifl = ((const uint32_t*)(p.init + ii * 4 + 2))[1];
      }
      const bool ib = (ifl & FL_PRESENT) != 0u, iq = (ifl & FL_POISONED) != 0u;
      const uint32_t range = G::kLaneMask & ~((1u << lo) - 1u);
      const uint32_t Pm = PD & range;
      uint32_t live = Pm ? (range & ((1u << __builtin_ctz(Pm)) - 1u)) : range;
      live = iq ? 0u : live;
      const uint32_t dec = (Mb | (PD >> 16)) & live;
      const uint32_t top = dec ? (31u - (uint32_t)__clz((int)dec)) : 0u;
      c_const = has_head || dec != 0u;
      c_val = dec ? (((Mb >> top) & 1u) != 0u) : ib;
      poi = Pm != 0u || iq;
    }
    // incoming (present, poisoned) of every lane from four ballots
    bool b_in, q_in;
    {
      const uint64_t Cm = __ballot(c_const);  // my chunk forces presence to a constant ...
      const uint64_t Vm = __ballot(c_val);    // ... this one
      const uint64_t Hm = __ballot(has_head);
      const uint64_t Qm = __ballot(poi);
      const uint64_t x = Cm & below;
      b_in = x ? (((Vm >> (63 - __clzll((long long)x))) & 1ull) != 0) : ((carry.fl & FL_PRESENT) != 0);
      const uint64_t h = Hm & below;
      bool base = (carry.fl & FL_POISONED) != 0;
      uint64_t range = below;
      if (h) {
        const int hp = 63 - __clzll((long long)h);
        base = ((Qm >> hp) & 1ull) != 0;
        range = below & ~((2ull << hp) - 1ull);
      }
      q_in = base || ((Qm & ~Hm & range) != 0ull);
    }

    // ---- pass B: one evaluation path per lane ----------------------------------------------------
    Acc a = (b_in || q_in) ? acc_identity() : acc_none();
    uint32_t frozenM = q_in ? ~0u : 0u;
    uint32_t corr = 0u;  // pending "+1"s of the sum64 complement trick
    Acc lead = a;
    bool seen = false;
    const int64_t lead_seg = seg_open;
    auto on_head = [&](int) {
      a.sum = (int64_t)((uint64_t)a.sum + corr);
      corr = 0u;
      if (!seen) {
        lead = a;
        seen = true;
      } else {
        const int64_t oi = p.out_map ? p.out_map[seg_open] : seg_open;
        store_state_flat(p, oi, a);
      }
      seg_open += 1;
      if (p.init) {
        const int64_t ii = p.out_map ? p.out_map[seg_open] : seg_open;
        a = load_state(p.init, ii);
      } else {
        a = acc_none();
      }
      frozenM = (uint32_t)__builtin_amdgcn_sbfe((int32_t)a.fl, 1, 1);
    };
    if (MODE == MODE_FIXED) {  // a head can only sit at j = 0 (L % LE == 0)
      if (hb & 1u) on_head(0);
      walk_events<LE, false>(a, frozenM, corr, ev, tyc, 0u, lds_tab, p, on_head);
    } else {
      walk_events<LE, true>(a, frozenM, corr, ev, tyc, hb, lds_tab, p, on_head);
    }
    a.sum = (int64_t)((uint64_t)a.sum + corr);

    // ---- wave-level segmented scan of the lane transformers --------------------------------------
    Acc el = a;
    if (seen) el.fl |= FL_HEAD;
    {
      const Acc seeded = seq_acc(carry, el);
      if (lane == 0 && !seen) el = seeded;
    }
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const Acc o = shfl_up_acc(el, d);
      const Acc cmb = seq_acc(o, el);
      const bool keep = (el.fl & FL_HEAD) || lane < d;
      el = select_acc(keep, el, cmb);
    }
    Acc prefix = shfl_up_acc(el, 1);
    if (lane == 0) prefix = carry;

    if (seen && lead_seg >= S0) {
      const Acc fin = seq_acc(prefix, lead);
      const int64_t oi = p.out_map ? p.out_map[lead_seg] : lead_seg;
      store_state_flat(p, oi, fin);
    }
    carry = readlane_acc(el, 63);

    if (MODE == MODE_FLAT) {
      c += heads_in_tile;
      if (tile + 1 < n_tiles) mark_heads(te0 + G::kTile);
    }
  }

  if (lane == 0) {
    const int64_t oi = p.out_map ? p.out_map[S1 - 1] : (S1 - 1);
    store_state_flat(p, oi, carry);
  }
}

}  // namespace
}  // namespace surge
