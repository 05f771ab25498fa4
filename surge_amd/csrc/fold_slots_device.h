// fold_slots_device.h — device code of the ABI v2 "slot" fold (include/surge_replay.h): up to 7 typed 8-byte slots, one
// operation per slot per event type, ANY mix of operations (ADD / SUB / SET / MIN / MAX, integer or IEEE double).
//
// The v1 kernels split aggregates across lanes and waves and therefore need every field to compose associatively; a
// generic slot schema promises nothing of the sort (an f64 ADD is order-sensitive, ADD followed by MIN followed by ADD
// on one slot has no closed form).  So these kernels never split an aggregate: ONE lane walks ONE aggregate's events
// (or one micro-batch group's) strictly in order with a concrete running state.  That also makes f64 accumulation
// bit-identical to the JVM's sequential fold (no tolerance).
//
// This header is compiled twice:
//   * ahead of time (fold_slots.hip) as the GENERIC INTERPRETER: slot type, operand source and the SET of operations any
//     event type ever applies to a slot come from the kernel argument (wave-uniform: scalar branches); only the
//     operation an individual event applies is per lane.  Rolled event loop — unrolled, the 7 slots x 3 value types x
//     5 operations do not fit the instruction cache (round 2: 190 KB of code, 21–27 % of peak);
//   * at run time (hiprtc, fold_slots.hip: rtc_slots_*) with SURGE_SLOTS_SPEC defined and the handle's schema in front
//     of it as SURGE_SPEC_* macros: every branch on the schema folds away, the walk is unrolled like the v1 kernels'
//     (events in registers, the next tile in flight during the walk) and costs what the schema's own arithmetic costs.
// Two transports, each in both builds: the bound CSR log through line-aligned row pieces (sorted-rows, also every
// micro-batch) and the tile-major copy of a bound log (fold_tiled.hip's layout: one linear 8 KiB run per subtile).
#pragma once
#include "fold_device.h"

namespace surge {

struct SlotParams {
  uint32_t n_slots;
  uint32_t count_events;
  uint32_t type[SURGE_MAX_SLOTS];
  uint32_t source[SURGE_MAX_SLOTS];
  uint32_t used_ops[SURGE_MAX_SLOTS];  // bit o set: some event type applies SURGE_OP_o to this slot
  uint64_t def[SURGE_MAX_SLOTS];
  uint32_t cls[SURGE_MAX_EVENT_TYPES + 2];  // [16] unknown type: throws; [17] null (padding) event
  uint32_t ops[SURGE_MAX_EVENT_TYPES + 2];
};

namespace {

#if defined(SURGE_SLOTS_SPEC)
constexpr bool kSlotsSpec = true;
// the schema, compiled in (SURGE_SPEC_* are ternary chains over a constant index: they fold after unrolling)
struct SlotSchema {
  __device__ __forceinline__ constexpr uint32_t n_slots() const { return SURGE_SPEC_N_SLOTS; }
  __device__ __forceinline__ constexpr uint32_t count_events() const { return SURGE_SPEC_COUNT_EVENTS; }
  __device__ __forceinline__ constexpr uint32_t type(int i) const { return SURGE_SPEC_TYPE(i); }
  __device__ __forceinline__ constexpr uint32_t source(int i) const { return SURGE_SPEC_SOURCE(i); }
  __device__ __forceinline__ constexpr uint32_t used_ops(int i) const { return SURGE_SPEC_USED(i); }
  __device__ __forceinline__ constexpr uint64_t def(int i) const { return SURGE_SPEC_DEF(i); }
  __device__ __forceinline__ uint32_t cls(int t) const { return SURGE_SPEC_CLS(t); }
  __device__ __forceinline__ uint32_t ops(int t) const { return SURGE_SPEC_OPS(t); }
};
#else
constexpr bool kSlotsSpec = false;
struct SlotSchema {  // the schema as the kernel argument (scalar registers)
  const SlotParams& p;
  __device__ __forceinline__ uint32_t n_slots() const { return p.n_slots; }
  __device__ __forceinline__ uint32_t count_events() const { return p.count_events; }
  __device__ __forceinline__ uint32_t type(int i) const { return p.type[i]; }
  __device__ __forceinline__ uint32_t source(int i) const { return p.source[i]; }
  __device__ __forceinline__ uint32_t used_ops(int i) const { return p.used_ops[i]; }
  __device__ __forceinline__ uint64_t def(int i) const { return p.def[i]; }
  __device__ __forceinline__ uint32_t cls(int t) const { return p.cls[t]; }
  __device__ __forceinline__ uint32_t ops(int t) const { return p.ops[t]; }
};
#endif

constexpr uint32_t CLS_NULL = 1u << 31;  // SlotParams.cls[17], the padding event: identity on every state

// The per-type table in LDS: one entry per event type at the v1 op table's stride (80 bytes — so an event of the
// tile-major log, whose type word already IS its entry's byte offset, needs no decode here either), six words used:
enum { SW_POISON = 0, SW_DELETE = 1, SW_NOT_REQUIRE = 2, SW_CREATE = 3, SW_OPS = 4, SW_EVC = 5 };
// masks are all-ones / all-zero; SW_OPS = the 4-bit operation codes of the 7 slots; SW_EVC = 1 when an applied event of
// this type counts towards event_count.  The null entry is all zero: it "applies" to a present aggregate as seven KEEPs.

__device__ __forceinline__ void slots_load_table(const SlotSchema& sc, uint32_t* lds_tab, int lane) {
  if (lane < SURGE_MAX_EVENT_TYPES + 2) {
    const uint32_t cls = sc.cls(lane), ops = sc.ops(lane);
    const bool null = (cls & CLS_NULL) != 0u, poison = !null && (cls & SURGE_D_POISON) != 0u;
    const bool real = !null && !poison;
    const uint32_t c = cls & SURGE_CLS_MASK;
    uint32_t* e = lds_tab + (lane == SURGE_MAX_EVENT_TYPES + 1 ? kNullEntryOff : lane * kTableStride);
    e[SW_POISON] = poison ? ~0u : 0u;
    e[SW_DELETE] = (real && c == SURGE_CLS_DELETE) ? ~0u : 0u;
    e[SW_NOT_REQUIRE] = (real && c != SURGE_CLS_REQUIRE) ? ~0u : 0u;
    e[SW_CREATE] = (real && c == SURGE_CLS_CREATE) ? ~0u : 0u;
    e[SW_OPS] = (real && c != SURGE_CLS_DELETE) ? ops : 0u;
    e[SW_EVC] = (real && c != SURGE_CLS_DELETE && sc.count_events()) ? 1u : 0u;
  }
}

struct SlotState {
  uint32_t lo[SURGE_MAX_SLOTS], hi[SURGE_MAX_SLOTS];
  uint32_t evc;
  uint32_t presentM, frozenM;  // all-ones / all-zero: the aggregate is Some; an event threw (or the prior state was POISONED)
};

__device__ __forceinline__ SlotState slots_none() {
  SlotState st;
#pragma unroll
  for (int i = 0; i < SURGE_MAX_SLOTS; ++i) st.lo[i] = st.hi[i] = 0u;
  st.evc = 0u;
  st.presentM = st.frozenM = 0u;
  return st;
}

__device__ __forceinline__ SlotState slots_load(const uint4* in, int64_t idx) {
  const uint4* s = in + idx * 4;
  const uint4 v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3];
  SlotState st;
  st.lo[0] = v0.x; st.hi[0] = v0.y; st.lo[1] = v0.z; st.hi[1] = v0.w;
  st.lo[2] = v1.x; st.hi[2] = v1.y; st.lo[3] = v1.z; st.hi[3] = v1.w;
  st.evc = v2.x;
  st.presentM = (uint32_t)__builtin_amdgcn_sbfe((int32_t)v2.y, 0, 1);
  st.frozenM = (uint32_t)__builtin_amdgcn_sbfe((int32_t)v2.y, 1, 1);
  st.lo[4] = v2.z; st.hi[4] = v2.w;
  st.lo[5] = v3.x; st.hi[5] = v3.y; st.lo[6] = v3.z; st.hi[6] = v3.w;
  return st;
}

__device__ __forceinline__ void slots_store(uint4* out, int64_t idx, const SlotState& st) {
  const uint32_t pm = st.presentM;  // None is canonically all-zero (plus, possibly, the POISONED flag)
  uint4* o = out + idx * 4;
  o[0] = make_uint4(st.lo[0] & pm, st.hi[0] & pm, st.lo[1] & pm, st.hi[1] & pm);
  o[1] = make_uint4(st.lo[2] & pm, st.hi[2] & pm, st.lo[3] & pm, st.hi[3] & pm);
  o[2] = make_uint4(st.evc & pm, (pm & FL_PRESENT) | (st.frozenM & FL_POISONED), st.lo[4] & pm, st.hi[4] & pm);
  o[3] = make_uint4(st.lo[5] & pm, st.hi[5] & pm, st.lo[6] & pm, st.hi[6] & pm);
}

__device__ __forceinline__ double bits_f64(uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo)); }

// java.lang.Math.min / max on doubles, as the JDK's library source states them (a JVM model's math.min / math.max):
// a NaN on either side gives that NaN, -0.0 is smaller than +0.0.  Returns true when the result is b.
__device__ __forceinline__ bool java_min_takes_b(double a, double b, uint32_t b_lo, uint32_t b_hi) {
  const bool a_nan = a != a;
  const bool zeros = a == 0.0 && b == 0.0 && b_hi == 0x80000000u && b_lo == 0u;  // b is -0.0
  return !a_nan && (zeros || !(a <= b));
}
__device__ __forceinline__ bool java_max_takes_b(double a, double b, uint32_t a_lo, uint32_t a_hi) {
  const bool a_nan = a != a;
  const bool zeros = a == 0.0 && b == 0.0 && a_hi == 0x80000000u && a_lo == 0u;  // a is -0.0
  return !a_nan && (zeros || !(a >= b));
}

// one handleEvent step on a concrete state.  q: the event type's mask words (SW_POISON .. SW_CREATE), opsw / evcw: its
// SW_OPS / SW_EVC words (all per lane).
__device__ __forceinline__ void slots_apply(SlotState& st, const uint4 q, uint32_t opsw, uint32_t evcw, uint32_t seq, uint32_t raw_lo,
                                            uint32_t raw_hi, const SlotSchema& sc) {
  const uint32_t goM = ~(st.frozenM | q.x);
  st.frozenM |= q.x;                                               // a throwing event freezes the aggregate
  const uint32_t delM = goM & q.y;
  const uint32_t appM = andn(goM, q.y) & (st.presentM | q.z);      // REQUIRE-class events skip None
  const uint32_t rstM = appM & (q.w | ~st.presentM);               // CREATE, or materialising from None
  st.presentM = andn(st.presentM, delM) | rstM;
  st.evc = andn(st.evc, rstM) + (evcw & appM);
  const uint32_t ops = opsw & appM;                                // an event that does not apply KEEPs every slot
#pragma unroll
  for (int i = 0; i < SURGE_MAX_SLOTS; ++i) {
    if (i >= (int)sc.n_slots()) break;                             // wave-uniform (spec build: compile time)
    const uint32_t ty = sc.type(i), src = sc.source(i), used = sc.used_ops(i);
    const uint64_t def = sc.def(i);
    const uint32_t op = (ops >> (4 * i)) & 15u;                    // per lane
    const uint32_t cur_lo = bfi(rstM, (uint32_t)def, st.lo[i]);
    const uint32_t cur_hi = ty == SURGE_SLOT_I32 ? 0u : bfi(rstM, (uint32_t)(def >> 32), st.hi[i]);
    // the operand, in the slot's own type
    uint32_t x_lo, x_hi;
    if (ty == SURGE_SLOT_F64) {
      if (src == SURGE_SRC_PAYLOAD) {
        x_lo = raw_lo; x_hi = raw_hi;
      } else {
        const double d = src == SURGE_SRC_ONE ? 1.0 : (double)(int32_t)(src == SURGE_SRC_SEQ ? seq : raw_lo);
        const uint64_t b = (uint64_t)__double_as_longlong(d);
        x_lo = (uint32_t)b; x_hi = (uint32_t)(b >> 32);
      }
    } else {
      if (src == SURGE_SRC_PAYLOAD) {
        x_lo = raw_lo; x_hi = raw_hi;
      } else if (src == SURGE_SRC_ONE) {
        x_lo = 1u; x_hi = 0u;
      } else {
        x_lo = src == SURGE_SRC_SEQ ? seq : raw_lo;
        x_hi = (uint32_t)((int32_t)x_lo >> 31);
      }
      if (ty == SURGE_SLOT_I32) x_hi = 0u;
    }
    uint32_t r_lo = cur_lo, r_hi = cur_hi;
    if (ty == SURGE_SLOT_F64) {
      const double a = bits_f64(cur_lo, cur_hi), b = bits_f64(x_lo, x_hi);
      if (used & (1u << SURGE_OP_ADD)) {
        const uint64_t s = (uint64_t)__double_as_longlong(a + b);
        const bool m = op == SURGE_OP_ADD;
        r_lo = m ? (uint32_t)s : r_lo; r_hi = m ? (uint32_t)(s >> 32) : r_hi;
      }
      if (used & (1u << SURGE_OP_SUB)) {
        const uint64_t s = (uint64_t)__double_as_longlong(a - b);
        const bool m = op == SURGE_OP_SUB;
        r_lo = m ? (uint32_t)s : r_lo; r_hi = m ? (uint32_t)(s >> 32) : r_hi;
      }
      if (used & (1u << SURGE_OP_MIN)) {
        const bool m = op == SURGE_OP_MIN && java_min_takes_b(a, b, x_lo, x_hi);
        r_lo = m ? x_lo : r_lo; r_hi = m ? x_hi : r_hi;
      }
      if (used & (1u << SURGE_OP_MAX)) {
        const bool m = op == SURGE_OP_MAX && java_max_takes_b(a, b, cur_lo, cur_hi);
        r_lo = m ? x_lo : r_lo; r_hi = m ? x_hi : r_hi;
      }
    } else if (ty == SURGE_SLOT_I64) {
      const uint64_t a = ((uint64_t)cur_hi << 32) | cur_lo, b = ((uint64_t)x_hi << 32) | x_lo;
      if (used & (1u << SURGE_OP_ADD)) {
        const uint64_t s = a + b;
        const bool m = op == SURGE_OP_ADD;
        r_lo = m ? (uint32_t)s : r_lo; r_hi = m ? (uint32_t)(s >> 32) : r_hi;
      }
      if (used & (1u << SURGE_OP_SUB)) {
        const uint64_t s = a - b;
        const bool m = op == SURGE_OP_SUB;
        r_lo = m ? (uint32_t)s : r_lo; r_hi = m ? (uint32_t)(s >> 32) : r_hi;
      }
      if (used & (1u << SURGE_OP_MIN)) {
        const bool m = op == SURGE_OP_MIN && (int64_t)b < (int64_t)a;
        r_lo = m ? x_lo : r_lo; r_hi = m ? x_hi : r_hi;
      }
      if (used & (1u << SURGE_OP_MAX)) {
        const bool m = op == SURGE_OP_MAX && (int64_t)b > (int64_t)a;
        r_lo = m ? x_lo : r_lo; r_hi = m ? x_hi : r_hi;
      }
    } else {
      const uint32_t a = cur_lo, b = x_lo;
      if (used & (1u << SURGE_OP_ADD)) r_lo = op == SURGE_OP_ADD ? a + b : r_lo;
      if (used & (1u << SURGE_OP_SUB)) r_lo = op == SURGE_OP_SUB ? a - b : r_lo;
      if (used & (1u << SURGE_OP_MIN)) r_lo = (op == SURGE_OP_MIN && (int32_t)b < (int32_t)a) ? b : r_lo;
      if (used & (1u << SURGE_OP_MAX)) r_lo = (op == SURGE_OP_MAX && (int32_t)b > (int32_t)a) ? b : r_lo;
    }
    if (used & (1u << SURGE_OP_SET)) {
      const bool m = op == SURGE_OP_SET;
      r_lo = m ? x_lo : r_lo; r_hi = m ? x_hi : r_hi;
    }
    st.lo[i] = r_lo;
    st.hi[i] = r_hi;
  }
}

struct SlotEntry { uint4 q; uint2 w; };
__device__ __forceinline__ SlotEntry slot_entry(const uint32_t* lds_tab, uint32_t off) {
  const uint4* te = table_entry(lds_tab, off);
  SlotEntry e;
  e.q = te[0];
  e.w = *(const uint2*)(te + 1);
  return e;
}

// The specialised walk: N events held in registers, table entries prefetched one event ahead (as the v1 walks).
template <int N>
__device__ __forceinline__ void slots_walk_regs(SlotState& st, const uint4* ev, const uint32_t* tyc, const uint32_t* lds_tab,
                                                const SlotSchema& sc) {
  SlotEntry e = slot_entry(lds_tab, tyc[0]);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    SlotEntry n = e;
    if (j + 1 < N) n = slot_entry(lds_tab, tyc[j + 1]);
    slots_apply(st, e.q, e.w.x, e.w.y, ev[j].y, ev[j].z, ev[j].w, sc);
    e = n;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The interpreter's walk: a ROLLED loop over the tile in LDS, each event read when its turn comes (one ahead).
// lo / hi: the lane's live event positions [lo, hi) of this tile (the rest is padding); TYPE_IS_OFFSET: the type word
// already is the table offset (tile-major log).
template <int N, bool TYPE_IS_OFFSET>
__device__ __forceinline__ void slots_walk_lds(SlotState& st, const char* lds_ev, uint32_t ev_row, int32_t lo, int32_t hi,
                                               const uint32_t* lds_tab, const SlotSchema& sc) {
  auto at = [&](int j) { return *(const uint4*)(lds_ev + (j >> 3) * (TYPE_IS_OFFSET ? kSubBytes : 0) + (ev_row ^ (uint32_t)((TYPE_IS_OFFSET ? (j & 7) : j) * 16))); };
  auto off_of = [&](const uint4& e, int j) -> uint32_t {
    if (TYPE_IS_OFFSET) return e.x;
    return (j >= lo && j < hi) ? type_off(e.x) : kNullEntryOffBytes;
  };
  uint4 e_n = at(0);
  SlotEntry t_n = slot_entry(lds_tab, off_of(e_n, 0));
#pragma unroll 2
  for (int j = 0; j < N; ++j) {
    const uint4 e = e_n;
    const SlotEntry t = t_n;
    if (j + 1 < N) {
      e_n = at(j + 1);
      t_n = slot_entry(lds_tab, off_of(e_n, j + 1));
    }
    slots_apply(st, t.q, t.w.x, t.w.y, e.y, e.z, e.w, sc);
  }
}

// ---- transport 1: the CSR log through line-aligned row pieces (sorted-rows; also every micro-batch) ---------------
template <int LE>
__device__ __forceinline__ void fold_slots_csr_body(const FoldParams& p, const SlotSchema& sc, char* smem) {
  using G = Geo<LE>;
  char* lds_ev = smem;
  int64_t* lds_rs = (int64_t*)(smem + G::kTileBytes);
  uint32_t* lds_len = (uint32_t*)(smem + G::kTileBytes + kWave * 8);
  uint32_t* lds_tab = (uint32_t*)(smem + G::kTileBytes + G::kAuxSorted);
  const int lane = threadIdx.x;
  slots_load_table(sc, lds_tab, lane);
  const uint32_t ev_row = G::ev_row(lane);
  const int64_t n_groups = (p.n_seg + kWave - 1) / kWave;
  const int64_t* perm = p.plan;

  auto grab = [&]() -> int64_t {
    unsigned long long g = 0;
    if (lane == 0) g = atomicAdd(p.counter, 1ull);
    return (int64_t)(((uint64_t)rl((uint32_t)(g >> 32), 0) << 32) | rl((uint32_t)g, 0));
  };
  struct Meta { int64_t s, start; uint32_t len, pad; };
  auto load_meta = [&](int64_t g) -> Meta {
    Meta m; m.s = -1; m.start = 0; m.len = 0u; m.pad = 0u;
    const int64_t idx = g * kWave + lane;
    if (g < n_groups && idx < p.n_seg) {
      m.s = perm[idx];
      const int64_t st = p.seg_off[m.s];
      m.pad = (uint32_t)(st & 7);
      m.start = st - m.pad;  // tiled from the 128-byte line that holds the first event (as the sorted-rows kernel)
      m.len = (uint32_t)(p.seg_off[m.s + 1] - st) + m.pad;
    }
    return m;
  };

  int64_t g = grab();
  Meta cur = load_meta(g);
  while (g < n_groups) {
    const int64_t g_next = grab();
    const Meta nxt = load_meta(g_next);
    uint32_t maxlen = cur.len, minlen = cur.s >= 0 ? cur.len : 0xffffffffu;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, d, 64));
      minlen = min(minlen, (uint32_t)__shfl_xor((int)minlen, d, 64));
    }
    const int n_tiles = (int)((maxlen + LE - 1) / LE);
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
    SlotState st = (p.init && oi >= 0) ? slots_load(p.init, oi) : slots_none();
    issue(0);
    for (int c = 0; c < n_tiles; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int32_t rem = (int32_t)cur.len - c * LE;        // my remaining events (may be <= 0)
      const int32_t skip = c == 0 ? (int32_t)cur.pad : 0;  // events in front of my segment
      if constexpr (kSlotsSpec) {
        uint4 ev[LE];
#pragma unroll
        for (int j = 0; j < LE; ++j) ev[j] = *(const uint4*)(lds_ev + (ev_row ^ (uint32_t)(j * 16)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (c + 1 < n_tiles) issue(c + 1);  // in flight during the walk
        uint32_t tyc[LE];
        if (c > 0 && (uint32_t)(c + 1) * LE <= minlen) {
#pragma unroll
          for (int j = 0; j < LE; ++j) tyc[j] = type_off(ev[j].x);
        } else {
#pragma unroll
          for (int j = 0; j < LE; ++j) tyc[j] = (j >= skip && j < rem) ? type_off(ev[j].x) : kNullEntryOffBytes;
        }
        slots_walk_regs<LE>(st, ev, tyc, lds_tab, sc);
      } else {
        // the interpreter is VALU-bound: the next tile is fetched AFTER this one is walked, the other resident waves
        // cover the latency (holding the tile in registers to free the buffer early cost a 45-select chain per event)
        slots_walk_lds<LE, false>(st, lds_ev, ev_row, skip, rem, lds_tab, sc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (c + 1 < n_tiles) issue(c + 1);
      }
    }
    if (oi >= 0) slots_store(p.out, oi, st);
    g = g_next;
    cur = nxt;
  }
  dispenser_leave(p.counter, lane);
}

// ---- transport 2: the tile-major copy of a bound log (fold_tiled.hip's layout; v2 rows are never cut) -------------
template <int SUBS>
__device__ __forceinline__ void fold_slots_tiled_body(const FoldParams& p, const TileTable& t, const SlotSchema& sc, char* lds_ev,
                                                      uint32_t* lds_tab) {
  constexpr int LE = kSubEvents * SUBS;
  constexpr int kLoads = SUBS * (kSubBytes / 1024);
  const int lane = threadIdx.x;
  slots_load_table(sc, lds_tab, lane);
  const uint32_t ev_row = Geo<8>::ev_row(lane);
  const int64_t n_groups = (t.n_vrows + kWave - 1) / kWave;

  auto grab = [&]() -> int64_t {
    unsigned long long g = 0;
    if (lane == 0) g = atomicAdd(p.counter, 1ull);
    return (int64_t)(((uint64_t)rl((uint32_t)(g >> 32), 0) << 32) | rl((uint32_t)g, 0));
  };
  auto load_dest = [&](int64_t g) -> int64_t {
    const int64_t idx = g * kWave + lane;
    return (g < n_groups && idx < t.n_vrows) ? t.v_dest[idx] : -1;
  };
  struct Shape { int64_t sub0; int n_sub; };  // wave-uniform
  auto load_shape = [&](int64_t g) -> Shape {
    Shape s; s.sub0 = 0; s.n_sub = 0;
    if (g < n_groups) {
      const int64_t a = t.g_sub0[g], b = t.g_sub0[g + 1];
      s.sub0 = uniform64(a);
      s.n_sub = (int)__builtin_amdgcn_readfirstlane((uint32_t)(b - a));
    }
    return s;
  };
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
  int64_t dest = load_dest(g);
  Shape sh = load_shape(g);
  if (g < n_groups) issue(sh, 0);
  while (g < n_groups) {
    const int64_t g_next = grab();
    const int64_t dest_next = load_dest(g_next);
    const Shape sh_next = load_shape(g_next);
    const int n_steps = (sh.n_sub + SUBS - 1) / SUBS;
    const bool odd_tail = SUBS == 2 && (sh.n_sub & 1);  // the last step holds one subtile: its second half is stale
    SlotState st = (p.init && dest >= 0) ? slots_load(p.init, dest) : slots_none();
    auto fetch_next = [&](int c) {
      if (c + 1 < n_steps) {
        issue(sh, c + 1);
      } else if (g_next < n_groups) {
        issue(sh_next, 0);  // the next group's first step is fetched while this group's last one is walked
      }
    };
    // PAD slots of the tile-major log carry the null entry's offset: no tail masking here
    for (int c = 0; c < n_steps; ++c) {
      const bool half = odd_tail && c == n_steps - 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (kSlotsSpec) {
        uint4 ev[LE];
#pragma unroll
        for (int j = 0; j < LE; ++j) ev[j] = *(const uint4*)(lds_ev + (j >> 3) * kSubBytes + (ev_row ^ (uint32_t)((j & 7) * 16)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fetch_next(c);
        uint32_t tyc[LE];
#pragma unroll
        for (int j = 0; j < LE; ++j) tyc[j] = (half && j >= kSubEvents) ? kNullEntryOffBytes : ev[j].x;
        slots_walk_regs<LE>(st, ev, tyc, lds_tab, sc);
      } else {
        if (half) slots_walk_lds<kSubEvents, true>(st, lds_ev, ev_row, 0, kSubEvents, lds_tab, sc);
        else slots_walk_lds<LE, true>(st, lds_ev, ev_row, 0, LE, lds_tab, sc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fetch_next(c);
      }
    }
    if (dest >= 0) slots_store(p.out, dest, st);
    g = g_next;
    dest = dest_next;
    sh = sh_next;
  }
  dispenser_leave(p.counter, lane);
}

}  // namespace
}  // namespace surge
