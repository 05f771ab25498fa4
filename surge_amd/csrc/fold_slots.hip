// fold_slots.hip — the fold for ABI v2 "slot" schemas (include/surge_replay.h): up to 7 typed 8-byte slots, one
// operation per slot per event type, ANY mix of operations (ADD / SUB / SET / MIN / MAX, integer or IEEE double).
//
// The v1 kernels split aggregates across lanes and waves and therefore need every field to compose associatively; a
// generic slot schema promises nothing of the sort (an f64 ADD is order-sensitive, ADD followed by MIN followed by ADD
// on one slot has no closed form).  So this kernel never splits an aggregate: ONE lane walks ONE aggregate's events
// (or one micro-batch group's) strictly in order with a concrete running state — the sorted-rows transport (length
// sort at load time, persistent waves pulling groups of 64, LDS-DMA tiles with line-aligned row pieces) around a slot
// interpreter.  That also makes f64 accumulation bit-identical to the JVM's sequential fold (no tolerance).
//
// Interpreter cost control: slot type, operand source and the SET of operations any event type ever applies to a slot
// are wave-uniform (schema constants), so they are scalar branches; only the operation an individual event applies is
// per lane (a 4-bit code from a 16-entry LDS table), resolved by selects among the candidates the schema allows.
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

static_assert(sizeof(SlotParams) <= kSlotParamsBytes, "grow kSlotParamsBytes");

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr uint32_t CLS_NULL = 1u << 31;  // the padding event: identity on every state

struct SlotState {
  uint64_t s[SURGE_MAX_SLOTS];
  uint32_t evc, fl;
};

__device__ __forceinline__ SlotState slots_none() {
  SlotState st;
#pragma unroll
  for (int i = 0; i < SURGE_MAX_SLOTS; ++i) st.s[i] = 0ull;
  st.evc = 0u;
  st.fl = 0u;
  return st;
}

__device__ __forceinline__ SlotState slots_load(const uint4* in, int64_t idx) {
  const uint64_t* w = (const uint64_t*)(in + idx * 4);
  SlotState st;
  st.s[0] = w[0]; st.s[1] = w[1]; st.s[2] = w[2]; st.s[3] = w[3];
  const uint64_t m = w[4];
  st.evc = (uint32_t)m;
  st.fl = (uint32_t)(m >> 32) & (FL_PRESENT | FL_POISONED);
  st.s[4] = w[5]; st.s[5] = w[6]; st.s[6] = w[7];
  return st;
}

__device__ __forceinline__ void slots_store(uint4* out, int64_t idx, const SlotState& st) {
  const bool pr = (st.fl & FL_PRESENT) != 0u;  // None is canonically all-zero (plus, possibly, the POISONED flag)
  uint4* o = out + idx * 4;
  auto lo = [&](int i) { return pr ? (uint32_t)st.s[i] : 0u; };
  auto hi = [&](int i) { return pr ? (uint32_t)(st.s[i] >> 32) : 0u; };
  o[0] = make_uint4(lo(0), hi(0), lo(1), hi(1));
  o[1] = make_uint4(lo(2), hi(2), lo(3), hi(3));
  o[2] = make_uint4(pr ? st.evc : 0u, st.fl & (FL_PRESENT | FL_POISONED), lo(4), hi(4));
  o[3] = make_uint4(lo(5), hi(5), lo(6), hi(6));
}

// one handleEvent step on a concrete state.  cls / ops: the event type's table words (per lane).
__device__ __forceinline__ void slots_apply(SlotState& st, bool& frozen, uint32_t cls, uint32_t ops, uint32_t seq, uint32_t raw_lo,
                                            uint32_t raw_hi, const SlotParams& p) {
  const bool is_null = (cls & CLS_NULL) != 0u;
  const bool live = !frozen && !is_null;
  const bool throws = live && (cls & SURGE_D_POISON);
  frozen = frozen || throws;
  if (throws) st.fl |= FL_POISONED;
  const uint32_t c = cls & SURGE_CLS_MASK;
  const bool go = live && !throws;
  const bool present = (st.fl & FL_PRESENT) != 0u;
  const bool del = go && c == SURGE_CLS_DELETE;
  const bool app = go && c != SURGE_CLS_DELETE && (present || c != SURGE_CLS_REQUIRE);  // REQUIRE-class events skip None
  const bool rst = app && (c == SURGE_CLS_CREATE || !present);                          // CREATE, or materialising from None
  if (del) st.fl &= ~FL_PRESENT;
  if (app) st.fl |= FL_PRESENT;
  if (rst) st.evc = 0u;
  if (app && p.count_events) st.evc += 1u;
#pragma unroll
  for (int i = 0; i < SURGE_MAX_SLOTS; ++i) {
    if (i >= (int)p.n_slots) break;  // wave-uniform
    uint64_t cur = rst ? p.def[i] : st.s[i];
    const uint32_t ty = p.type[i], src = p.source[i], used = p.used_ops[i];  // wave-uniform
    const uint32_t op = (ops >> (4 * i)) & 15u;                              // per lane
    // the operand, in the slot's own type
    uint64_t x;
    if (ty == SURGE_SLOT_F64) {
      double d;
      if (src == SURGE_SRC_PAYLOAD) d = __longlong_as_double((long long)(((uint64_t)raw_hi << 32) | raw_lo));
      else if (src == SURGE_SRC_ONE) d = 1.0;
      else d = (double)(int32_t)(src == SURGE_SRC_SEQ ? seq : raw_lo);
      x = (uint64_t)__double_as_longlong(d);
    } else {
      int64_t v;
      if (src == SURGE_SRC_PAYLOAD) v = (int64_t)(((uint64_t)raw_hi << 32) | raw_lo);
      else if (src == SURGE_SRC_ONE) v = 1;
      else v = (int64_t)(int32_t)(src == SURGE_SRC_SEQ ? seq : raw_lo);
      x = ty == SURGE_SLOT_I32 ? (uint64_t)(uint32_t)v : (uint64_t)v;
    }
    uint64_t r = cur;
    if (ty == SURGE_SLOT_F64) {
      const double a = __longlong_as_double((long long)cur), b = __longlong_as_double((long long)x);
      if (used & (1u << SURGE_OP_ADD)) r = op == SURGE_OP_ADD ? (uint64_t)__double_as_longlong(a + b) : r;
      if (used & (1u << SURGE_OP_SUB)) r = op == SURGE_OP_SUB ? (uint64_t)__double_as_longlong(a - b) : r;
      if (used & (1u << SURGE_OP_MIN)) r = (op == SURGE_OP_MIN && b < a) ? x : r;
      if (used & (1u << SURGE_OP_MAX)) r = (op == SURGE_OP_MAX && b > a) ? x : r;
    } else if (ty == SURGE_SLOT_I64) {
      if (used & (1u << SURGE_OP_ADD)) r = op == SURGE_OP_ADD ? cur + x : r;
      if (used & (1u << SURGE_OP_SUB)) r = op == SURGE_OP_SUB ? cur - x : r;
      if (used & (1u << SURGE_OP_MIN)) r = (op == SURGE_OP_MIN && (int64_t)x < (int64_t)cur) ? x : r;
      if (used & (1u << SURGE_OP_MAX)) r = (op == SURGE_OP_MAX && (int64_t)x > (int64_t)cur) ? x : r;
    } else {
      const uint32_t a = (uint32_t)cur, b = (uint32_t)x;
      uint32_t q = a;
      if (used & (1u << SURGE_OP_ADD)) q = op == SURGE_OP_ADD ? a + b : q;
      if (used & (1u << SURGE_OP_SUB)) q = op == SURGE_OP_SUB ? a - b : q;
      if (used & (1u << SURGE_OP_MIN)) q = (op == SURGE_OP_MIN && (int32_t)b < (int32_t)a) ? b : q;
      if (used & (1u << SURGE_OP_MAX)) q = (op == SURGE_OP_MAX && (int32_t)b > (int32_t)a) ? b : q;
      r = (uint64_t)q;
    }
    if (used & (1u << SURGE_OP_SET)) r = op == SURGE_OP_SET ? x : r;
    st.s[i] = app ? r : st.s[i];
  }
}

template <int LE>
__global__ void __launch_bounds__(kWave) fold_slots_kernel(const FoldParams p, const SlotParams sp) {
  using G = Geo<LE>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_ev = smem;
  int64_t* lds_rs = (int64_t*)(smem + G::kTileBytes);
  uint32_t* lds_len = (uint32_t*)(smem + G::kTileBytes + kWave * 8);
  uint32_t* lds_cls = (uint32_t*)(smem + G::kTileBytes + G::kAuxSorted);  // 18 class words then 18 op words
  uint32_t* lds_ops = lds_cls + SURGE_MAX_EVENT_TYPES + 2;
  const int lane = threadIdx.x;
  if (lane < SURGE_MAX_EVENT_TYPES + 2) {
    lds_cls[lane] = sp.cls[lane];
    lds_ops[lane] = sp.ops[lane];
  }
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
#pragma unroll
      for (int q = 0; q < G::kLoads; ++q) {  // never read past a row's own events
        const int r = G::kRowsPerLoad * q + lane / LE;
        const uint32_t rlen = lds_len[r];
        uint32_t j = (uint32_t)c * LE + G::load_j(lane, q % G::kClasses);
        const uint32_t lastj = rlen ? rlen - 1u : 0u;
        j = j < lastj ? j : lastj;
        __builtin_amdgcn_global_load_lds((gptr_t)(p.events + (lds_rs[r] + j)), (lptr_t)(lds_ev + q * 1024), 16, 0, kLoadAux);
      }
    };
    const int64_t oi = cur.s >= 0 ? (p.out_map ? p.out_map[cur.s] : cur.s) : -1;
    SlotState st = (p.init && oi >= 0) ? slots_load(p.init, oi) : slots_none();
    bool frozen = (st.fl & FL_POISONED) != 0u;
    issue(0);
    for (int c = 0; c < n_tiles; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // The interpreter is a ROLLED loop over my LE events, each read from LDS when its turn comes (one ahead): unrolling
      // 16 events x 7 slots x 3 value types made ~190 KB of code, far beyond the instruction cache, and holding the tile in
      // registers to free the buffer early cost a 45-select chain per event.  So, unlike the v1 kernels, the next tile is
      // fetched AFTER this one is walked — the interpreter is VALU-bound, the other resident waves cover the latency.
      const int32_t rem = (int32_t)cur.len - c * LE;
      const int32_t skip = c == 0 ? (int32_t)cur.pad : 0;
      uint4 e_n = *(const uint4*)(lds_ev + ev_row);
      uint32_t t_n = (0 >= skip && 0 < rem) ? (e_n.x < 16u ? e_n.x : 16u) : 17u;
      uint32_t cls_n = lds_cls[t_n], ops_n = lds_ops[t_n];
#pragma unroll 2
      for (int j = 0; j < LE; ++j) {
        const uint4 e = e_n;
        const uint32_t cls = cls_n, ops = ops_n;
        if (j + 1 < LE) {
          e_n = *(const uint4*)(lds_ev + (ev_row ^ (uint32_t)((j + 1) * 16)));
          t_n = (j + 1 >= skip && j + 1 < rem) ? (e_n.x < 16u ? e_n.x : 16u) : 17u;
          cls_n = lds_cls[t_n];
          ops_n = lds_ops[t_n];
        }
        slots_apply(st, frozen, cls, ops, e.y, e.z, e.w, sp);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (c + 1 < n_tiles) issue(c + 1);
    }
    if (oi >= 0) slots_store(p.out, oi, st);
    g = g_next;
    cur = nxt;
  }
  dispenser_leave(p.counter, lane);
}

}  // namespace

void slot_params_from_schema(const surge_replay_schema_v2& sc, SlotParams* out) {
  SlotParams p{};
  p.n_slots = sc.n_slots;
  p.count_events = (sc.flags & SURGE_V2_COUNT_EVENTS) ? 1u : 0u;
  for (uint32_t i = 0; i < SURGE_MAX_SLOTS; ++i) {
    p.type[i] = i < sc.n_slots ? sc.slot[i].type : 0u;
    p.source[i] = i < sc.n_slots ? sc.slot[i].source : 0u;
    p.def[i] = i < sc.n_slots ? (sc.slot[i].type == SURGE_SLOT_I32 ? (uint64_t)(uint32_t)sc.slot[i].default_bits : sc.slot[i].default_bits) : 0ull;
    p.used_ops[i] = 0u;
  }
  for (uint32_t t = 0; t < SURGE_MAX_EVENT_TYPES + 2; ++t) {
    p.cls[t] = SURGE_D_POISON;  // unused types and [16]: no such case = MatchError
    p.ops[t] = 0u;
  }
  for (uint32_t t = 0; t < sc.n_types && t < SURGE_MAX_EVENT_TYPES; ++t) {
    p.cls[t] = sc.cls[t] & (SURGE_CLS_MASK | SURGE_D_POISON);
    p.ops[t] = sc.ops[t];
    for (uint32_t i = 0; i < sc.n_slots; ++i) p.used_ops[i] |= 1u << ((sc.ops[t] >> (4 * i)) & 15u);
  }
  p.cls[SURGE_MAX_EVENT_TYPES + 1] = 1u << 31;  // [17]: the null event
  *out = p;
}

hipError_t launch_fold_slots(const FoldParams& p, const SlotParams& sp, int64_t n_waves, int lane_events, hipStream_t stream) {
  if (n_waves <= 0) return hipSuccess;
  if (lane_events == 8)
    hipLaunchKernelGGL((fold_slots_kernel<8>), dim3((unsigned)n_waves), dim3(kWave), Geo<8>::lds_bytes(Geo<8>::kAuxSorted), stream, p, sp);
  else
    hipLaunchKernelGGL((fold_slots_kernel<16>), dim3((unsigned)n_waves), dim3(kWave), Geo<16>::lds_bytes(Geo<16>::kAuxSorted), stream, p, sp);
  return hipGetLastError();
}

}  // namespace surge
