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
//   * foldLeft is sequential, the GPU is not: each lane folds its 16 consecutive events into
//     a *state transformer*  f : Option[State] -> Option[State]  kept as two accumulators
//     (z = f(None), t = f restricted to Some(x), per field "relative" or "absolute").  These
//     transformers form a monoid under composition, so a wave-level segmented scan
//     (segment heads = aggregate boundaries) yields every aggregate's final state while
//     preserving the strict left-to-right event order.  Integer adds wrap mod 2^32 / 2^64
//     exactly like JVM Int/Long, min/max/set are exact, f64 payloads are only moved — so the
//     result is bit-identical to the sequential fold under any association.
//   * A wave task is a contiguous range of WHOLE segments (~256 KiB of events) chosen by a
//     tiny plan kernel from the CSR offsets, so waves never exchange carries through memory:
//     the running state of a segment that spans tiles is carried in scalar registers.
//   * No MFMA: this is an HBM-bound fold (16 B in per event, 64 B out per aggregate).
#include "replay_internal.h"

namespace surge {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr uint32_t FL_PRESENT = 1u;
constexpr uint32_t FL_POISONED = 2u;
constexpr uint32_t FL_HEAD = 16u;
constexpr uint32_t SM_COUNT = 1u << 8;
constexpr uint32_t SM_VERSION = 1u << 9;
constexpr uint32_t SM_SUM = 1u << 10;
constexpr uint32_t SM_BAL = 1u << 11;
constexpr uint32_t SM_MIN = 1u << 12;
constexpr uint32_t SM_MAX = 1u << 13;
constexpr uint32_t SM_N = 1u << 14;
constexpr uint32_t SM_ALL = 0x7Fu << 8;

// One evaluation path of a transformer.  With fl & SM_x the field x holds an absolute value,
// otherwise a value relative to the (unknown) incoming state: a delta for count/sum/n, a clamp
// for min/max, "unchanged" for version/balance.
struct Acc {
  int32_t count, version;
  int64_t sum;
  uint64_t bal;
  int32_t mn, mx;
  uint32_t n, fl;
};

// z: result when the incoming aggregate is None (always absolute: z.fl has SM_ALL).
// t: result when the incoming aggregate is Some(x).  FL_HEAD lives in t.fl.
struct Part {
  Acc z, t;
};

__device__ __forceinline__ Acc acc_none() {
  Acc a;
  a.count = 0; a.version = 0; a.sum = 0; a.bal = 0; a.mn = 0x7fffffff; a.mx = (int32_t)0x80000000; a.n = 0;
  a.fl = SM_ALL;
  return a;
}

__device__ __forceinline__ Acc acc_identity() {
  Acc a = acc_none();
  a.fl = FL_PRESENT;
  return a;
}

// One case of handleEvent (see surge_replay.h for the descriptor semantics and the reference
// lines each presence class restates).
__device__ __forceinline__ void apply_event(Acc& a, uint32_t d, int32_t seq, uint64_t raw, bool valid,
                                            const FoldParams& p) {
  const bool live = valid && !(a.fl & FL_POISONED);
  const bool poison = live && (d & SURGE_D_POISON);
  const bool go = live && !poison;
  const uint32_t cls = d & SURGE_CLS_MASK;
  const bool present = (a.fl & FL_PRESENT) != 0;
  const bool del = go && cls == SURGE_CLS_DELETE;
  const bool app = go && cls != SURGE_CLS_DELETE && (present || cls != SURGE_CLS_REQUIRE);
  const bool rst = app && (cls == SURGE_CLS_CREATE || !present);
  if (poison) a.fl |= FL_POISONED;
  if (del) a.fl = (a.fl & ~FL_PRESENT) | SM_ALL;
  if (rst) {
    a.count = p.d_count; a.version = p.d_version; a.sum = p.d_sum; a.bal = p.d_balance;
    a.mn = p.d_min; a.mx = p.d_max; a.n = p.d_evcount;
    a.fl |= FL_PRESENT | SM_ALL;
  }
  if (app) {
    const int32_t arg = (int32_t)(uint32_t)raw;
    const uint32_t cop = d & SURGE_D_COUNT_MASK;
    if (cop == SURGE_D_COUNT_ADD) a.count = (int32_t)((uint32_t)a.count + (uint32_t)arg);
    if (cop == SURGE_D_COUNT_SUB) a.count = (int32_t)((uint32_t)a.count - (uint32_t)arg);
    if (cop == SURGE_D_COUNT_SET) { a.count = arg; a.fl |= SM_COUNT; }
    if (d & SURGE_D_VERSION_SET) { a.version = seq; a.fl |= SM_VERSION; }
    const uint32_t sop = d & SURGE_D_SUM_MASK;
    if (sop == SURGE_D_SUM_ADD) a.sum = (int64_t)((uint64_t)a.sum + (uint64_t)(int64_t)arg);
    if (sop == SURGE_D_SUM_SUB) a.sum = (int64_t)((uint64_t)a.sum - (uint64_t)(int64_t)arg);
    if (d & SURGE_D_BALANCE_SET) { a.bal = raw; a.fl |= SM_BAL; }
    if (d & SURGE_D_MIN_ARG) a.mn = min(a.mn, arg);
    if (d & SURGE_D_MAX_ARG) a.mx = max(a.mx, arg);
    if (d & SURGE_D_EVCOUNT_INC) a.n += 1u;
  }
}

// g after f on the Some-path: absolute fields of g win, relative ones combine with f's.
__device__ __forceinline__ Acc seq_acc(const Acc& f, const Acc& g) {
  Acc r;
  r.count = (g.fl & SM_COUNT) ? g.count : (int32_t)((uint32_t)f.count + (uint32_t)g.count);
  r.version = (g.fl & SM_VERSION) ? g.version : f.version;
  r.sum = (g.fl & SM_SUM) ? g.sum : (int64_t)((uint64_t)f.sum + (uint64_t)g.sum);
  r.bal = (g.fl & SM_BAL) ? g.bal : f.bal;
  r.mn = (g.fl & SM_MIN) ? g.mn : min(f.mn, g.mn);
  r.mx = (g.fl & SM_MAX) ? g.mx : max(f.mx, g.mx);
  r.n = (g.fl & SM_N) ? g.n : f.n + g.n;
  r.fl = (g.fl & (FL_PRESENT | FL_POISONED)) | ((f.fl | g.fl) & SM_ALL);
  return r;
}

__device__ __forceinline__ Acc select_acc(bool c, const Acc& a, const Acc& b) {
  Acc r;
  r.count = c ? a.count : b.count; r.version = c ? a.version : b.version;
  r.sum = c ? a.sum : b.sum; r.bal = c ? a.bal : b.bal;
  r.mn = c ? a.mn : b.mn; r.mx = c ? a.mx : b.mx; r.n = c ? a.n : b.n; r.fl = c ? a.fl : b.fl;
  return r;
}

// Segmented composition: f is the earlier (left) element.
__device__ __forceinline__ Part combine(const Part& f, const Part& g) {
  const bool g_head = (g.t.fl & FL_HEAD) != 0;
  const bool f_poisoned = (f.t.fl & FL_POISONED) != 0;  // z and t are poisoned together
  const uint32_t f_head = f.t.fl & FL_HEAD;
  Acc cz = select_acc((f.z.fl & FL_PRESENT) != 0, seq_acc(f.z, g.t), g.z);
  Acc ct = select_acc((f.t.fl & FL_PRESENT) != 0, seq_acc(f.t, g.t), g.z);
  ct.fl |= f_head;
  const bool keep_f = f_poisoned && !g_head;
  Part r;
  r.z = select_acc(g_head, g.z, select_acc(keep_f, f.z, cz));
  r.t = select_acc(g_head, g.t, select_acc(keep_f, f.t, ct));
  return r;
}

__device__ __forceinline__ Acc shfl_up_acc(const Acc& a, int d) {
  Acc r;
  r.count = __shfl_up(a.count, d, 64); r.version = __shfl_up(a.version, d, 64);
  r.sum = __shfl_up(a.sum, d, 64); r.bal = __shfl_up(a.bal, d, 64);
  r.mn = __shfl_up(a.mn, d, 64); r.mx = __shfl_up(a.mx, d, 64);
  r.n = __shfl_up(a.n, d, 64); r.fl = __shfl_up(a.fl, d, 64);
  return r;
}

__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

__device__ __forceinline__ Acc readlane_acc(const Acc& a, int lane) {
  Acc r;
  r.count = (int32_t)rl((uint32_t)a.count, lane); r.version = (int32_t)rl((uint32_t)a.version, lane);
  r.sum = (int64_t)(((uint64_t)rl((uint32_t)((uint64_t)a.sum >> 32), lane) << 32) | rl((uint32_t)a.sum, lane));
  r.bal = ((uint64_t)rl((uint32_t)(a.bal >> 32), lane) << 32) | rl((uint32_t)a.bal, lane);
  r.mn = (int32_t)rl((uint32_t)a.mn, lane); r.mx = (int32_t)rl((uint32_t)a.mx, lane);
  r.n = rl(a.n, lane); r.fl = rl(a.fl, lane);
  return r;
}

__device__ __forceinline__ void store_state(uint4* out, int64_t idx, const Acc& a) {
  const bool pr = (a.fl & FL_PRESENT) != 0;
  uint4 v0, v1, v2, v3;
  v0.x = pr ? (uint32_t)a.count : 0u; v0.y = pr ? (uint32_t)a.version : 0u;
  v0.z = pr ? (uint32_t)a.sum : 0u; v0.w = pr ? (uint32_t)((uint64_t)a.sum >> 32) : 0u;
  v1.x = pr ? (uint32_t)a.bal : 0u; v1.y = pr ? (uint32_t)(a.bal >> 32) : 0u;
  v1.z = pr ? (uint32_t)a.mn : 0u; v1.w = pr ? (uint32_t)a.mx : 0u;
  v2.x = pr ? a.n : 0u; v2.y = a.fl & (FL_PRESENT | FL_POISONED); v2.z = 0u; v2.w = 0u;
  v3.x = v3.y = v3.z = v3.w = 0u;
  uint4* o = out + idx * 4;
  o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3;
}

__device__ __forceinline__ Acc load_state(const uint4* in, int64_t idx) {
  const uint4* s = in + idx * 4;
  const uint4 v0 = s[0], v1 = s[1], v2 = s[2];
  Acc a;
  a.count = (int32_t)v0.x; a.version = (int32_t)v0.y;
  a.sum = (int64_t)(((uint64_t)v0.w << 32) | v0.z);
  a.bal = ((uint64_t)v1.y << 32) | v1.x;
  a.mn = (int32_t)v1.z; a.mx = (int32_t)v1.w; a.n = v2.x;
  a.fl = (v2.y & (FL_PRESENT | FL_POISONED)) | SM_ALL;
  return a;
}

// Direct global->LDS load of one 16 KiB tile.  Instruction q writes LDS bytes [q*1024, q*1024+1024)
// linearly by lane (that is what the hardware does); WHICH event a lane fetches is ours to choose:
// LDS slot (q*64 + m) belongs to chunk-lane l = 4q + (m >> 4) and holds its event j = ((m & 15) - l) & 15.
// Every instruction still covers one contiguous, fully used 1 KiB of the log.
__device__ __forceinline__ void issue_tile_loads(const FoldParams& p, int64_t te0, char* lds, int lane) {
  const int64_t last = p.n_events - 1;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int l = 4 * q + (lane >> 4);
    const int j = ((lane & 15) - l) & 15;
    int64_t e = te0 + l * 16 + j;
    e = e < last ? e : last;
    __builtin_amdgcn_global_load_lds((gptr_t)(p.events + e), (lptr_t)(lds + q * 1024), 16, 0, 0);
  }
}

enum { MODE_FIXED = 0, MODE_FLAT = 1 };

template <int MODE>
__global__ void __launch_bounds__(kWave) fold_kernel(const FoldParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_ev = smem;
  uint32_t* lds_hb = (uint32_t*)(smem + kTileBytes);
  uint32_t* lds_desc = (uint32_t*)(smem + kTileBytes + kHeadWords * 4);

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
    S0 = p.plan[task];
    S1 = p.plan[task + 1];
    if (S0 >= S1) return;
    E0 = p.seg_off[S0];
    E1 = p.seg_off[S1];
  }
  if (E1 <= E0) return;

  if (lane < 17) lds_desc[lane] = p.desc[lane];
  if (MODE == MODE_FLAT && lane < kHeadWords) lds_hb[lane] = 0u;

  const int n_tiles = (int)((E1 - E0 + kTileEvents - 1) / kTileEvents);
  issue_tile_loads(p, E0, lds_ev, lane);

  // FLAT: head marking.  next_s = first segment whose start has not been marked yet.
  int64_t next_s = S0;
  auto mark_heads = [&](int64_t te0) {
    const int64_t te1 = (te0 + kTileEvents < E1) ? te0 + kTileEvents : E1;
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
  if (MODE == MODE_FLAT) mark_heads(E0);

  // The running segment that enters the next tile; starts as "nothing" (a head, None).
  Part carry;
  carry.z = acc_none();
  carry.t = acc_none();
  carry.t.fl |= FL_HEAD;
  int64_t c = S0 - 1;  // FLAT: index of the segment open when the tile starts

  for (int tile = 0; tile < n_tiles; ++tile) {
    const int64_t te0 = E0 + (int64_t)tile * kTileEvents;

    // tile `tile` has landed in LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint4 ev[kLaneEvents];
#pragma unroll
    for (int j = 0; j < kLaneEvents; ++j)
      ev[j] = *(const uint4*)(lds_ev + lane * 256 + ((j + lane) & 15) * 16);

    uint32_t hb;          // bit j: my event j starts a new segment
    int64_t seg_open;     // segment open when my chunk starts (before a head at j = 0)
    int heads_in_tile = 0;
    if (MODE == MODE_FLAT) {
      hb = (lds_hb[lane >> 1] >> ((lane & 1) * 16)) & 0xffffu;
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
      const uint32_t e_rel = (uint32_t)(te0 - E0) + (uint32_t)lane * kLaneEvents;
      const uint32_t q = e_rel / L;
      const uint32_t r = e_rel - q * L;
      hb = (r == 0u && te0 + lane * kLaneEvents < E1) ? 1u : 0u;
      seg_open = S0 - 1 + (int64_t)q + (r != 0u ? 1 : 0);
    }
    uint32_t desc[kLaneEvents];
#pragma unroll
    for (int j = 0; j < kLaneEvents; ++j) {
      const uint32_t ty = ev[j].x;
      desc[j] = lds_desc[ty < 16u ? ty : 16u];
    }
    // all my LDS reads are done: the buffer can take the next tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == MODE_FLAT && lane < kHeadWords) lds_hb[lane] = 0u;
    if (tile + 1 < n_tiles) issue_tile_loads(p, te0 + kTileEvents, lds_ev, lane);

    // ---- per-lane sequential walk over 16 consecutive events -------------------------------
    int64_t rem = E1 - (te0 + (int64_t)lane * kLaneEvents);
    const int nvalid = rem >= kLaneEvents ? kLaneEvents : (rem > 0 ? (int)rem : 0);
    Acc z = acc_none();
    Acc t = acc_identity();
    Acc lead_z = z;
    bool seen = false;
    const int64_t lead_seg = seg_open;
#pragma unroll
    for (int j = 0; j < kLaneEvents; ++j) {
      if ((hb >> j) & 1u) {
        if (!seen) {
          lead_z = z;
          seen = true;
        } else {
          const int64_t oi = p.out_map ? p.out_map[seg_open] : seg_open;
          store_state(p.out, oi, z);
        }
        seg_open += 1;
        if (p.init) {
          const int64_t ii = p.out_map ? p.out_map[seg_open] : seg_open;
          z = load_state(p.init, ii);
        } else {
          z = acc_none();
        }
      }
      const bool valid = j < nvalid;
      const uint64_t raw = ((uint64_t)ev[j].w << 32) | ev[j].z;
      apply_event(z, desc[j], (int32_t)ev[j].y, raw, valid, p);
      if (!seen) apply_event(t, desc[j], (int32_t)ev[j].y, raw, valid, p);
    }

    // ---- wave-level segmented scan of the lane transformers --------------------------------
    Part lead;           // my leading partial (events before my first head), if I saw a head
    lead.z = lead_z;
    lead.t = t;
    Part el;
    el.z = z;
    el.t = seen ? z : t;
    if (seen) el.t.fl |= FL_HEAD;
    {
      const Part seeded = combine(carry, el);
      if (lane == 0) el = seeded;
    }
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      Part o;
      o.z = shfl_up_acc(el.z, d);
      o.t = shfl_up_acc(el.t, d);
      const Part cmb = combine(o, el);
      if (lane >= d) el = cmb;
    }
    Part prefix;
    prefix.z = shfl_up_acc(el.z, 1);
    prefix.t = shfl_up_acc(el.t, 1);
    if (lane == 0) prefix = carry;

    if (seen && lead_seg >= S0) {
      const Part fin = combine(prefix, lead);
      const int64_t oi = p.out_map ? p.out_map[lead_seg] : lead_seg;
      store_state(p.out, oi, fin.z);
    }
    carry.z = readlane_acc(el.z, 63);
    carry.t = readlane_acc(el.t, 63);

    if (MODE == MODE_FLAT) {
      c += heads_in_tile;
      if (tile + 1 < n_tiles) mark_heads(te0 + kTileEvents);
    }
  }

  if (lane == 0) {
    const int64_t oi = p.out_map ? p.out_map[S1 - 1] : (S1 - 1);
    store_state(p.out, oi, carry.z);
  }
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

// ---- K4: shard map  abs(MurmurHash3.stringHash(id.takeWhile(_ != ':')) % n)
// (modules/common/src/main/scala/surge/kafka/KafkaPartitioner.scala:8,38-42; scala-library 2.13.8 algorithm)
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__global__ void partition_hash_kernel(const uint16_t* __restrict__ utf16, const int64_t* __restrict__ str_off, int64_t n,
                                      int32_t n_partitions, int32_t* __restrict__ part_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint16_t* s = utf16 + str_off[i];
  const int64_t full = str_off[i + 1] - str_off[i];
  int64_t len = 0;
  while (len < full && s[len] != (uint16_t)':') ++len;
  uint32_t h = 0xf7ca7fd2u;
  int64_t k = 0;
  for (; k + 1 < len; k += 2) {
    uint32_t d = ((uint32_t)s[k] << 16) + (uint32_t)s[k + 1];
    d *= 0xcc9e2d51u; d = rotl32(d, 15); d *= 0x1b873593u;
    h ^= d; h = rotl32(h, 13); h = h * 5u + 0xe6546b64u;
  }
  if (k < len) {
    uint32_t d = (uint32_t)s[k];
    d *= 0xcc9e2d51u; d = rotl32(d, 15); d *= 0x1b873593u;
    h ^= d;
  }
  h ^= (uint32_t)len;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  const int32_t r = (int32_t)h % n_partitions;  // truncated, like the JVM
  part_out[i] = r < 0 ? -r : r;
}

// ---- HBM read-stream ceiling probe -------------------------------------------------------------
__global__ void __launch_bounds__(256) stream_probe_kernel(const uint4* __restrict__ src, int64_t n_vec,
                                                           uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n_vec; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c2 = src[i + 2 * stride], d = src[i + 3 * stride];
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c2.x ^ c2.y ^ c2.z ^ c2.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n_vec; i += stride) {
    const uint4 a = src[i];
    acc ^= a.x ^ a.y ^ a.z ^ a.w;
  }
  if (acc == 0x9e3779b9u) sink[0] = acc;  // practically never; keeps the loads alive
}

__global__ void count_poisoned_kernel(const uint4* __restrict__ states, int64_t n, unsigned long long* count) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool p = s < n && (states[s * 4 + 2].y & FL_POISONED);
  const int c = __popcll(__ballot(p));
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}

}  // namespace

hipError_t launch_fold_fixed(const FoldParams& p, int64_t n_tasks, hipStream_t stream) {
  if (n_tasks <= 0) return hipSuccess;
  hipLaunchKernelGGL(fold_kernel<MODE_FIXED>, dim3((unsigned)n_tasks), dim3(kWave), kFoldLdsBytes, stream, p);
  return hipGetLastError();
}

hipError_t launch_fold_flat(const FoldParams& p, int64_t n_tasks, hipStream_t stream) {
  if (n_tasks <= 0) return hipSuccess;
  hipLaunchKernelGGL(fold_kernel<MODE_FLAT>, dim3((unsigned)n_tasks), dim3(kWave), kFoldLdsBytes, stream, p);
  return hipGetLastError();
}

hipError_t launch_plan(const int64_t* off, int64_t n_seg, int64_t task_events, int64_t n_tasks, int64_t* plan,
                       hipStream_t stream) {
  const int64_t n = n_tasks + 1;
  hipLaunchKernelGGL(plan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, off, n_seg, task_events,
                     n_tasks, plan);
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

hipError_t launch_partition_hash(const uint16_t* utf16, const int64_t* str_off, int64_t n, int32_t n_partitions,
                                 int32_t* part_out, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(partition_hash_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, utf16, str_off, n,
                     n_partitions, part_out);
  return hipGetLastError();
}

hipError_t launch_stream_probe(const uint4* src, int64_t n_vec, uint32_t* sink, hipStream_t stream) {
  hipLaunchKernelGGL(stream_probe_kernel, dim3(256 * 8), dim3(256), 0, stream, src, n_vec, sink);
  return hipGetLastError();
}

hipError_t launch_count_poisoned(const uint4* states, int64_t n, unsigned long long* d_count, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream);
  if (e != hipSuccess || n <= 0) return e;
  hipLaunchKernelGGL(count_poisoned_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, states, n, d_count);
  return hipGetLastError();
}

}  // namespace surge
