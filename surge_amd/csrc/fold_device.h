// fold_device.h — device-side pieces shared by the fold kernels (fold_kernels.hip, fold_chunked.hip,
// stream_kernels.hip): the lane transformer `Acc`, one case of handleEvent as mask arithmetic (apply_event),
// transformer composition, the LDS tile geometry and the per-lane event walk.  Everything here is
// `static`/inline in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#ifdef __HIPCC_RTC__  // run-time compiled kernels (fold_slots.hip): device code only, no host declarations
#include "fold_layout.h"
#else
#include "replay_internal.h"
#endif

namespace surge {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Cache policy of the event-stream loads: the log is read exactly once, so mark it non-temporal
// (aux bit 1 = nt on gfx950 global_load_lds).
#ifndef SURGE_LOAD_AUX
#define SURGE_LOAD_AUX 2
#endif
constexpr int kLoadAux = SURGE_LOAD_AUX;

// A build specialised for one v1 schema (hiprtc; fold_kernels.hip writes the program): the op table is not data but 16
// compile-time bit masks over the 18 table entries — SURGE_V1_MASK(k) has bit e set when word k of entry e is non-zero
// (every walk word is all-ones or zero, TW_EVC 0 or 1; TW_FLAGS is split into SURGE_V1_THROWS / SURGE_V1_DELETES) — so an
// event's reference is its entry INDEX, a table word is one v_bfe_i32 of a constant, and a word that is zero for every
// entry is the constant 0: the arithmetic of a field no event type touches folds away.  No LDS copy of the table.
#ifdef SURGE_V1_SPEC
constexpr bool kSpecV1 = true;
#else
constexpr bool kSpecV1 = false;
#define SURGE_V1_MASK(k) 0u
#define SURGE_V1_THROWS 0u
#define SURGE_V1_DELETES 0u
#endif
// Fields no event type of the schema touches (Counter: everything but count and version).  Such a field of a present
// aggregate is either its default — a reset (CREATE, or materialising from None) happened since the state the fold started
// from — or still the prior state's value; one flag bit (FL_DFLT) says which, so the field itself is not carried through
// the walk, the scan and the carries at all: store_state_flat writes the default or copies the prior's bytes.
constexpr bool kLiveCount = !kSpecV1 || (SURGE_V1_MASK(TW_CNT_NZ) | SURGE_V1_MASK(TW_CNT_SET)) != 0u;
constexpr bool kLiveVersion = !kSpecV1 || SURGE_V1_MASK(TW_VER_SET) != 0u;
constexpr bool kLiveSum = !kSpecV1 || SURGE_V1_MASK(TW_SUM_NZ) != 0u;
constexpr bool kLiveBal = !kSpecV1 || SURGE_V1_MASK(TW_BAL_SET) != 0u;
constexpr bool kLiveMin = !kSpecV1 || SURGE_V1_MASK(TW_MIN) != 0u;
constexpr bool kLiveMax = !kSpecV1 || SURGE_V1_MASK(TW_MAX) != 0u;
constexpr bool kLiveN = !kSpecV1 || SURGE_V1_MASK(TW_EVC) != 0u;
constexpr bool kAnyDead = !(kLiveCount && kLiveVersion && kLiveSum && kLiveBal && kLiveMin && kLiveMax && kLiveN);
constexpr uint32_t FL_PRESENT = 1u;
constexpr uint32_t FL_POISONED = 2u;
constexpr uint32_t FL_DFLT = 4u;  // (specialised builds with untouched fields only) the untouched fields hold their defaults
constexpr uint32_t FL_HEAD = 16u;
constexpr uint32_t SM_COUNT = 1u << 8;
constexpr uint32_t SM_VERSION = 1u << 9;
constexpr uint32_t SM_BAL = 1u << 11;
constexpr uint32_t SM_ALL = 0x7Fu << 8;  // count, version, sum, balance, min, max, event_count

// A lane's transformer.  With fl & SM_x the field x holds an absolute value, otherwise a value
// relative to the incoming state.  sum64/min/max/event_count only become absolute through a
// reset (SM_ALL); count/version/balance also through their SET ops.
struct Acc {
  int32_t count, version;
  int64_t sum;
  uint64_t bal;
  int32_t mn, mx;
  uint32_t n, fl;
};

__device__ __forceinline__ Acc acc_none() {  // the aggregate is None (absolute)
  Acc a;
  a.count = 0; a.version = 0; a.sum = 0; a.bal = 0; a.mn = 0x7fffffff; a.mx = (int32_t)0x80000000; a.n = 0;
  a.fl = SM_ALL | (kAnyDead ? FL_DFLT : 0u);
  return a;
}

__device__ __forceinline__ Acc acc_identity() {  // "whatever came in", present
  Acc a = acc_none();
  a.fl = FL_PRESENT;
  return a;
}

__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

__device__ __forceinline__ uint32_t andn(uint32_t x, uint32_t m) { return bfi(m, 0u, x); }  // x & ~m in one v_bfi

// One case of handleEvent applied to one evaluation path, as pure VALU mask arithmetic: every
// "condition" is an all-ones / all-zero dword (table words q0..q3, see TW_* in fold_layout.h),
// selects are v_bfi_b32, nothing touches the scalar unit.  frozenM: events are being ignored
// (the aggregate is poisoned).  validM: this event exists (tail of the last tile).
__device__ __forceinline__ void apply_event(Acc& a, uint32_t& frozenM, uint32_t& corr, const uint4 q0, const uint4 q1,
                                            const uint4 q2, const uint2 q3, uint32_t seq, uint32_t raw_lo,
                                            uint32_t raw_hi, const FoldParams& p) {
  const uint32_t ispM = andn(q2.x, frozenM);                            // throws (and is not ignored)
  const uint32_t goM = ~(frozenM | q2.x);
  const uint32_t presentM = (uint32_t)__builtin_amdgcn_sbfe((int32_t)a.fl, 0, 1);
  const uint32_t delM = goM & q2.y;
  const uint32_t appM = andn(goM, q2.y) & (presentM | q2.z);            // REQUIRE-class events skip None
  const uint32_t rstM = appM & bfi(presentM, q2.w, ~0u);                // CREATE, or materialising from None
  frozenM |= ispM;

  uint32_t fl = a.fl | (ispM & FL_POISONED);
  fl = andn(fl, delM & FL_PRESENT) | (delM & SM_ALL);
  fl |= rstM & (FL_PRESENT | SM_ALL | (kAnyDead ? FL_DFLT : 0u));

  uint32_t count = bfi(rstM, (uint32_t)p.d_count, (uint32_t)a.count);
  uint32_t version = bfi(rstM, (uint32_t)p.d_version, (uint32_t)a.version);
  uint32_t sum_lo = bfi(rstM, (uint32_t)p.d_sum, (uint32_t)a.sum);
  uint32_t sum_hi = bfi(rstM, (uint32_t)((uint64_t)p.d_sum >> 32), (uint32_t)((uint64_t)a.sum >> 32));
  uint32_t bal_lo = bfi(rstM, (uint32_t)p.d_balance, (uint32_t)a.bal);
  uint32_t bal_hi = bfi(rstM, (uint32_t)(p.d_balance >> 32), (uint32_t)(a.bal >> 32));
  uint32_t mn = bfi(rstM, (uint32_t)p.d_min, (uint32_t)a.mn);
  uint32_t mx = bfi(rstM, (uint32_t)p.d_max, (uint32_t)a.mx);
  uint32_t n = bfi(rstM, p.d_evcount, a.n);
  corr = andn(corr, rstM);

  const uint32_t arg = raw_lo;
  // count: += / -= arg (JVM Int wrap) or := arg
  count += ((arg ^ q0.y) - q0.y) & (q0.x & appM);
  const uint32_t msetM = q0.z & appM;
  count = bfi(msetM, arg, count);
  const uint32_t mverM = q0.w & appM;
  version = bfi(mverM, seq, version);
  // sum64 += / -= (long) arg.  -(long)x == (long)~x + 1 exactly (also for Int.MinValue), so add the
  // sign-extended complement now and count the "+1"s in corr (folded into the sum when the walk ends).
  {
    const uint32_t m = q1.x & appM;
    const uint32_t x = (arg ^ q1.y) & m;
    const uint64_t sum = (((uint64_t)sum_hi << 32) | sum_lo) + (uint64_t)(int64_t)(int32_t)x;
    sum_lo = (uint32_t)sum;
    sum_hi = (uint32_t)(sum >> 32);
    corr -= q1.y & m;
  }
  // balance := value (bit copy)
  const uint32_t mbalM = q1.z & appM;
  bal_lo = bfi(mbalM, raw_lo, bal_lo);
  bal_hi = bfi(mbalM, raw_hi, bal_hi);
  mn = (uint32_t)min((int32_t)mn, (int32_t)bfi(q3.x & appM, arg, 0x7fffffffu));
  mx = (uint32_t)max((int32_t)mx, (int32_t)bfi(q3.y & appM, arg, 0x80000000u));
  n += q1.w & appM;
  fl |= (msetM & SM_COUNT) | (mverM & SM_VERSION) | (mbalM & SM_BAL);

  a.count = (int32_t)count; a.version = (int32_t)version;
  a.sum = (int64_t)(((uint64_t)sum_hi << 32) | sum_lo);
  a.bal = ((uint64_t)bal_hi << 32) | bal_lo;
  a.mn = (int32_t)mn; a.mx = (int32_t)mx; a.n = n; a.fl = fl;
}

// The same case of handleEvent for a lane whose running state is CONCRETE (it walks an aggregate from its known prior
// state: the row kernels' whole aggregates): presence and "an event threw" live in two mask registers and none of the
// transformer's absolute / relative bookkeeping (the SM_* bits) is kept — 7 VALU instructions fewer per event.  It did
// not pay over the CSR log (round 2: the row-piece transport was the bound); over the tile-major log the walk's VALU
// time is what stands between the kernel and the transport ceiling.
__device__ __forceinline__ void apply_event_concrete(Acc& a, uint32_t& presentM, uint32_t& frozenM, uint32_t& corr, const uint4 q0,
                                                     const uint4 q1, const uint4 q2, const uint2 q3, uint32_t seq, uint32_t raw_lo,
                                                     uint32_t raw_hi, const FoldParams& p) {
  const uint32_t goM = ~(frozenM | q2.x);
  frozenM |= q2.x;                                                      // a throwing event freezes the aggregate
  const uint32_t delM = goM & q2.y;
  const uint32_t appM = andn(goM, q2.y) & (presentM | q2.z);            // REQUIRE-class events skip None
  const uint32_t rstM = appM & (q2.w | ~presentM);                      // CREATE, or materialising from None
  presentM = andn(presentM, delM) | rstM;
  // A build for one schema (kSpecV1) does not walk the fields no event type touches: such a field of a present aggregate is
  // its default when a reset happened since the state the walk started from, else still that state's value — one sticky
  // mask (kept in a.fl's place: the concrete walk does not use a.fl) says which, finish_concrete() applies it.
  if (kAnyDead) a.fl |= rstM;

  const uint32_t arg = raw_lo;
  if (kLiveCount) {
    uint32_t count = bfi(rstM, (uint32_t)p.d_count, (uint32_t)a.count);
    count += ((arg ^ q0.y) - q0.y) & (q0.x & appM);
    count = bfi(q0.z & appM, arg, count);
    a.count = (int32_t)count;
  }
  if (kLiveVersion) {
    uint32_t version = bfi(rstM, (uint32_t)p.d_version, (uint32_t)a.version);
    version = bfi(q0.w & appM, seq, version);
    a.version = (int32_t)version;
  }
  if (kLiveSum) {
    uint32_t sum_lo = bfi(rstM, (uint32_t)p.d_sum, (uint32_t)a.sum);
    uint32_t sum_hi = bfi(rstM, (uint32_t)((uint64_t)p.d_sum >> 32), (uint32_t)((uint64_t)a.sum >> 32));
    corr = andn(corr, rstM);
    const uint32_t m = q1.x & appM;
    const uint32_t x = (arg ^ q1.y) & m;
    const uint64_t sum = (((uint64_t)sum_hi << 32) | sum_lo) + (uint64_t)(int64_t)(int32_t)x;
    corr -= q1.y & m;
    a.sum = (int64_t)sum;
  }
  if (kLiveBal) {
    uint32_t bal_lo = bfi(rstM, (uint32_t)p.d_balance, (uint32_t)a.bal);
    uint32_t bal_hi = bfi(rstM, (uint32_t)(p.d_balance >> 32), (uint32_t)(a.bal >> 32));
    const uint32_t mbalM = q1.z & appM;
    bal_lo = bfi(mbalM, raw_lo, bal_lo);
    bal_hi = bfi(mbalM, raw_hi, bal_hi);
    a.bal = ((uint64_t)bal_hi << 32) | bal_lo;
  }
  if (kLiveMin) {
    const uint32_t mn = bfi(rstM, (uint32_t)p.d_min, (uint32_t)a.mn);
    a.mn = min((int32_t)mn, (int32_t)bfi(q3.x & appM, arg, 0x7fffffffu));
  }
  if (kLiveMax) {
    const uint32_t mx = bfi(rstM, (uint32_t)p.d_max, (uint32_t)a.mx);
    a.mx = max((int32_t)mx, (int32_t)bfi(q3.y & appM, arg, 0x80000000u));
  }
  if (kLiveN) {
    const uint32_t n = bfi(rstM, p.d_evcount, a.n);
    a.n = n + (q1.w & appM);
  }
}

// Before a concrete walk: a.fl becomes the "a reset happened" mask of the untouched fields (builds with such fields only).
__device__ __forceinline__ void begin_concrete(Acc& a) {
  if (kAnyDead) a.fl = 0u;
}
// After it: the untouched fields take their defaults where a reset happened (elsewhere they still hold the value the walk
// started from), a.fl is rebuilt from the two masks, the "+1"s of the subtractions go into the sum.
__device__ __forceinline__ void finish_concrete(Acc& a, uint32_t presentM, uint32_t frozenM, uint32_t corr, const FoldParams& p) {
  if (kAnyDead) {
    const uint32_t r = a.fl;
    if (!kLiveCount) a.count = (int32_t)bfi(r, (uint32_t)p.d_count, (uint32_t)a.count);
    if (!kLiveVersion) a.version = (int32_t)bfi(r, (uint32_t)p.d_version, (uint32_t)a.version);
    if (!kLiveSum) a.sum = (int64_t)(((uint64_t)bfi(r, (uint32_t)((uint64_t)p.d_sum >> 32), (uint32_t)((uint64_t)a.sum >> 32)) << 32) | bfi(r, (uint32_t)p.d_sum, (uint32_t)a.sum));
    if (!kLiveBal) a.bal = ((uint64_t)bfi(r, (uint32_t)(p.d_balance >> 32), (uint32_t)(a.bal >> 32)) << 32) | bfi(r, (uint32_t)p.d_balance, (uint32_t)a.bal);
    if (!kLiveMin) a.mn = (int32_t)bfi(r, (uint32_t)p.d_min, (uint32_t)a.mn);
    if (!kLiveMax) a.mx = (int32_t)bfi(r, (uint32_t)p.d_max, (uint32_t)a.mx);
    if (!kLiveN) a.n = bfi(r, p.d_evcount, a.n);
  }
  a.fl = (presentM & FL_PRESENT) | (frozenM & FL_POISONED);
  if (kLiveSum) a.sum = (int64_t)((uint64_t)a.sum + corr);
}

// g after f.  Absolute fields of g win, relative ones combine with f's.  A poisoned f is only ever
// followed (inside its segment) by lanes that ignored their events, i.e. by identity transformers,
// so the fields need no special case; the presence bit then has to come from f.
__device__ __forceinline__ Acc seq_acc(const Acc& f, const Acc& g) {
  Acc r;
  const bool all = (g.fl & SM_ALL) == SM_ALL;  // sum/min/max/n become absolute only through a reset
  r.count = (g.fl & SM_COUNT) ? g.count : (int32_t)((uint32_t)f.count + (uint32_t)g.count);
  r.version = (g.fl & SM_VERSION) ? g.version : f.version;
  r.sum = all ? g.sum : (int64_t)((uint64_t)f.sum + (uint64_t)g.sum);
  r.bal = (g.fl & SM_BAL) ? g.bal : f.bal;
  r.mn = all ? g.mn : min(f.mn, g.mn);
  r.mx = all ? g.mx : max(f.mx, g.mx);
  r.n = all ? g.n : f.n + g.n;
  const uint32_t present = (f.fl & FL_POISONED) ? (f.fl & FL_PRESENT) : (g.fl & FL_PRESENT);
  r.fl = present | ((f.fl | g.fl) & (FL_POISONED | SM_ALL)) | (f.fl & FL_HEAD);
  if (kAnyDead) r.fl |= (all ? g.fl : f.fl) & FL_DFLT;  // untouched fields become absolute only through a reset / delete / head
  return r;
}

__device__ __forceinline__ Acc select_acc(bool c, const Acc& a, const Acc& b) {
  Acc r;
  r.count = c ? a.count : b.count; r.version = c ? a.version : b.version;
  r.sum = c ? a.sum : b.sum; r.bal = c ? a.bal : b.bal;
  r.mn = c ? a.mn : b.mn; r.mx = c ? a.mx : b.mx; r.n = c ? a.n : b.n; r.fl = c ? a.fl : b.fl;
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

// a value every lane holds identically, moved to scalar registers (buffer descriptors need wave-uniform bases)
__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

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

// store_state for a build with untouched fields: those are not in `a` (whatever it holds for them is dead code) — the
// default when a reset happened since the prior state (FL_DFLT), else the prior state's own bytes.  Without a prior
// snapshot every segment starts from None, whose only way to Some is a reset: FL_DFLT is always set on a present state.
__device__ __forceinline__ void store_state_flat(const FoldParams& p, int64_t idx, const Acc& a) {
  if (!kAnyDead) {
    store_state(p.out, idx, a);
    return;
  }
  const bool pr = (a.fl & FL_PRESENT) != 0;
  const bool dflt = (a.fl & FL_DFLT) != 0 || p.init == nullptr;
  uint4 q0 = {0u, 0u, 0u, 0u}, q1 = {0u, 0u, 0u, 0u}, q2 = {0u, 0u, 0u, 0u};
  if (pr && !dflt) {
    const uint4* s = p.init + idx * 4;
    q0 = s[0]; q1 = s[1]; q2 = s[2];
  }
  const uint32_t count = kLiveCount ? (uint32_t)a.count : (dflt ? (uint32_t)p.d_count : q0.x);
  const uint32_t version = kLiveVersion ? (uint32_t)a.version : (dflt ? (uint32_t)p.d_version : q0.y);
  const uint32_t sum_lo = kLiveSum ? (uint32_t)a.sum : (dflt ? (uint32_t)p.d_sum : q0.z);
  const uint32_t sum_hi = kLiveSum ? (uint32_t)((uint64_t)a.sum >> 32) : (dflt ? (uint32_t)((uint64_t)p.d_sum >> 32) : q0.w);
  const uint32_t bal_lo = kLiveBal ? (uint32_t)a.bal : (dflt ? (uint32_t)p.d_balance : q1.x);
  const uint32_t bal_hi = kLiveBal ? (uint32_t)(a.bal >> 32) : (dflt ? (uint32_t)(p.d_balance >> 32) : q1.y);
  const uint32_t mn = kLiveMin ? (uint32_t)a.mn : (dflt ? (uint32_t)p.d_min : q1.z);
  const uint32_t mx = kLiveMax ? (uint32_t)a.mx : (dflt ? (uint32_t)p.d_max : q1.w);
  const uint32_t n = kLiveN ? a.n : (dflt ? p.d_evcount : q2.x);
  uint4 v0, v1, v2, v3;
  v0.x = pr ? count : 0u; v0.y = pr ? version : 0u; v0.z = pr ? sum_lo : 0u; v0.w = pr ? sum_hi : 0u;
  v1.x = pr ? bal_lo : 0u; v1.y = pr ? bal_hi : 0u; v1.z = pr ? mn : 0u; v1.w = pr ? mx : 0u;
  v2.x = pr ? n : 0u; v2.y = a.fl & (FL_PRESENT | FL_POISONED); v2.z = 0u; v2.w = 0u;
  v3.x = v3.y = v3.z = v3.w = 0u;
  uint4* o = p.out + idx * 4;
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

// ---- tile geometry ---------------------------------------------------------------------------------
// One wave = 64 lanes x LE consecutive events per tile (LE = 16: 16 KiB tiles, LE = 8: 8 KiB tiles and
// twice the resident waves).  A tile is fetched by direct global->LDS loads; instruction q writes LDS
// bytes [q*1024, q*1024+1024) linearly by lane (that is what the hardware does); WHICH event a lane
// fetches is ours to choose: LDS slot (q*64 + m) belongs to chunk-lane l = (64/LE) q + m / LE and holds
// its event j = (m % LE) ^ key(l), an XOR swizzle inside the lane's own LE*16-byte row that makes the
// later ds_read_b128 of "event j of lane l" bank-conflict free.  Every instruction still covers one
// contiguous, fully used 1 KiB of the log.  The lane offset inside a 1 KiB piece only depends on
// q mod kClasses, so it is computed once per wave.
template <int LE>
struct Geo {
  static constexpr int kTile = kWave * LE;
  static constexpr int kTileBytes = kTile * 16;
  static constexpr int kRowBytes = LE * 16;
  static constexpr int kLoads = kTileBytes / 1024;
  static constexpr int kRowsPerLoad = kWave / LE;
  static constexpr int kHeadWords = kTile / 32;
  // LDS layout of every fold kernel: [tile][aux][op table]; the aux area is the head bitmask of the flat
  // kernels, 64 row starts + lengths of the sorted kernel, nothing for the uniform rows kernel
  static constexpr int kAuxFlat = kHeadWords * 4;
  static constexpr int kAuxSorted = kWave * 12;
  static constexpr int kAuxRows = 0;
  static constexpr int kClasses = LE == 32 ? 8 : (LE == 16 ? 4 : 2);
  static constexpr int lds_bytes(int aux) { return kTileBytes + aux + kTableLdsDwords * 4; }
  static constexpr uint32_t kLaneMask = LE >= 32 ? 0xffffffffu : ((1u << (LE & 31)) - 1u);
  __device__ static __forceinline__ uint32_t key(int l) { return LE >= 16 ? (uint32_t)(l & 15) : (uint32_t)((l >> 1) & 7); }
  // my pre-swizzled LDS row: event j lives at (row ^ (j * 16))
  __device__ static __forceinline__ uint32_t ev_row(int lane) { return (uint32_t)lane * kRowBytes + key(lane) * 16u; }
  // event index j that load-lane m of an instruction of class k fetches, and its chunk-lane within the instruction
  __device__ static __forceinline__ uint32_t load_j(int m, int k) {
    const int l = kRowsPerLoad * k + m / LE;  // only key(l) matters and it is periodic in q with period kClasses
    return (uint32_t)(m % LE) ^ key(l);
  }
};

// Op-table entries are addressed by BYTE offset into the LDS copy of the table (tyc[] below): an event's entry is
// min(type, 16) * 80 bytes in — one v_min + one v_mul per event, no shift in front of the ds_read; the tile-major re-layout
// stores that offset in place of the type word, so its fold spends nothing on it.
constexpr uint32_t kTableStrideBytes = kTableStride * 4u;
constexpr uint32_t kNullEntryOffBytes = kSpecV1 ? 17u : kNullEntryOff * 4u;  // the null event's reference: its entry index in a specialised build
template <int K>
__device__ __forceinline__ uint32_t spec_word(uint32_t entry) {
  constexpr uint32_t m = SURGE_V1_MASK(K);
  if (m == 0u) return 0u;
  if (K == TW_EVC) return (m >> entry) & 1u;
  return (uint32_t)__builtin_amdgcn_sbfe((int32_t)m, entry, 1u);
}
__device__ __forceinline__ uint32_t type_off(uint32_t ty) { return (ty < 16u ? ty : 16u) * (kSpecV1 ? 1u : kTableStrideBytes); }
__device__ __forceinline__ const uint4* table_entry(const uint32_t* lds_tab, uint32_t off) {
  return (const uint4*)((const char*)lds_tab + off);
}
__device__ __forceinline__ uint32_t table_word(const uint32_t* lds_tab, uint32_t off, int word) {
  return *(const uint32_t*)((const char*)lds_tab + off + word * 4);
}
// the two words of the flat kernel's presence pre-pass
__device__ __forceinline__ uint32_t flags_word(const uint32_t* lds_tab, uint32_t ref) {
  if (kSpecV1) return ((SURGE_V1_THROWS >> ref) & 1u) | (((SURGE_V1_DELETES >> ref) & 1u) << 16);
  return table_word(lds_tab, ref, TW_FLAGS);
}
__device__ __forceinline__ uint32_t materializes_word(const uint32_t* lds_tab, uint32_t ref) {
  if (kSpecV1) return spec_word<TW_MATERIALIZES>(ref);
  return table_word(lds_tab, ref, TW_MATERIALIZES);
}

template <int LE>
__device__ __forceinline__ void load_table(const FoldParams& p, uint32_t* lds_tab, int lane) {
  if (kSpecV1) return;
  const uint32_t* src = &p.table[0][0];
  for (int i = lane; i < kTableEntries * kTableWords; i += kWave)
    lds_tab[(i / kTableWords == kTableEntries - 1 ? kNullEntryOff : (i / kTableWords) * kTableStride) + (i % kTableWords)] = src[i];
}

// One linear tile [te0, te0 + kTile) of the events buffer -> LDS, as buffer_load_dwordx4 ... lds: descriptor base = the
// tile's first byte (wave-uniform), lane offset = the swizzled slot, instruction q at scalar offset 1024 q.  MUBUF rather
// than global_load_lds on purpose: hipcc models global_load_lds as a FLAT access that may touch LDS, and while one is
// outstanding (the next tile's fetch is, during the whole walk) it turns EVERY s_waitcnt in front of an LDS read into
// lgkmcnt(0) — the one-event-ahead prefetch of op-table entries then waits for the entry it has just requested.  With
// buffer loads the waits are the exact counts.  (The row kernels over the CSR log cannot do this: their lanes address
// rows anywhere in a log of up to 2^38 bytes and a buffer offset has 32 bits; the tile-major kernel can.)
template <int LE>
__device__ __forceinline__ void issue_tile_loads(const FoldParams& p, int64_t te0, int64_t end, char* lds, const uint32_t* voff) {
  using G = Geo<LE>;
  const char* base = (const char*)(p.events + te0);  // wave-uniform
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  if (te0 + G::kTile <= end) {
#pragma unroll
    for (int q = 0; q < G::kLoads; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds + q * 1024), 16, (int)voff[q % G::kClasses], q * 1024, 0, kLoadAux);
  } else {
    // The task's last tile.  Only the 1 KiB pieces that hold events of the task are fetched: a task is a few tiles long on
    // a small log (3 on a 0.1 M-aggregate one) and fetching its last tile whole — on average half a tile of the NEXT
    // task's events — is what the counters showed as 1.17 x the algorithmic bytes in rounds 2 and 3.  The LDS slots of
    // the pieces left out keep whatever they held: those events are past the task's end and the walk replaces them
    // with the null event.  `end` never exceeds p.n_events, and the clamp keeps the last piece inside the buffer.
    const int need = (int)((end - te0) * 16);  // wave-uniform, in (0, kTileBytes)
    const int last = need - 16;
#pragma unroll
    for (int q = 0; q < G::kLoads; ++q) {
      if (q * 1024 < need) {
        int off = q * 1024 + (int)voff[q % G::kClasses];
        off = off < last ? off : last;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(lds + q * 1024), 16, off, 0, 0, kLoadAux);
      }
    }
  }
}

// The walk over my LE events: one evaluation path, op-table entries prefetched one event ahead.
// on_head(j) is called before event j when it starts a new segment (flat kernels only).
template <int LE, bool HEADS, typename OnHead>
__device__ __forceinline__ void walk_events(Acc& a, uint32_t& frozenM, uint32_t& corr, const uint4* ev, const uint32_t* tyc,
                                            uint32_t hb, const uint32_t* lds_tab, const FoldParams& p, OnHead on_head) {
  if (kSpecV1) {
#pragma unroll
    for (int j = 0; j < LE; ++j) {
      const uint32_t e = tyc[j];
      const uint4 q0 = {spec_word<0>(e), spec_word<1>(e), spec_word<2>(e), spec_word<3>(e)};
      const uint4 q1 = {spec_word<4>(e), spec_word<5>(e), spec_word<6>(e), spec_word<7>(e)};
      const uint4 q2 = {spec_word<8>(e), spec_word<9>(e), spec_word<10>(e), spec_word<11>(e)};
      const uint2 q3 = {spec_word<12>(e), spec_word<13>(e)};
      if (HEADS && ((hb >> j) & 1u)) on_head(j);
      apply_event(a, frozenM, corr, q0, q1, q2, q3, ev[j].y, ev[j].z, ev[j].w, p);
    }
    return;
  }
  uint4 tq0, tq1, tq2;
  uint2 tq3;
  {
    const uint4* te = table_entry(lds_tab, tyc[0]);
    tq0 = te[0]; tq1 = te[1]; tq2 = te[2]; tq3 = *(const uint2*)(te + 3);
  }
#pragma unroll
  for (int j = 0; j < LE; ++j) {
    uint4 nq0 = tq0, nq1 = tq1, nq2 = tq2;
    uint2 nq3 = tq3;
#if !defined(SURGE_DBG_FIXED_TABLE)
    if (j + 1 < LE) {
      const uint4* te = table_entry(lds_tab, tyc[j + 1]);
      nq0 = te[0]; nq1 = te[1]; nq2 = te[2]; nq3 = *(const uint2*)(te + 3);
    }
#endif
    if (HEADS && ((hb >> j) & 1u)) on_head(j);
#if defined(SURGE_DBG_SKIP_APPLY)  // experiment builds only: keep the loads alive, skip the arithmetic
    a.count ^= (int32_t)(tq0.x ^ tq1.x ^ tq2.x ^ tq3.x ^ ev[j].y ^ ev[j].z ^ ev[j].w);
#else
    apply_event(a, frozenM, corr, tq0, tq1, tq2, tq3, ev[j].y, ev[j].z, ev[j].w, p);
#endif
    tq0 = nq0; tq1 = nq1; tq2 = nq2; tq3 = nq3;
    __builtin_amdgcn_sched_barrier(0);  // keep the table prefetch one event deep (bounds VGPR pressure)
  }
}


// The concrete-state walk (apply_event_concrete) between begin_concrete() and finish_concrete().
template <int LE>
__device__ __forceinline__ void walk_events_concrete(Acc& a, uint32_t& presentM, uint32_t& frozenM, uint32_t& corr, const uint4* ev,
                                                     const uint32_t* tyc, const uint32_t* lds_tab, const FoldParams& p) {
  if (kSpecV1) {  // table words are bit tests of compile-time masks at the event's entry index: nothing is read from LDS
#pragma unroll
    for (int j = 0; j < LE; ++j) {
      const uint32_t e = tyc[j];
      const uint4 q0 = {spec_word<0>(e), spec_word<1>(e), spec_word<2>(e), spec_word<3>(e)};
      const uint4 q1 = {spec_word<4>(e), spec_word<5>(e), spec_word<6>(e), spec_word<7>(e)};
      const uint4 q2 = {spec_word<8>(e), spec_word<9>(e), spec_word<10>(e), spec_word<11>(e)};
      const uint2 q3 = {spec_word<12>(e), spec_word<13>(e)};
#ifdef SURGE_EXP_SKIP_APPLY  // (experiment builds: the transport without the arithmetic)
      a.count ^= (int32_t)(e ^ ev[j].y ^ ev[j].z ^ ev[j].w);
#else
      apply_event_concrete(a, presentM, frozenM, corr, q0, q1, q2, q3, ev[j].y, ev[j].z, ev[j].w, p);
#endif
    }
    return;
  }
  uint4 tq0, tq1, tq2;
  uint2 tq3;
  {
    const uint4* te = table_entry(lds_tab, tyc[0]);
    tq0 = te[0]; tq1 = te[1]; tq2 = te[2]; tq3 = *(const uint2*)(te + 3);
  }
#pragma unroll
  for (int j = 0; j < LE; ++j) {
    uint4 nq0 = tq0, nq1 = tq1, nq2 = tq2;
    uint2 nq3 = tq3;
    if (j + 1 < LE) {
      const uint4* te = table_entry(lds_tab, tyc[j + 1]);
      nq0 = te[0]; nq1 = te[1]; nq2 = te[2]; nq3 = *(const uint2*)(te + 3);
    }
#ifdef SURGE_EXP_SKIP_APPLY_AOT  // (experiment builds: the transport and the table reads without the arithmetic)
    a.count ^= (int32_t)(tq0.x ^ tq1.x ^ tq2.x ^ tq3.x ^ ev[j].y ^ ev[j].z ^ ev[j].w);
#else
    apply_event_concrete(a, presentM, frozenM, corr, tq0, tq1, tq2, tq3, ev[j].y, ev[j].z, ev[j].w, p);
#endif
    tq0 = nq0; tq1 = nq1; tq2 = nq2; tq3 = nq3;
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace
// Persistent kernels pull groups of 64 rows from an atomic ticket counter.  Every wave draws until its first ticket
// beyond the last group and then checks out; the last wave to check out re-arms the dispenser for the next launch (every
// other wave's final draw returned before that wave checked out, so nothing can still be drawing).
__device__ __forceinline__ void dispenser_leave(unsigned long long* counter, int lane) {
  if (lane == 0) {
    const unsigned long long left = atomicAdd(counter + 1, 1ull);
    if (left == (unsigned long long)gridDim.x - 1ull) {
      atomicExch(counter + 1, 0ull);
      atomicExch(counter, 0ull);
    }
  }
}

}  // namespace surge
