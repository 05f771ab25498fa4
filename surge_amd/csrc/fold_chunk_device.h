// fold_chunk_device.h — device code shared by the two kernels that walk CHUNKS of aggregates (fold_chunked.hip over the
// CSR log, fold_tiled.hip over the tile-major log): the virtual-row flags, the walk that watches for a chunk's
// "deciding" event, the resolution of a chunk summary against a concrete incoming state and the 80-byte side entries.
// See the file comment of fold_chunked.hip for the P / S presence split these implement.
#pragma once
#include "fold_device.h"

namespace surge {
namespace {

constexpr uint32_t VI_RELATIVE = 1u;        // bit 0: chunk of a cut aggregate (walked relative to an unknown incoming state)
constexpr int VI_PAD_SHIFT = 16;            // bits 16..18: null events in front of an aggregate's first event (line alignment)
constexpr uint32_t VI_SIDE = 1u << 25;      // the chunk's summary goes to the side buffer (slot = v_dest), not to the state array
constexpr uint32_t SIDE_DECIDED = 1u << 31; // in a side entry's S.fl: the chunk contained a deciding event
constexpr int kSideDwords = 20;             // a side entry: P (10 dwords) then S (10 dwords)

// walk of LE events that also watches for the lane's first "deciding" event (see the file comment); lanes with
// undecM == 0 (an aggregate in one piece, or already decided) just walk
template <int LE>
__device__ __forceinline__ void walk_events_track(Acc& a, Acc& P, uint32_t& undecM, uint32_t& frozenM, uint32_t& corr,
                                                  const uint4* ev, const uint32_t* tyc, const uint32_t* lds_tab,
                                                  const FoldParams& p) {
  auto decide = [&](uint32_t firstM) {  // wave-uniform branch around it; taken once or twice per chunk
    a.sum = (int64_t)((uint64_t)a.sum + corr);
    corr = 0u;
    const bool f = firstM != 0u;
    P = select_acc(f, a, P);
    a = select_acc(f, acc_identity(), a);
    undecM = andn(undecM, firstM);
  };
  if (kSpecV1) {
#pragma unroll
    for (int j = 0; j < LE; ++j) {
      const uint32_t e = tyc[j];
      const uint4 q0 = {spec_word<0>(e), spec_word<1>(e), spec_word<2>(e), spec_word<3>(e)};
      const uint4 q1 = {spec_word<4>(e), spec_word<5>(e), spec_word<6>(e), spec_word<7>(e)};
      const uint4 q2 = {spec_word<8>(e), spec_word<9>(e), spec_word<10>(e), spec_word<11>(e)};
      const uint2 q3 = {spec_word<12>(e), spec_word<13>(e)};
      const uint32_t firstM = undecM & q2.z & ~(frozenM | q2.x);
      if (__builtin_amdgcn_ballot_w64(firstM != 0u) != 0ull) decide(firstM);
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
    if (j + 1 < LE) {
      const uint4* te = table_entry(lds_tab, tyc[j + 1]);
      nq0 = te[0]; nq1 = te[1]; nq2 = te[2]; nq3 = *(const uint2*)(te + 3);
    }
    // live (not ignored, not throwing) and not of class REQUIRE: from here on the state is Some or an absolute None
    const uint32_t firstM = undecM & tq2.z & ~(frozenM | tq2.x);
    if (__builtin_amdgcn_ballot_w64(firstM != 0u) != 0ull) decide(firstM);
    apply_event(a, frozenM, corr, tq0, tq1, tq2, tq3, ev[j].y, ev[j].z, ev[j].w, p);
    tq0 = nq0; tq1 = nq1; tq2 = nq2; tq3 = nq3;
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ Acc acc_defaults(const FoldParams& p) {  // Some(defaults), absolute
  Acc a;
  a.count = p.d_count; a.version = p.d_version; a.sum = p.d_sum; a.bal = p.d_balance; a.mn = p.d_min; a.mx = p.d_max;
  a.n = p.d_evcount;
  a.fl = FL_PRESENT | SM_ALL;
  return a;
}

// the state after a chunk whose incoming state is x (concrete): see the file comment
__device__ __forceinline__ Acc resolve_chunk(const Acc& x, const Acc& P, const Acc& S, bool decided, const FoldParams& p) {
  const Acc some = seq_acc(seq_acc(x, P), S);
  const Acc none_dec = seq_acc(acc_defaults(p), S);
  Acc none_und = x;
  none_und.fl |= P.fl & FL_POISONED;
  const bool xpres = (x.fl & FL_PRESENT) != 0u, xpois = (x.fl & FL_POISONED) != 0u;
  Acc r = select_acc(xpres, some, select_acc(decided, none_dec, none_und));
  return select_acc(xpois, x, r);  // an aggregate that threw ignores every later event
}

__device__ __forceinline__ void acc_to_words(const Acc& a, uint32_t* w) {
  w[0] = (uint32_t)a.count; w[1] = (uint32_t)a.version; w[2] = (uint32_t)a.sum; w[3] = (uint32_t)((uint64_t)a.sum >> 32);
  w[4] = (uint32_t)a.bal; w[5] = (uint32_t)(a.bal >> 32); w[6] = (uint32_t)a.mn; w[7] = (uint32_t)a.mx; w[8] = a.n; w[9] = a.fl;
}
__device__ __forceinline__ Acc acc_from_words(const uint32_t* w) {
  Acc a;
  a.count = (int32_t)w[0]; a.version = (int32_t)w[1]; a.sum = (int64_t)(((uint64_t)w[3] << 32) | w[2]);
  a.bal = ((uint64_t)w[5] << 32) | w[4]; a.mn = (int32_t)w[6]; a.mx = (int32_t)w[7]; a.n = w[8]; a.fl = w[9];
  return a;
}
__device__ __forceinline__ void store_side(uint32_t* side, int64_t slot, const Acc& P, const Acc& S) {
  uint32_t w[kSideDwords];
  acc_to_words(P, w);
  acc_to_words(S, w + 10);
  uint4* o = (uint4*)(side + slot * kSideDwords);  // 80-byte entries: 16-byte aligned
#pragma unroll
  for (int i = 0; i < 5; ++i) o[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

struct ChunkTable {
  const int64_t* v_start;  // first event slot of the virtual row (a multiple of 8 events: line aligned)
  const uint32_t* v_len;   // event slots from there (pad included)
  const uint32_t* v_info;  // VI_*
  const int64_t* v_dest;   // aggregate index (state array) or side-buffer slot
  int64_t n_vrows;
  uint32_t* side;
};
struct ChunkArgs {  // the argument block of a run-time compiled chunked kernel (one by-value struct: hipModuleLaunchKernel's buffer form)
  FoldParams p;
  ChunkTable t;
};

}  // namespace
}  // namespace surge
