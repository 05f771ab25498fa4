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

#include "fold_lane_device.h"

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

hipError_t launch_fold_sorted_pf(const FoldParams& p, const V1Kernels* lanes, int64_t n_waves, int lane_events, hipStream_t stream) {
  if (n_waves <= 0) return hipSuccess;
  if (lanes) {  // compiled for the op table: no table in LDS — the tile and the row table only
    const char* padv = std::getenv("SURGE_REPLAY_RTC_LDS_PAD");  // (experiments: extra dynamic LDS = fewer resident waves per CU)
    const unsigned pad = padv ? (unsigned)std::atoi(padv) : 0u;
    if (lane_events == 32) return launch_v1_lane(lanes->sorted32, &p, sizeof(p), n_waves, Geo<32>::kTileBytes + Geo<32>::kAuxSorted + pad, stream);
    return launch_v1_lane(lane_events == 8 ? lanes->sorted8 : lanes->sorted16, &p, sizeof(p), n_waves,
                          (lane_events == 8 ? Geo<8>::kTileBytes + Geo<8>::kAuxSorted : Geo<16>::kTileBytes + Geo<16>::kAuxSorted) + pad, stream);
  }
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
                               const int64_t* r_out, int64_t n_cut, const V1Kernels* lanes, int64_t n_waves, int lane_events, hipStream_t stream) {
  if (n_waves <= 0 || n_vrows <= 0) return hipSuccess;
  ChunkTable t;
  t.v_start = v_start; t.v_len = v_len; t.v_info = v_info; t.v_dest = v_dest; t.n_vrows = n_vrows; t.side = side;
  const bool conc = concrete_walk();
  if (lanes) {
    ChunkArgs a;
    a.p = p;
    a.t = t;
    const hipError_t e = launch_v1_lane(lane_events == 8 ? lanes->chunked8 : lanes->chunked16, &a, sizeof(a), n_waves,
                                        lane_events == 8 ? Geo<8>::kTileBytes + Geo<8>::kAuxSorted : Geo<16>::kTileBytes + Geo<16>::kAuxSorted, stream);
    if (e != hipSuccess) return e;
  } else if (lane_events == 8) {
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
