// fold_slots.hip — the fold of ABI v2 "slot" schemas (include/surge_replay.h): the kernels' device code lives in
// fold_slots_device.h (read its header comment first); this file instantiates the GENERIC INTERPRETER ahead of time,
// and builds, caches and launches the SCHEMA-SPECIALISED kernels that hiprtc compiles from the same source when a v2
// handle is created (rtc.cpp).
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "fold_slots_device.h"

namespace surge {

static_assert(sizeof(SlotParams) <= kSlotParamsBytes, "grow kSlotParamsBytes");

namespace {

template <int LE>
__global__ void __launch_bounds__(kWave) fold_slots_kernel(const FoldParams p, const SlotParams sp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const SlotSchema sc{sp};
  fold_slots_csr_body<LE>(p, sc, smem);
}

template <int SUBS>
__global__ void __launch_bounds__(kWave) fold_slots_tiled_kernel(const FoldParams p, const TileTable t, const SlotParams sp) {
  __shared__ __attribute__((aligned(16))) char lds_ev[SUBS * kSubBytes];
  __shared__ __attribute__((aligned(16))) uint32_t lds_tab[kTableLdsDwords];
  const SlotSchema sc{sp};
  fold_slots_tiled_body<SUBS>(p, t, sc, lds_ev, lds_tab);
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
    const bool applies = !(p.cls[t] & SURGE_D_POISON) && (p.cls[t] & SURGE_CLS_MASK) != SURGE_CLS_DELETE;
    for (uint32_t i = 0; i < sc.n_slots; ++i)
      if (applies) p.used_ops[i] |= 1u << ((sc.ops[t] >> (4 * i)) & 15u);
  }
  p.cls[SURGE_MAX_EVENT_TYPES + 1] = CLS_NULL;  // [17]: the null event
  *out = p;
}

// ---- the schema-specialised build ---------------------------------------------------------------------------------
struct SlotKernels {
  hipModule_t module = nullptr;
  hipFunction_t csr8 = nullptr, csr16 = nullptr, tiled1 = nullptr, tiled2 = nullptr;
  int device = 0;
  double compile_ms = 0.0;
};

// The program handed to hiprtc: the ABI constants the device code names, the schema as SURGE_SPEC_* macros (ternary
// chains over an index that is a constant after unrolling), the shared device header, four extern "C" entry points.
std::string slots_spec_source(const SlotParams& sp) {
  std::string s;
  char b[256];
  auto def = [&](const char* name, unsigned long long v, const char* suffix) {
    std::snprintf(b, sizeof b, "#define %s %llu%s\n", name, v, suffix);
    s += b;
  };
#define SURGE_EMIT(name) def(#name, (unsigned long long)(name), "u")
  SURGE_EMIT(SURGE_MAX_SLOTS);
  SURGE_EMIT(SURGE_MAX_EVENT_TYPES);
  SURGE_EMIT(SURGE_CLS_MATERIALIZE); SURGE_EMIT(SURGE_CLS_REQUIRE); SURGE_EMIT(SURGE_CLS_CREATE); SURGE_EMIT(SURGE_CLS_DELETE);
  SURGE_EMIT(SURGE_CLS_MASK); SURGE_EMIT(SURGE_D_POISON);
  SURGE_EMIT(SURGE_SLOT_I32); SURGE_EMIT(SURGE_SLOT_I64); SURGE_EMIT(SURGE_SLOT_F64);
  SURGE_EMIT(SURGE_SRC_ARG); SURGE_EMIT(SURGE_SRC_SEQ); SURGE_EMIT(SURGE_SRC_PAYLOAD); SURGE_EMIT(SURGE_SRC_ONE);
  SURGE_EMIT(SURGE_OP_KEEP); SURGE_EMIT(SURGE_OP_ADD); SURGE_EMIT(SURGE_OP_SUB); SURGE_EMIT(SURGE_OP_SET); SURGE_EMIT(SURGE_OP_MIN);
  SURGE_EMIT(SURGE_OP_MAX);
#undef SURGE_EMIT
  // MAX_SLOTS / MAX_EVENT_TYPES size arrays: they must be plain ints
  s += "#undef SURGE_MAX_SLOTS\n#undef SURGE_MAX_EVENT_TYPES\n";
  def("SURGE_MAX_SLOTS", SURGE_MAX_SLOTS, "");
  def("SURGE_MAX_EVENT_TYPES", SURGE_MAX_EVENT_TYPES, "");
  s += "#define SURGE_SLOTS_SPEC 1\n";
  def("SURGE_SPEC_N_SLOTS", sp.n_slots, "u");
  def("SURGE_SPEC_COUNT_EVENTS", sp.count_events, "u");
  auto chain = [&](const char* name, int n, auto value, const char* suffix) {
    s += std::string("#define ") + name + "(i) (";
    for (int i = 0; i < n; ++i) {
      std::snprintf(b, sizeof b, "(i) == %d ? %llu%s : ", i, (unsigned long long)value(i), suffix);
      s += b;
    }
    s += std::string("0") + suffix + ")\n";
  };
  chain("SURGE_SPEC_TYPE", SURGE_MAX_SLOTS, [&](int i) { return sp.type[i]; }, "u");
  chain("SURGE_SPEC_SOURCE", SURGE_MAX_SLOTS, [&](int i) { return sp.source[i]; }, "u");
  chain("SURGE_SPEC_USED", SURGE_MAX_SLOTS, [&](int i) { return sp.used_ops[i]; }, "u");
  chain("SURGE_SPEC_DEF", SURGE_MAX_SLOTS, [&](int i) { return sp.def[i]; }, "ul");
  chain("SURGE_SPEC_CLS", SURGE_MAX_EVENT_TYPES + 2, [&](int t) { return sp.cls[t]; }, "u");
  chain("SURGE_SPEC_OPS", SURGE_MAX_EVENT_TYPES + 2, [&](int t) { return sp.ops[t]; }, "u");
  s += R"SRC(
#include "fold_slots_device.h"
using namespace surge;
extern "C" __global__ void __launch_bounds__(64) surge_slots_csr8(const FoldParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fold_slots_csr_body<8>(p, SlotSchema{}, smem);
}
extern "C" __global__ void __launch_bounds__(64) surge_slots_csr16(const FoldParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fold_slots_csr_body<16>(p, SlotSchema{}, smem);
}
extern "C" __global__ void __launch_bounds__(64) surge_slots_tiled1(const FoldParams p, const TileTable t) {
  __shared__ __attribute__((aligned(16))) char lds_ev[kSubBytes];
  __shared__ __attribute__((aligned(16))) uint32_t lds_tab[kTableLdsDwords];
  fold_slots_tiled_body<1>(p, t, SlotSchema{}, lds_ev, lds_tab);
}
extern "C" __global__ void __launch_bounds__(64) surge_slots_tiled2(const FoldParams p, const TileTable t) {
  __shared__ __attribute__((aligned(16))) char lds_ev[2 * kSubBytes];
  __shared__ __attribute__((aligned(16))) uint32_t lds_tab[kTableLdsDwords];
  fold_slots_tiled_body<2>(p, t, SlotSchema{}, lds_ev, lds_tab);
}
)SRC";
  return s;
}

namespace {

std::mutex g_spec_mu;
// key: device + the SlotParams bytes (everything the generated source depends on); entries live as long as the process
// (a handful of schemas per host; a module is a few tens of KB)
std::map<std::string, std::unique_ptr<SlotKernels>> g_spec_cache;
std::map<std::string, std::string> g_spec_failed;  // key -> why (never retried: the answer will not change)

}  // namespace

void slot_kernels_acquire(const surge_replay_schema_v2& sc, const SlotParams& sp, int device, SlotKernels** out, double* compile_ms,
                          std::string* why) {
  (void)sc;
  *out = nullptr;
  *compile_ms = 0.0;
  if (const char* v = std::getenv("SURGE_REPLAY_RTC")) {
    if (std::atoi(v) == 0) {
      *why = "disabled by SURGE_REPLAY_RTC=0";
      return;
    }
  }
  std::string key((const char*)&sp, sizeof(sp));
  key += "@" + std::to_string(device);
  std::lock_guard<std::mutex> lk(g_spec_mu);
  auto hit = g_spec_cache.find(key);
  if (hit != g_spec_cache.end()) {
    *out = hit->second.get();
    *compile_ms = hit->second->compile_ms;
    return;
  }
  auto miss = g_spec_failed.find(key);
  if (miss != g_spec_failed.end()) {
    *why = miss->second;
    return;
  }
  auto give_up = [&](const std::string& m) {
    g_spec_failed[key] = m;
    *why = m;
  };
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return give_up("hipGetDeviceProperties failed");
  std::string arch = prop.gcnArchName;  // "gfx950:sramecc+:xnack-"
  const size_t colon = arch.find(':');
  if (colon != std::string::npos) arch.resize(colon);
  std::vector<char> code;
  std::string log;
  double ms = 0.0;
  if (!rtc_compile(slots_spec_source(sp), arch.c_str(), &code, &log, &ms)) return give_up(log);
  auto k = std::make_unique<SlotKernels>();
  k->device = device;
  k->compile_ms = ms;
  hipError_t e = hipModuleLoadData(&k->module, code.data());
  if (e != hipSuccess) return give_up(std::string("hipModuleLoadData: ") + hipGetErrorString(e));
  struct { hipFunction_t* f; const char* name; int lds; } fns[] = {
      {&k->csr8, "surge_slots_csr8", Geo<8>::lds_bytes(Geo<8>::kAuxSorted)},
      {&k->csr16, "surge_slots_csr16", Geo<16>::lds_bytes(Geo<16>::kAuxSorted)},
      {&k->tiled1, "surge_slots_tiled1", 0},
      {&k->tiled2, "surge_slots_tiled2", 0}};
  for (auto& f : fns) {
    e = hipModuleGetFunction(f.f, k->module, f.name);
    if (e != hipSuccess) {
      (void)hipModuleUnload(k->module);
      return give_up(std::string("hipModuleGetFunction(") + f.name + "): " + hipGetErrorString(e));
    }
  }
  *compile_ms = ms;
  *out = k.get();
  g_spec_cache[key] = std::move(k);
}

namespace {

hipError_t launch_module_kernel(hipFunction_t f, unsigned grid, unsigned lds, hipStream_t stream, void* args, size_t args_bytes) {
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &args_bytes, HIP_LAUNCH_PARAM_END};
  return hipModuleLaunchKernel(f, grid, 1, 1, kWave, 1, 1, lds, stream, nullptr, config);
}

}  // namespace

hipError_t launch_fold_slots(const FoldParams& p, const SlotParams& sp, const SlotKernels* spec, int64_t n_waves, int lane_events,
                             hipStream_t stream) {
  if (n_waves <= 0) return hipSuccess;
  const unsigned lds = lane_events == 8 ? Geo<8>::lds_bytes(Geo<8>::kAuxSorted) : Geo<16>::lds_bytes(Geo<16>::kAuxSorted);
  if (spec) {
    FoldParams args = p;
    return launch_module_kernel(lane_events == 8 ? spec->csr8 : spec->csr16, (unsigned)n_waves, lds, stream, &args, sizeof(args));
  }
  if (lane_events == 8)
    hipLaunchKernelGGL((fold_slots_kernel<8>), dim3((unsigned)n_waves), dim3(kWave), lds, stream, p, sp);
  else
    hipLaunchKernelGGL((fold_slots_kernel<16>), dim3((unsigned)n_waves), dim3(kWave), lds, stream, p, sp);
  return hipGetLastError();
}

hipError_t launch_fold_slots_tiled(const FoldParams& p, const SlotParams& sp, const SlotKernels* spec, const TileTable& t, int64_t n_waves,
                                   int subs, hipStream_t stream) {
  if (n_waves <= 0 || t.n_vrows <= 0) return hipSuccess;
  if (spec) {
    struct Args { FoldParams p; TileTable t; } args;  // the kernel-argument segment: both structs are 8-byte aligned
    static_assert(sizeof(FoldParams) % 8 == 0, "TileTable must follow FoldParams without padding");
    args.p = p;
    args.t = t;
    return launch_module_kernel(subs == 1 ? spec->tiled1 : spec->tiled2, (unsigned)n_waves, 0, stream, &args, sizeof(args));
  }
  if (subs == 1)
    hipLaunchKernelGGL((fold_slots_tiled_kernel<1>), dim3((unsigned)n_waves), dim3(kWave), 0, stream, p, t, sp);
  else
    hipLaunchKernelGGL((fold_slots_tiled_kernel<2>), dim3((unsigned)n_waves), dim3(kWave), 0, stream, p, t, sp);
  return hipGetLastError();
}

}  // namespace surge
