// engine.hip — host side of libsurge_replay.so: handle, device memory, launch sequencing and
// the extern "C" boundary declared in include/surge_replay.h.  No torch types, no CPU fold:
// if HIP is unusable every entry point reports SURGE_E_DEVICE.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <string>
#include <vector>

#include "replay_internal.h"
#include "f64_text.h"

using namespace surge;

namespace {

thread_local std::string g_last_error;

struct DevBuf {
  void* ptr = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&ptr, bytes ? bytes : 16);
    if (e == hipSuccess) cap = bytes ? bytes : 16;
    return e;
  }
  // for the buffers of a stream of micro-batches: the next batch is a few per cent larger or smaller than this one, and a
  // buffer that grows is freed — hipFree waits for the whole device (3 - 6 ms spikes per fetch on the bytes -> states path)
  hipError_t reserve_roomy(size_t bytes) { return bytes <= cap ? hipSuccess : reserve(bytes + bytes / 4 + 4096); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
};

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace

struct surge_replay_handle {
  int device = 0;
  int n_cus = 256;
  hipStream_t stream = nullptr;
  surge_replay_schema schema{};
  bool v2 = false;                     // ABI v2 slot schema: folds only through fold_slots.hip
  surge_replay_schema_v2 schema2{};
  alignas(16) unsigned char slot_params[kSlotParamsBytes] = {};
  SlotKernels* spec = nullptr;         // v2: the kernels hiprtc compiled for this schema (process-wide cache); nullptr = interpreter
  V1Kernels* spec1 = nullptr;          // v1: the flat kernel compiled for this handle's op table (acquired at the first flat fold)
  bool spec1_tried = false;
  double spec1_compile_ms = 0.0;
  std::string spec1_why;
  V1Kernels* lanes1 = nullptr;         // v1: the lane-per-row kernels (SORTED / CHUNKED / ROWS) compiled for the op table (first lane fold / prepare)
  bool lanes1_tried = false;
  double lanes1_compile_ms = 0.0;
  std::string lanes1_why;
  double spec_compile_ms = 0.0;
  std::string spec_why;                // why the interpreter runs instead / which libhiprtc compiled the kernels
  std::string err;
  std::mutex err_mu;  // concurrent point readers may fail at the same time

  // the bound log (owned copies or borrowed device pointers)
  DevBuf own_seg_off, own_events, own_init, own_state;
  const int64_t* d_seg_off = nullptr;
  const uint4* d_events = nullptr;
  const uint4* d_init = nullptr;
  uint4* d_state = nullptr;
  int64_t n_agg = 0, n_events = 0;
  bool bound = false;
  bool log_valid = false;  // false once the resident state was grown past the bound CSR (append_* only until the next load)

  // analysis of the bound CSR (computed at load/bind time)
  CsrAnalysis an{};
  DevBuf d_analysis, nz_off, nz_map, block_counts;
  int64_t n_nz = 0;
  DevBuf perm, counter;  // SORTED: segments by descending length (built lazily, per bound log)
  // scratch of the index builds (index_kernels.hip): rocPRIM temp, sort keys / values, the chunk table's counts and its
  // rows in aggregate order; released once the bound log's index stands
  // (two allocations, carved: a hipMalloc costs 50 - 300 us and a hipFree waits for the device — ten of each were most of
  // the chunk table's 3 - 7 ms in round 5)
  DevBuf ix_arena, ix_cnt;
  bool perm_valid = false;
  // CHUNKED / TILED: the chunk table (built lazily, per bound log), the chunk summaries and the list of cut aggregates
  struct ChunkIndex {
    DevBuf arena;  // one allocation; the pointers below are views into it
    void *v_start = nullptr, *v_len = nullptr, *v_info = nullptr, *v_seg = nullptr, *v_side = nullptr, *r_slot0 = nullptr, *r_c = nullptr, *r_out = nullptr;
    int64_t n_vrows = 0, n_cut_rows = 0;
    uint32_t T = 0;  // the chunk target the table was built for (0 = none built)
    void release() {
      arena.release();
      v_start = v_len = v_info = v_seg = v_side = r_slot0 = r_c = r_out = nullptr;
      n_vrows = n_cut_rows = 0;
      T = 0;
    }
  };
  ChunkIndex cidx;              // CHUNKED: rows tiled from their 128-byte lines in the CSR log
  ChunkIndex tidx;              // TILED: rows copied to tile boundaries
  DevBuf t_tiles, t_gsub;  // TILED: the tile-major copy of the log, first subtile of every group
  int64_t t_n_sub = 0;          // subtiles (8 KiB each) of the tile-major copy
  bool tiled_valid = false;
  // one-off costs of the bound log's index (device time between HIP events), reported by surge_replay_layout_info
  hipEvent_t ev_i0 = nullptr, ev_i1 = nullptr, ev_r0 = nullptr, ev_r1 = nullptr;
  bool index_timed = false, relayout_timed = false;
  int32_t index_algo = 0;

  // per-fold scratch
  DevBuf plan, batch_group_agg, batch_group_off, batch_events, poison_count, gather_idx, gather_out, scan_totals;

  hipEvent_t ev_total0 = nullptr, ev_k0 = nullptr, ev_k1 = nullptr, ev_total1 = nullptr, ev_h0 = nullptr,
             ev_h1 = nullptr;
  bool timing_valid = false, h2d_valid = false;
  surge_replay_stats_t st{};
  // one HIP-event pair per fold since the last stats_reset (kernel time of the dominant kernel)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> fold_events;
  size_t folds_since_reset = 0;

  // append_events: device group-by scratch (stream_kernels.hip) and pinned H2D staging of host batches
  DevBuf gb_temp, gb_u32, gb_flags, gb_agg_idx, gb_events;
  // hipHostMalloc'ed staging of host batches (agg_idx then events), two areas used in turn: the host fills one while the
  // copy engine still drains the other; ev_staged[k] = "the H2D copies out of area k are done"
  void* pinned[2] = {nullptr, nullptr};
  size_t pinned_cap[2] = {0, 0};
  hipEvent_t ev_staged[2] = {nullptr, nullptr};
  bool staged_busy[2] = {false, false};
  int pinned_next = 0;
  uint32_t* host_flags = nullptr;  // pinned: {groups, bad, skipped batches} of the last device group-by, copied back async
  uint32_t skipped_seen = 0;       // skipped batches already reported to the host

  // the packer's staging log (surge_replay_stage_events_device): aggregate indices (u32) and events (16 B) in topic order
  DevBuf stage_keys, stage_events;
  int64_t staged_n = 0, stage_cap = 0;

  DevBuf published;                      // the last committed snapshot (surge_replay_snapshot_delta), n_agg x 64 B
  int64_t published_n = 0;
  const uint8_t* encode_filter = nullptr;  // surge_replay_set_encode_filter
  JsonSide json_side{};                  // Double-text tables (device copy, made on first use), side string columns
  DevBuf f64_tables, nan_count;

  CommState* comm = nullptr;  // the snapshot exchange (comm.hip), created by surge_replay_comm_init
  DevBuf gathered[2];         // handle-owned output of allgather_snapshot(d_out = NULL), per slot
  int64_t gathered_rows[2] = {0, 0};
  int32_t comm_world = 1;

  // host mirror for point reads (S2)
  std::shared_mutex mu;  // readers share it against the published mirror; snapshot / device reads take it exclusively
  std::vector<uint8_t> mirror;
  std::atomic<int64_t> fold_epoch{0};
  int64_t delta_epoch = -1, delta_n = -1;  // fold epoch / aggregate count the last snapshot_delta's kinds describe
  int64_t mirror_epoch = -1;
};

namespace {

int32_t fail(surge_replay_handle* h, int32_t code, const std::string& msg) {
  if (h) {
    std::lock_guard<std::mutex> lk(h->err_mu);
    h->err = msg;
  }
  g_last_error = msg;  // thread-local: what surge_replay_last_error(NULL) returns to the failing thread
  return code;
}

int32_t fail_hip(surge_replay_handle* h, hipError_t e, const char* what) {
  std::string m = std::string(what) + ": " + hipGetErrorString(e);
  const int32_t code = (e == hipErrorOutOfMemory) ? SURGE_E_NOMEM : SURGE_E_DEVICE;
  return fail(h, code, m);
}

#define HIPCHK(h, call)                                   \
  do {                                                    \
    hipError_t e_ = (call);                               \
    if (e_ != hipSuccess) return fail_hip(h, e_, #call); \
  } while (0)

// the flat kernel for this handle's op table: compiled (hiprtc, ~1 s) the first time a process folds with the table, shared
// by every handle with the same table; nullptr = the ahead-of-time kernel (why: surge_replay_kernel_info)
const V1Kernels* flat_spec(surge_replay_handle* h, const FoldParams& p) {
  if (!h->spec1_tried) {
    h->spec1_tried = true;
    v1_kernels_acquire(p.table, h->device, V1_FLAT, &h->spec1, &h->spec1_compile_ms, &h->spec1_why);
    if (h->spec1) h->spec1_why = std::string("flat kernel compiled for the op table by ") + rtc_library_path();
  }
  return h->spec1;
}

// ... and the lane-per-row kernels (SORTED / CHUNKED / ROWS): compiled at surge_replay_prepare or the first such fold — like
// the per-log index, before the fold's timing events, never between them
const V1Kernels* lane_spec(surge_replay_handle* h, const FoldParams& p) {
  if (!h->lanes1_tried) {
    h->lanes1_tried = true;
    v1_kernels_acquire(p.table, h->device, V1_LANES, &h->lanes1, &h->lanes1_compile_ms, &h->lanes1_why);
    if (h->lanes1) h->lanes1_why = "compiled for the op table";
  }
  return h->lanes1;
}

void fill_params(const surge_replay_schema& schema, FoldParams& p) {
  std::memset(&p, 0, sizeof(p));
  for (int i = 0; i < kTableEntries; ++i) {
    const uint32_t d = ((uint32_t)i < schema.n_types && i < SURGE_MAX_EVENT_TYPES) ? schema.desc[i] : SURGE_D_POISON;
    uint32_t* w = p.table[i];
    const uint32_t cop = d & SURGE_D_COUNT_MASK, sop = d & SURGE_D_SUM_MASK, cls = d & SURGE_CLS_MASK;
    if (d & SURGE_D_POISON) {
      w[TW_POISON] = ~0u;  // everything else stays zero: a throwing event has no effect on the fields
      w[TW_FLAGS] = 1u;
      if (i == kTableEntries - 1) {  // [17]: the null event that pads the last tile — identity on every state
        w[TW_POISON] = 0u;
        w[TW_FLAGS] = 0u;
      }
      continue;
    }
    if (cls == SURGE_CLS_DELETE) {
      w[TW_DELETE] = ~0u;  // a tombstone has no field ops
      w[TW_NOT_REQUIRE] = ~0u;
      w[TW_FLAGS] = 1u << 16;
      continue;
    }
    w[TW_CNT_NZ] = (cop == SURGE_D_COUNT_ADD || cop == SURGE_D_COUNT_SUB) ? ~0u : 0u;
    w[TW_CNT_NEG] = (cop == SURGE_D_COUNT_SUB) ? ~0u : 0u;
    w[TW_CNT_SET] = (cop == SURGE_D_COUNT_SET) ? ~0u : 0u;
    w[TW_VER_SET] = (d & SURGE_D_VERSION_SET) ? ~0u : 0u;
    w[TW_SUM_NZ] = (sop == SURGE_D_SUM_ADD || sop == SURGE_D_SUM_SUB) ? ~0u : 0u;
    w[TW_SUM_NEG] = (sop == SURGE_D_SUM_SUB) ? ~0u : 0u;
    w[TW_BAL_SET] = (d & SURGE_D_BALANCE_SET) ? ~0u : 0u;
    w[TW_EVC] = (d & SURGE_D_EVCOUNT_INC) ? 1u : 0u;
    w[TW_MATERIALIZES] = (cls == SURGE_CLS_MATERIALIZE || cls == SURGE_CLS_CREATE) ? ~0u : 0u;
    w[TW_NOT_REQUIRE] = (cls != SURGE_CLS_REQUIRE) ? ~0u : 0u;
    w[TW_CREATE] = (cls == SURGE_CLS_CREATE) ? ~0u : 0u;
    w[TW_MIN] = (d & SURGE_D_MIN_ARG) ? ~0u : 0u;
    w[TW_MAX] = (d & SURGE_D_MAX_ARG) ? ~0u : 0u;
    w[TW_FLAGS] = 0u;  // bit0 poison, bit16 delete; materializes goes in its own accumulator (TW_MATERIALIZES & 1)
  }
  const surge_state64& d = schema.default_state;
  p.d_count = d.count;
  p.d_version = d.version;
  p.d_sum = d.sum64;
  std::memcpy(&p.d_balance, &d.balance, 8);
  p.d_min = d.min_arg;
  p.d_max = d.max_arg;
  p.d_evcount = d.event_count;
}

void fill_params(const surge_replay_handle* h, FoldParams& p) { fill_params(h->schema, p); }

// Wave-task size in events: a multiple of one tile (64 * lane_events events), about kTaskBytes of
// events at most, small enough that short logs still spread over the chip.
int64_t choose_task_events(int64_t n_events, int lane_events) {
  const int64_t tile = (int64_t)kWave * lane_events;
  int64_t task_bytes = kTaskBytes;
  if (const char* v = std::getenv("SURGE_REPLAY_TASK_KB")) task_bytes = (int64_t)std::atoi(v) * 1024;
  const int64_t max_tiles = task_bytes / (tile * 16) > 0 ? task_bytes / (tile * 16) : 1;
  int64_t target = kTargetTasks;
  if (const char* v = std::getenv("SURGE_REPLAY_TARGET_TASKS")) target = std::atoi(v) > 0 ? std::atoi(v) : target;
  int64_t tiles = (n_events / target + tile - 1) / tile;
  if (tiles < 1) tiles = 1;
  if (tiles > max_tiles) tiles = max_tiles;
  return tiles * tile;
}

// Events per lane per tile for each kernel (8 -> 8 KiB tiles and twice the resident waves, 16 -> 16 KiB
// tiles and half the per-tile scan overhead).  Tunable through the environment for experiments.
int env_lane_events(const char* name, int dflt) {
  const char* v = std::getenv(name);
  if (!v) return dflt;
  const int x = std::atoi(v);
  return (x == 8 || x == 16 || x == 32) ? x : dflt;
}

int32_t validate_schema(const surge_replay_schema* s) {
  if (!s) return fail(nullptr, SURGE_E_INVALID, "schema is NULL");
  if (s->abi_version != SURGE_REPLAY_ABI_VERSION) return fail(nullptr, SURGE_E_UNSUPPORTED, "schema.abi_version mismatch");
  if (s->state_size != 64 || s->event_size != 16)
    return fail(nullptr, SURGE_E_UNSUPPORTED, "only 64-byte states and 16-byte events are supported");
  if (s->n_types < 1 || s->n_types > SURGE_MAX_EVENT_TYPES) return fail(nullptr, SURGE_E_INVALID, "schema.n_types out of range");
  const uint32_t known = SURGE_CLS_MASK | SURGE_D_POISON | SURGE_D_COUNT_MASK | SURGE_D_VERSION_SET | SURGE_D_SUM_MASK |
                         SURGE_D_BALANCE_SET | SURGE_D_MIN_ARG | SURGE_D_MAX_ARG | SURGE_D_EVCOUNT_INC;
  for (uint32_t i = 0; i < s->n_types; ++i) {
    if (s->desc[i] & ~known) return fail(nullptr, SURGE_E_UNSUPPORTED, "schema descriptor uses unknown bits");
    if ((s->desc[i] & SURGE_D_SUM_MASK) == SURGE_D_SUM_MASK) return fail(nullptr, SURGE_E_UNSUPPORTED, "invalid sum64 op");
  }
  return SURGE_OK;
}

// Analyse the bound CSR once: monotone? empty segments? uniform length?  Synchronous (load time).
int32_t analyze_bound(surge_replay_handle* h) {
  HIPCHK(h, h->d_analysis.reserve(sizeof(CsrAnalysis)));
  h->n_nz = h->n_agg;
  std::memset(&h->an, 0, sizeof(h->an));
  if (h->n_agg == 0) return SURGE_OK;
  HIPCHK(h, launch_analyze_csr(h->d_seg_off, h->n_agg, (CsrAnalysis*)h->d_analysis.ptr, h->stream));
  HIPCHK(h, hipMemcpyAsync(&h->an, h->d_analysis.ptr, sizeof(CsrAnalysis), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->an.bad) return fail(h, SURGE_E_INVALID, "seg_off is not monotone non-decreasing");
  if (h->an.first < 0 || h->an.last > h->n_events)
    return fail(h, SURGE_E_INVALID, "seg_off range exceeds the events buffer");
  if (h->an.n_empty > 0) {
    // kernel-facing CSR without empty segments (+ rank -> aggregate map)
    const int64_t nb = (h->n_agg + 1023) / 1024;
    h->n_nz = h->n_agg - h->an.n_empty;
    HIPCHK(h, h->block_counts.reserve((size_t)(nb + 1) * 8));
    HIPCHK(h, h->nz_off.reserve((size_t)(h->n_nz + 1) * 8));
    HIPCHK(h, h->nz_map.reserve((size_t)(h->n_nz > 0 ? h->n_nz : 1) * 8));
    HIPCHK(h, launch_compact_nonempty(h->d_seg_off, h->n_agg, (int64_t*)h->block_counts.ptr, (int64_t*)h->nz_off.ptr,
                                      (int64_t*)h->nz_map.ptr, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return SURGE_OK;
}

int64_t algorithmic_bytes(int64_t n_events, int64_t n_agg, bool has_init) {
  return 16 * n_events + 8 * (n_agg + 1) + 64 * n_agg * (has_init ? 2 : 1);
}

constexpr size_t kMaxTimedFolds = 256;

// Event pair bracketing the dominant kernel of this fold; pairs are kept per fold (up to
// kMaxTimedFolds since the last stats_reset) so a benchmark can average them without syncing per step.
int32_t next_fold_events(surge_replay_handle* h, hipEvent_t* e0, hipEvent_t* e1) {
  size_t i = h->folds_since_reset < kMaxTimedFolds ? h->folds_since_reset : kMaxTimedFolds - 1;
  while (h->fold_events.size() <= i) {
    hipEvent_t a = nullptr, b = nullptr;
    HIPCHK(h, hipEventCreate(&a));
    hipError_t e = hipEventCreate(&b);
    if (e != hipSuccess) {
      (void)hipEventDestroy(a);
      return fail_hip(h, e, "hipEventCreate");
    }
    h->fold_events.emplace_back(a, b);
  }
  *e0 = h->fold_events[i].first;
  *e1 = h->fold_events[i].second;
  h->ev_k0 = *e0;
  h->ev_k1 = *e1;
  h->folds_since_reset += 1;
  return SURGE_OK;
}

// plan + flat fold over an arbitrary kernel-facing CSR
int32_t run_flat(surge_replay_handle* h, FoldParams& p, const int64_t* off, int64_t n_seg, int64_t span_events) {
  // short rows (a head in almost every lane): 8 KiB tiles — three waves per SIMD instead of two hide the per-head state
  // stores better than the halved scan overhead of 16 KiB tiles pays (uniform 1..32 events: 0.48 -> 0.53 of peak at 20 M
  // aggregates, 0.36 -> 0.42 at 2 M; Zipf(1..4096), mean 460: 16 KiB tiles stay ahead)
  const int le = env_lane_events("SURGE_REPLAY_LE_FLAT", (n_seg > 0 && span_events / n_seg < 64) ? 8 : 16);
  const int64_t task_events = choose_task_events(span_events, le);
  const int64_t n_tasks = (span_events + task_events - 1) / task_events;
  HIPCHK(h, h->plan.reserve((size_t)(n_tasks + 1) * 8));
  HIPCHK(h, launch_plan(off, n_seg, task_events, n_tasks, (int64_t*)h->plan.ptr, h->stream));
  p.seg_off = off;
  p.plan = (const int64_t*)h->plan.ptr;
  p.n_seg = n_seg;
  hipEvent_t e0, e1;
  const int32_t rc = next_fold_events(h, &e0, &e1);
  if (rc != SURGE_OK) return rc;
  const V1Kernels* spec = flat_spec(h, p);  // (a process's first fold with this op table compiles it: before the timed region, not inside it)
  HIPCHK(h, hipEventRecord(e0, h->stream));
  HIPCHK(h, launch_fold_flat(p, spec, n_tasks, le, h->stream));
  HIPCHK(h, hipEventRecord(e1, h->stream));
  h->st.n_tasks = (int32_t)n_tasks;
  return SURGE_OK;
}

// The persistent kernels pull groups from an atomic ticket counter that the last wave of every launch re-arms; the
// host only zeroes it when it is allocated.
int32_t dispenser_begin(surge_replay_handle* h, FoldParams& p) {
  if (!h->counter.ptr) {
    HIPCHK(h, h->counter.reserve(16));
    HIPCHK(h, hipMemset(h->counter.ptr, 0, 16));
  }
  p.counter = (unsigned long long*)h->counter.ptr;
  return SURGE_OK;
}

// scratch for ordering n rows by length (vals_b only when the caller does not supply its own output).  max_key: the largest
// key among them — below kCountSortMaxBins the hand-written counting sort orders them (its histograms are the only scratch),
// else rocPRIM's radix sort (temp + key / value double buffers).  min_temp: bytes the caller wants of `temp` besides.
int32_t index_scratch(surge_replay_handle* h, int64_t n, bool need_vals_b, int64_t max_key, size_t min_temp, IndexScratch* sc, size_t extra_bytes = 0,
                      void** extra = nullptr) {
  const size_t rows = (size_t)(n > 0 ? n : 1);
  const char* sort_env = std::getenv("SURGE_REPLAY_INDEX_SORT");  // "radix": rocPRIM's radix sort whatever the keys (the test that compares the two orders)
  const bool force_radix = sort_env && std::strcmp(sort_env, "radix") == 0;
  sc->counting = !force_radix && max_key >= 0 && max_key < kCountSortMaxBins;
  sc->max_key = (uint32_t)(max_key > 0 ? max_key : 0);
  sc->n_cus = h->n_cus;
  size_t tb = 0;
  if (sc->counting) tb = count_sort_scratch_bytes(n, sc->max_key, h->n_cus);
  else HIPCHK(h, index_temp_bytes(n, &tb));
  tb = tb > min_temp ? tb : min_temp;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t o_keys_a = up(tb), o_keys_b = o_keys_a + up(rows * 4), o_vals_a = o_keys_b + (sc->counting ? 0 : up(rows * 4)),
               o_vals_b = o_vals_a + (sc->counting ? 0 : up(rows * 8)), o_extra = o_vals_b + (need_vals_b ? up(rows * 8) : 0);
  HIPCHK(h, h->ix_arena.reserve(o_extra + extra_bytes));
  char* base = (char*)h->ix_arena.ptr;
  sc->temp = base;
  sc->temp_bytes = tb;
  sc->keys_a = (uint32_t*)(base + o_keys_a);
  sc->keys_b = sc->counting ? nullptr : (uint32_t*)(base + o_keys_b);
  sc->vals_a = sc->counting ? nullptr : (int64_t*)(base + o_vals_a);
  sc->vals_b = need_vals_b ? (int64_t*)(base + o_vals_b) : nullptr;
  if (extra) *extra = base + o_extra;
  return SURGE_OK;
}

// a bound log's index stands: give the build scratch back (a 10 M-aggregate log's is ~0.6 GB); micro-batch sorts keep theirs
void index_scratch_release(surge_replay_handle* h) {
  h->ix_arena.release();
  h->ix_cnt.release();
}

// v2: length-sort the kernel-facing segments (once per bound log / per micro-batch), then one lane per segment
int32_t run_slots(surge_replay_handle* h, FoldParams& p, const int64_t* off, int64_t n_seg, bool cache_perm) {
  if (!cache_perm || !h->perm_valid) {
    HIPCHK(h, h->perm.reserve((size_t)(n_seg > 0 ? n_seg : 1) * 8));
    IndexScratch sc;
    // (a micro-batch's longest group is not known on the host: the radix sort; a bound log's longest aggregate is)
    const int32_t rcs = index_scratch(h, n_seg, false, cache_perm ? h->an.max_len : -1, 0, &sc);
    if (rcs != SURGE_OK) return rcs;
    HIPCHK(h, launch_sort_by_length(off, n_seg, sc, (int64_t*)h->perm.ptr, h->stream));
    h->perm_valid = cache_perm;
  }
  p.seg_off = off;
  p.plan = (const int64_t*)h->perm.ptr;
  {
    const int32_t rcd = dispenser_begin(h, p);
    if (rcd != SURGE_OK) return rcd;
  }
  p.n_seg = n_seg;
  const int64_t groups = (n_seg + kWave - 1) / kWave;
  // the interpreter is VALU-bound and light on registers (93 VGPRs): 8 KiB tiles and as many resident waves as LDS
  // allows; the schema-specialised kernels keep their tile in registers like the v1 sorted-rows kernel: 16 KiB tiles, 8 waves
  const int le = env_lane_events("SURGE_REPLAY_LE_SLOTS", h->spec ? 16 : 8) == 16 ? 16 : 8;
  int64_t per_cu = le == 8 ? (h->spec ? 12 : 14) : 8;
  if (const char* v = std::getenv("SURGE_REPLAY_SLOTS_WAVES")) per_cu = std::atoi(v) > 0 ? std::atoi(v) : per_cu;
  const int64_t slots = (int64_t)h->n_cus * per_cu;
  const int64_t n_waves = groups < slots ? groups : slots;
  hipEvent_t e0, e1;
  const int32_t rc = next_fold_events(h, &e0, &e1);
  if (rc != SURGE_OK) return rc;
  HIPCHK(h, hipEventRecord(e0, h->stream));
  HIPCHK(h, launch_fold_slots(p, *(const SlotParams*)h->slot_params, h->spec, n_waves, le, h->stream));
  HIPCHK(h, hipEventRecord(e1, h->stream));
  h->st.n_tasks = (int32_t)n_waves;
  return SURGE_OK;
}

struct FoldPlan;
int32_t ensure_index(surge_replay_handle* h, const FoldPlan& pl);
int32_t run_slots_tiled(surge_replay_handle* h, FoldParams& p);

int32_t fold_slots_bound(surge_replay_handle* h, bool tiled) {
  FoldParams p;
  std::memset(&p, 0, sizeof(p));
  p.events = h->d_events;
  p.n_events = h->n_events;
  p.init = h->d_init;
  p.out = h->d_state;
  const int64_t span = h->an.last - h->an.first;
  HIPCHK(h, hipEventRecord(h->ev_total0, h->stream));
  h->st.n_tasks = 0;
  if (h->n_agg > 0 && span > 0) {
    if (h->an.max_len >= (1ll << 31)) return fail(h, SURGE_E_UNSUPPORTED, "segments must be shorter than 2^31 events");
    const bool nz = h->an.n_empty > 0;
    if (nz) p.out_map = (const int64_t*)h->nz_map.ptr;
    const int32_t rc = tiled ? run_slots_tiled(h, p)
                             : run_slots(h, p, nz ? (const int64_t*)h->nz_off.ptr : h->d_seg_off, nz ? h->n_nz : h->n_agg, true);
    if (rc != SURGE_OK) return rc;
  } else {
    hipEvent_t e0, e1;
    const int32_t rc = next_fold_events(h, &e0, &e1);
    if (rc != SURGE_OK) return rc;
    HIPCHK(h, hipEventRecord(e0, h->stream));
    HIPCHK(h, hipEventRecord(e1, h->stream));
  }
  if (h->an.n_empty > 0 || span == 0) HIPCHK(h, launch_fill_empty(h->d_seg_off, h->n_agg, h->d_init, h->d_state, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_total1, h->stream));
  h->timing_valid = true;
  h->st.last_algo = tiled ? SURGE_ALGO_TILED : SURGE_ALGO_SLOTS;
  h->st.n_folds += 1;
  h->st.n_poisoned = -1;
  h->fold_epoch.fetch_add(1);
  return SURGE_OK;
}

}  // namespace

extern "C" {

int32_t surge_replay_default_schema(surge_replay_schema* out) {
  if (!out) return fail(nullptr, SURGE_E_INVALID, "out is NULL");
  std::memset(out, 0, sizeof(*out));
  out->abi_version = SURGE_REPLAY_ABI_VERSION;
  out->state_size = 64;
  out->event_size = 16;
  out->n_types = 7;
  const uint32_t extras = SURGE_D_MIN_ARG | SURGE_D_MAX_ARG | SURGE_D_EVCOUNT_INC;
  out->desc[SURGE_EVT_NOOP] = SURGE_CLS_MATERIALIZE;
  out->desc[SURGE_EVT_INC] = SURGE_CLS_MATERIALIZE | SURGE_D_COUNT_ADD | SURGE_D_VERSION_SET | SURGE_D_SUM_ADD | extras;
  out->desc[SURGE_EVT_DEC] = SURGE_CLS_MATERIALIZE | SURGE_D_COUNT_SUB | SURGE_D_VERSION_SET | SURGE_D_SUM_SUB | extras;
  out->desc[SURGE_EVT_CREATE] = SURGE_CLS_CREATE | SURGE_D_BALANCE_SET | SURGE_D_EVCOUNT_INC;
  out->desc[SURGE_EVT_SET_BALANCE] = SURGE_CLS_REQUIRE | SURGE_D_BALANCE_SET | SURGE_D_EVCOUNT_INC;
  out->desc[SURGE_EVT_DELETE] = SURGE_CLS_DELETE;
  out->desc[SURGE_EVT_THROW] = SURGE_D_POISON;
  out->default_state.min_arg = 0x7fffffff;
  out->default_state.max_arg = (int32_t)0x80000000;
  out->default_state.flags = SURGE_STATE_PRESENT;
  return SURGE_OK;
}

int32_t surge_replay_create(const surge_replay_schema* schema, int32_t device_id, surge_replay_handle** out) {
  if (!out) return fail(nullptr, SURGE_E_INVALID, "out is NULL");
  *out = nullptr;
  const int32_t v = validate_schema(schema);
  if (v != SURGE_OK) return v;
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0)
    return fail(nullptr, SURGE_E_DEVICE, "no usable HIP device (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= n_dev) return fail(nullptr, SURGE_E_INVALID, "device_id out of range");
  surge_replay_handle* h = new (std::nothrow) surge_replay_handle();
  if (!h) return fail(nullptr, SURGE_E_NOMEM, "out of host memory");
  h->device = device_id;
  h->schema = *schema;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) h->n_cus = cus;
  }
  DeviceGuard g(device_id);
  if (!g.ok) {
    delete h;
    return fail(nullptr, SURGE_E_DEVICE, "hipSetDevice failed");
  }
  hipEvent_t* evs[] = {&h->ev_total0, &h->ev_total1, &h->ev_h0, &h->ev_h1, &h->ev_i0, &h->ev_i1, &h->ev_r0, &h->ev_r1};
  for (hipEvent_t* ev : evs) {
    e = hipEventCreate(ev);
    if (e != hipSuccess) {
      const int32_t rc = fail_hip(nullptr, e, "hipEventCreate");
      surge_replay_destroy(h);
      return rc;
    }
  }
  h->st.n_poisoned = -1;
  *out = h;
  return SURGE_OK;
}

static int32_t validate_schema_v2(const surge_replay_schema_v2* sc) {
  if (!sc) return fail(nullptr, SURGE_E_INVALID, "schema is NULL");
  if (sc->abi_version != SURGE_REPLAY_ABI_VERSION_2) return fail(nullptr, SURGE_E_UNSUPPORTED, "schema.abi_version is not 2");
  if (sc->state_size != 64 || sc->event_size != 16) return fail(nullptr, SURGE_E_UNSUPPORTED, "only 64-byte states and 16-byte events are supported");
  if (sc->n_types < 1 || sc->n_types > SURGE_MAX_EVENT_TYPES) return fail(nullptr, SURGE_E_INVALID, "schema.n_types out of range");
  if (sc->n_slots < 1 || sc->n_slots > SURGE_MAX_SLOTS) return fail(nullptr, SURGE_E_INVALID, "schema.n_slots out of range");
  if (sc->flags & ~SURGE_V2_COUNT_EVENTS) return fail(nullptr, SURGE_E_UNSUPPORTED, "schema.flags uses unknown bits");
  for (uint32_t i = 0; i < sc->n_slots; ++i) {
    if (sc->slot[i].type < SURGE_SLOT_I32 || sc->slot[i].type > SURGE_SLOT_F64) return fail(nullptr, SURGE_E_UNSUPPORTED, "unknown slot type");
    if (sc->slot[i].source > SURGE_SRC_ONE) return fail(nullptr, SURGE_E_UNSUPPORTED, "unknown operand source");
  }
  for (uint32_t t = 0; t < sc->n_types; ++t) {
    if (sc->cls[t] & ~(SURGE_CLS_MASK | SURGE_D_POISON)) return fail(nullptr, SURGE_E_UNSUPPORTED, "cls uses unknown bits");
    for (uint32_t i = 0; i < 8; ++i) {
      const uint32_t op = (sc->ops[t] >> (4 * i)) & 15u;
      if (op > SURGE_OP_MAX) return fail(nullptr, SURGE_E_UNSUPPORTED, "unknown slot operation");
      if (i >= sc->n_slots && op != SURGE_OP_KEEP) return fail(nullptr, SURGE_E_INVALID, "operation on a slot the schema does not declare");
    }
  }
  return SURGE_OK;
}

int32_t surge_replay_create_v2(const surge_replay_schema_v2* sc, int32_t device_id, surge_replay_handle** out) {
  if (!out) return fail(nullptr, SURGE_E_INVALID, "out is NULL");
  *out = nullptr;
  {
    const int32_t rcv = validate_schema_v2(sc);
    if (rcv != SURGE_OK) return rcv;
  }
  surge_replay_schema v1;
  surge_replay_default_schema(&v1);  // the handle's v1 half is inert; every fold of a v2 handle goes through the slot kernel
  const int32_t rc = surge_replay_create(&v1, device_id, out);
  if (rc != SURGE_OK) return rc;
  (*out)->v2 = true;
  (*out)->schema2 = *sc;
  slot_params_from_schema(*sc, (SlotParams*)(*out)->slot_params);
  {
    // the schema never changes: compile the slot kernels for it (hiprtc, ~1 s the first time a process sees the schema);
    // any failure leaves the generic interpreter in charge and is reported by surge_replay_kernel_info, not here
    DeviceGuard g(device_id);
    surge_replay_handle* h = *out;
    slot_kernels_acquire(*sc, *(const SlotParams*)h->slot_params, device_id, &h->spec, &h->spec_compile_ms, &h->spec_why);
    if (h->spec) h->spec_why = std::string("compiled by ") + rtc_library_path();
  }
  return SURGE_OK;
}

int32_t surge_replay_destroy(surge_replay_handle* h) {
  if (!h) return SURGE_OK;
  DeviceGuard g(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (h->comm) comm_destroy(h->comm);
  h->comm = nullptr;
  for (int k = 0; k < 2; ++k) {
    if (h->pinned[k]) (void)hipHostFree(h->pinned[k]);
    h->pinned[k] = nullptr;
    if (h->ev_staged[k]) (void)hipEventDestroy(h->ev_staged[k]);
    h->ev_staged[k] = nullptr;
  }
  if (h->host_flags) (void)hipHostFree(h->host_flags);
  h->host_flags = nullptr;
  h->cidx.release();
  h->tidx.release();
  DevBuf* bufs[] = {&h->gb_temp, &h->gb_u32, &h->gb_flags, &h->gb_agg_idx, &h->gb_events, &h->published, &h->gathered[0], &h->gathered[1], &h->f64_tables, &h->nan_count, &h->ix_arena, &h->ix_cnt, &h->stage_keys, &h->stage_events, &h->t_tiles, &h->t_gsub, &h->perm, &h->counter, &h->own_seg_off, &h->own_events, &h->own_init, &h->own_state, &h->d_analysis, &h->nz_off,
                    &h->nz_map, &h->block_counts, &h->plan, &h->batch_group_agg, &h->batch_group_off,
                    &h->batch_events, &h->poison_count, &h->gather_idx, &h->gather_out, &h->scan_totals};
  for (DevBuf* b : bufs) b->release();
  hipEvent_t evs[] = {h->ev_total0, h->ev_total1, h->ev_h0, h->ev_h1, h->ev_i0, h->ev_i1, h->ev_r0, h->ev_r1};
  for (hipEvent_t ev : evs)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& pr : h->fold_events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  delete h;
  return SURGE_OK;
}

const char* surge_replay_last_error(const surge_replay_handle* h) {
  return h ? h->err.c_str() : g_last_error.c_str();
}

int32_t surge_replay_set_stream(surge_replay_handle* h, void* hip_stream) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  h->stream = (hipStream_t)hip_stream;
  return SURGE_OK;
}

int32_t surge_replay_get_stream(surge_replay_handle* h, void** hip_stream_out) {
  if (!h || !hip_stream_out) return fail(h, SURGE_E_INVALID, "NULL argument");
  *hip_stream_out = (void*)h->stream;
  return SURGE_OK;
}

// A micro-batch whose aggregate indices were out of range is skipped on the device (stream_kernels.hip) and reported
// here, at the host's next synchronisation point, once.
static int32_t report_skipped_batches(surge_replay_handle* h) {
  if (!h->host_flags) return SURGE_OK;
  const uint32_t skipped = h->host_flags[2];
  if (skipped == h->skipped_seen) return SURGE_OK;
  const uint32_t n = skipped - h->skipped_seen;
  h->skipped_seen = skipped;
  return fail(h, SURGE_E_RANGE, std::to_string(n) + " micro-batch(es) carried an aggregate index out of range and were skipped (agg_idx out of range)");
}

int32_t surge_replay_synchronize(surge_replay_handle* h) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  DeviceGuard g(h->device);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return report_skipped_batches(h);
}

int32_t surge_replay_bind_device_csr(surge_replay_handle* h, const int64_t* d_seg_off, int64_t n_agg,
                                     const void* d_events, int64_t n_events, const void* d_init_state,
                                     void* d_state_out) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (n_agg < 0 || n_events < 0) return fail(h, SURGE_E_INVALID, "negative size");
  if (!d_seg_off) return fail(h, SURGE_E_INVALID, "seg_off is NULL");
  if (n_events > 0 && !d_events) return fail(h, SURGE_E_INVALID, "events is NULL");
  if (((uintptr_t)d_events & 15) || ((uintptr_t)d_init_state & 15) || ((uintptr_t)d_state_out & 15) ||
      ((uintptr_t)d_seg_off & 7))
    return fail(h, SURGE_E_INVALID, "device buffers must be 16-byte aligned (seg_off: 8)");
  DeviceGuard g(h->device);
  h->bound = false;
  h->perm_valid = false;
  h->cidx.T = 0;
  h->tidx.T = 0;
  h->tiled_valid = false;
  h->index_timed = h->relayout_timed = false;
  h->index_algo = 0;
  h->d_seg_off = d_seg_off;
  h->d_events = (const uint4*)d_events;
  h->d_init = (const uint4*)d_init_state;
  h->n_agg = n_agg;
  h->n_events = n_events;
  if (d_state_out) {
    h->d_state = (uint4*)d_state_out;
  } else {
    HIPCHK(h, h->own_state.reserve((size_t)n_agg * 64));
    h->d_state = (uint4*)h->own_state.ptr;
  }
  const int32_t rc = analyze_bound(h);
  if (rc != SURGE_OK) return rc;
  h->bound = true;
  h->log_valid = true;
  h->st.n_aggregates = n_agg;
  h->st.n_events = h->an.last - h->an.first;
  h->st.algorithmic_bytes = algorithmic_bytes(h->st.n_events, n_agg, d_init_state != nullptr);
  h->fold_epoch.fetch_add(1);
  return SURGE_OK;
}

int32_t surge_replay_load_csr(surge_replay_handle* h, const int64_t* seg_off, int64_t n_agg, const void* events,
                              int64_t n_events, const void* init_state) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (n_agg < 0 || n_events < 0) return fail(h, SURGE_E_INVALID, "negative size");
  if (!seg_off) return fail(h, SURGE_E_INVALID, "seg_off is NULL");
  if (n_events > 0 && !events) return fail(h, SURGE_E_INVALID, "events is NULL");
  // cheap host-side validation before touching the device
  for (int64_t a = 0; a < n_agg; ++a)
    if (seg_off[a + 1] < seg_off[a]) return fail(h, SURGE_E_INVALID, "seg_off is not monotone non-decreasing");
  if (seg_off[0] < 0 || seg_off[n_agg] > n_events) return fail(h, SURGE_E_INVALID, "seg_off range exceeds the events buffer");
  DeviceGuard g(h->device);
  h->bound = false;
  HIPCHK(h, h->own_seg_off.reserve((size_t)(n_agg + 1) * 8));
  HIPCHK(h, h->own_events.reserve((size_t)n_events * 16));
  if (init_state) HIPCHK(h, h->own_init.reserve((size_t)n_agg * 64));
  HIPCHK(h, hipEventRecord(h->ev_h0, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->own_seg_off.ptr, seg_off, (size_t)(n_agg + 1) * 8, hipMemcpyHostToDevice, h->stream));
  if (n_events > 0)
    HIPCHK(h, hipMemcpyAsync(h->own_events.ptr, events, (size_t)n_events * 16, hipMemcpyHostToDevice, h->stream));
  if (init_state && n_agg > 0)
    HIPCHK(h, hipMemcpyAsync(h->own_init.ptr, init_state, (size_t)n_agg * 64, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_h1, h->stream));
  h->h2d_valid = true;
  return surge_replay_bind_device_csr(h, (const int64_t*)h->own_seg_off.ptr, n_agg, h->own_events.ptr, n_events,
                                      init_state ? h->own_init.ptr : nullptr, nullptr);
}

}  // extern "C" (reopened below)

// ---- which kernel folds the bound log, and the per-log index that kernel needs -------------------------------------
namespace {

struct FoldPlan {
  int32_t use = SURGE_ALGO_FLAT;
  bool uniform = false;
  uint32_t chunk_T = 0;   // CHUNKED / TILED: aggregates longer than this are cut
  int64_t span = 0;
};

int32_t plan_fold(surge_replay_handle* h, int32_t algo, FoldPlan& pl) {
  if (!h->bound) return fail(h, SURGE_E_STATE, "fold before load_csr/bind_device_csr");
  if (!h->log_valid) return fail(h, SURGE_E_STATE, "the resident state was grown past the bound log (surge_replay_grow): load a log again");
  if (algo < SURGE_ALGO_AUTO || algo > SURGE_ALGO_SHORT) return fail(h, SURGE_E_INVALID, "unknown algo");
  if (h->v2 != (algo == SURGE_ALGO_SLOTS) && !(h->v2 && (algo == SURGE_ALGO_AUTO || algo == SURGE_ALGO_TILED)))
    return fail(h, SURGE_E_UNSUPPORTED, h->v2 ? "a v2 slot schema folds with SURGE_ALGO_AUTO / SURGE_ALGO_SLOTS / SURGE_ALGO_TILED only"
                                               : "SURGE_ALGO_SLOTS needs a handle created with surge_replay_create_v2");
  if (h->v2) {
    // one lane per WHOLE aggregate whatever the transport: the tile-major copy is built with nothing cut
    pl.use = algo == SURGE_ALGO_TILED ? SURGE_ALGO_TILED : SURGE_ALGO_SLOTS;
    pl.span = h->an.last - h->an.first;
    pl.chunk_T = 0x7ffffff8u;
    if (pl.use == SURGE_ALGO_TILED && h->an.max_len >= (1ll << 31))
      return fail(h, SURGE_E_UNSUPPORTED, "ALGO_SORTED / ALGO_CHUNKED / ALGO_TILED need segments shorter than 2^31 events");
    return SURGE_OK;
  }
  const int64_t span = h->an.last - h->an.first;
  pl.span = span;
  const bool uniform = h->n_agg > 0 && !h->an.nonuniform && h->an.n_empty == 0 && h->an.len0 > 0 &&
                       (h->an.len0 % 16) == 0 && h->an.len0 < (1ll << 31) && h->an.first == 0;
  pl.uniform = uniform;
  if ((algo == SURGE_ALGO_FIXED || algo == SURGE_ALGO_ROWS) && !uniform)
    return fail(h, SURGE_E_UNSUPPORTED, "ALGO_FIXED / ALGO_ROWS need equal segment lengths that are a multiple of 16");
  const bool rows_ok = uniform && h->an.len0 <= (1 << 20);  // 64 rows x L x 16 B must fit a 31-bit buffer offset
  if (algo == SURGE_ALGO_ROWS && !rows_ok) return fail(h, SURGE_E_UNSUPPORTED, "ALGO_ROWS needs L <= 2^20");
  // one lane per aggregate only pays when 64-aggregate groups alone can fill the chip: measured crossover
  // with FIXED between 512 groups (FIXED 20-50 % faster) and 1024 groups (ROWS 10-18 % faster, L = 64..1024)
  const bool rows_auto = rows_ok && h->n_agg / kWave >= 1024;
  const bool sorted_ok = h->an.max_len < (1ll << 31);
  if ((algo == SURGE_ALGO_SORTED || algo == SURGE_ALGO_CHUNKED || algo == SURGE_ALGO_TILED) && !sorted_ok)
    return fail(h, SURGE_E_UNSUPPORTED, "ALGO_SORTED / ALGO_CHUNKED / ALGO_TILED need segments shorter than 2^31 events");
  // Measured on MI355X (C3: 10 M aggregates, Zipf 1..4096): FLAT 16.2 ms (4.6 TB/s); SORTED (line-aligned
  // 256 B row pieces, 8 resident waves per CU) 12.1-12.7 ms (5.9-6.2 TB/s).  One lane per aggregate pays only when
  // rows are long enough to fill their 256-byte pieces (mean >= 64 events: at <= 32 events per aggregate the
  // lane-per-aggregate kernels measured 2-4x slower than the linear-stream FLAT kernel, at ~64 they tie).
  const double mean_len = h->n_nz > 0 ? (double)span / (double)h->n_nz : 0.0;
  // CHUNKED bounds the critical path: no wave walks more than ~T events alone.  T grows with the log (the longest
  // chunk's walk should stay a small fraction of the kernel; measured optimum on Zipf(1..4096) logs of 2–15 GB with the
  // final kernels: T ~ algorithmic bytes / 4 MB — 0.5 M aggregates 925, 0.8 M 1475, 1.25 M 2320; the optimum is flat
  // to the right and falls off quickly to the left of it) and cut aggregates get at most 256 chunks (the stitch kernel
  // walks them one by one).
  // When T reaches the longest aggregate nothing is cut and the plain sorted-rows kernel runs instead.
  uint32_t chunk_T = 0;
  {
    double t = (double)h->st.algorithmic_bytes / 4.0e6;
    const double t_min = (double)h->an.max_len / 256.0;
    t = t < t_min ? t_min : t;
    t = t < 256.0 ? 256.0 : (t > 65528.0 ? 65528.0 : t);
    chunk_T = ((uint32_t)t + 7u) & ~7u;
    if (const char* v = std::getenv("SURGE_REPLAY_CHUNK_T")) chunk_T = (uint32_t)std::atoi(v);
    chunk_T = chunk_T < 16u ? 16u : (chunk_T > 65528u ? 65528u : chunk_T);
    chunk_T &= ~7u;
  }
  pl.chunk_T = chunk_T;
  const bool nothing_to_cut = (int64_t)chunk_T >= h->an.max_len + 7;
  // one lane per aggregate / chunk pays from ~1.5 GB of log and a mean of 64 events per aggregate (shorter aggregates
  // run 2-4x faster on the linear-stream FLAT kernel; at 0.2 M Zipf aggregates = 1.5 GB CHUNKED and FLAT tie)
  const bool lanes_auto = sorted_ok && mean_len >= 64.0 && (double)h->st.algorithmic_bytes >= 1.5e9;
  // many short rows (a packed events topic whose aggregates published a handful of events each): one lane per row straight from
  // the CSR arrays.  Measured on the e2e topic's packed log (10 M aggregates, 1.4 events each): FLAT 0.29 of 8 TB/s.
  const bool short_auto = h->n_agg >= 65536 && h->an.max_len <= 64 && mean_len < 16.0 && h->an.max_len > 0;
  int32_t use = algo;
  if (algo == SURGE_ALGO_AUTO && short_auto && !uniform) {
    use = SURGE_ALGO_SHORT;
  } else if (algo == SURGE_ALGO_AUTO) {
    // AUTO never picks TILED: the tile-major copy costs about four folds and doubles the log's footprint, which only a
    // caller that replays the bound log repeatedly (or binds it long before it needs the states) wants to pay
    if (uniform)
      use = rows_auto ? SURGE_ALGO_ROWS : SURGE_ALGO_FIXED;
    else if (lanes_auto)
      use = nothing_to_cut ? SURGE_ALGO_SORTED : SURGE_ALGO_CHUNKED;
    else
      use = SURGE_ALGO_FLAT;
  }
  pl.use = use;
  return SURGE_OK;
}

// chunk table of the kernel-facing CSR for chunk target T (align: rows tiled from their 128-byte lines — CHUNKED)
int32_t build_chunk_index(surge_replay_handle* h, surge_replay_handle::ChunkIndex& ci, const int64_t* off, int64_t n_seg, uint32_t T,
                          bool align) {
  // a virtual row is at most T + 7 slots long (an aggregate in one piece: at most T; a chunk: span / c rounded to lines)
  const int64_t max_row = (int64_t)T + 7 < h->an.max_len + 7 ? (int64_t)T + 7 : h->an.max_len + 7;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  // phase 1: the three counts per aggregate and their scans (one allocation: counts, then the scans' block sums)
  const size_t cnt_bytes = up((size_t)(n_seg + 1) * 3 * 8);
  HIPCHK(h, h->ix_cnt.reserve(cnt_bytes + scan_i64_scratch_bytes(n_seg + 1, 3)));
  int64_t* cnt = (int64_t*)h->ix_cnt.ptr;
  HIPCHK(h, launch_chunk_count(off, n_seg, T, align, cnt, (char*)h->ix_cnt.ptr + cnt_bytes, h->stream));
  int64_t totals[3] = {0, 0, 0};  // virtual rows, cut aggregates, side slots
  for (int k = 0; k < 3; ++k)
    HIPCHK(h, hipMemcpyAsync(&totals[k], cnt + (int64_t)k * (n_seg + 1) + n_seg, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  ci.n_vrows = totals[0];
  ci.n_cut_rows = totals[1];
  // phase 2: the table itself (one allocation) ...
  const size_t rows = (size_t)(ci.n_vrows > 0 ? ci.n_vrows : 1), cut = (size_t)(totals[1] > 0 ? totals[1] : 1), slots = (size_t)(totals[2] > 0 ? totals[2] : 1);
  {
    const size_t o_start = 0, o_seg = o_start + up(rows * 8), o_len = o_seg + up(rows * 8), o_info = o_len + up(rows * 4), o_side = o_info + up(rows * 4),
                 o_slot0 = o_side + up(slots * 80), o_out = o_slot0 + up(cut * 8), o_c = o_out + up(cut * 8), total = o_c + up(cut * 4);
    HIPCHK(h, ci.arena.reserve(total));
    char* b = (char*)ci.arena.ptr;
    ci.v_start = b + o_start; ci.v_seg = b + o_seg; ci.v_len = b + o_len; ci.v_info = b + o_info; ci.v_side = b + o_side;
    ci.r_slot0 = b + o_slot0; ci.r_out = b + o_out; ci.r_c = b + o_c;
  }
  // ... and the scratch of its build (one allocation): the sort's, then the rows in aggregate order
  IndexScratch sc;
  void* extra = nullptr;
  const size_t o_ustart = 0, o_udest = o_ustart + up(rows * 8), o_ulen = o_udest + up(rows * 8), o_uinfo = o_ulen + up(rows * 4), u_total = o_uinfo + up(rows * 4);
  {
    const int32_t rcs = index_scratch(h, ci.n_vrows, true, max_row, 0, &sc, u_total, &extra);  // the rows (>= aggregates) are what gets sorted
    if (rcs != SURGE_OK) return rcs;
  }
  char* u = (char*)extra;
  HIPCHK(h, launch_chunk_table(off, n_seg, h->an.n_empty > 0 ? (const int64_t*)h->nz_map.ptr : nullptr, T, align, cnt, ci.n_vrows, sc,
                               (int64_t*)(u + o_ustart), (uint32_t*)(u + o_ulen), (uint32_t*)(u + o_uinfo), (int64_t*)(u + o_udest),
                               (int64_t*)ci.v_start, (uint32_t*)ci.v_len, (uint32_t*)ci.v_info, (int64_t*)ci.v_seg,
                               (int64_t*)ci.r_slot0, (uint32_t*)ci.r_c, (int64_t*)ci.r_out, h->stream));
  ci.T = T;
  return SURGE_OK;
}

// Build (once per bound log) whatever index the chosen kernel needs: the length order (SORTED), the chunk table
// (CHUNKED), the chunk table + the tile-major copy of the log (TILED).  Timed with HIP events; see
// surge_replay_layout_info.
int32_t ensure_index(surge_replay_handle* h, const FoldPlan& pl) {
  if (h->n_agg <= 0 || pl.span <= 0) return SURGE_OK;
  const int64_t* off = h->an.n_empty > 0 ? (const int64_t*)h->nz_off.ptr : h->d_seg_off;
  const int64_t n_seg = h->an.n_empty > 0 ? h->n_nz : h->n_agg;
  if (pl.use == SURGE_ALGO_SORTED && !h->perm_valid) {
    HIPCHK(h, h->perm.reserve((size_t)n_seg * 8));
    IndexScratch sc;
    const int32_t rcs = index_scratch(h, n_seg, false, h->an.max_len, 0, &sc);
    if (rcs != SURGE_OK) return rcs;
    HIPCHK(h, hipEventRecord(h->ev_i0, h->stream));
    HIPCHK(h, launch_sort_by_length(off, n_seg, sc, (int64_t*)h->perm.ptr, h->stream));
    HIPCHK(h, hipEventRecord(h->ev_i1, h->stream));
    h->perm_valid = true;
    h->index_timed = true;
    h->relayout_timed = false;
    h->index_algo = SURGE_ALGO_SORTED;
  } else if (pl.use == SURGE_ALGO_CHUNKED && h->cidx.T != pl.chunk_T) {
    HIPCHK(h, hipEventRecord(h->ev_i0, h->stream));
    const int32_t rc = build_chunk_index(h, h->cidx, off, n_seg, pl.chunk_T, true);
    if (rc != SURGE_OK) return rc;
    HIPCHK(h, hipEventRecord(h->ev_i1, h->stream));
    h->index_timed = true;
    h->relayout_timed = false;
    h->index_algo = SURGE_ALGO_CHUNKED;
  } else if (pl.use == SURGE_ALGO_TILED && (!h->tiled_valid || h->tidx.T != pl.chunk_T)) {
    h->tiled_valid = false;
    HIPCHK(h, hipEventRecord(h->ev_i0, h->stream));
    const int32_t rc = build_chunk_index(h, h->tidx, off, n_seg, pl.chunk_T, false);
    if (rc != SURGE_OK) return rc;
    const int64_t n_groups = (h->tidx.n_vrows + kWave - 1) / kWave;
    HIPCHK(h, h->t_gsub.reserve((size_t)(n_groups + 1) * 8));
    HIPCHK(h, launch_tile_index((const uint32_t*)h->tidx.v_len, h->tidx.n_vrows, (int64_t*)h->t_gsub.ptr, h->stream));
    HIPCHK(h, launch_exclusive_scan_i64((int64_t*)h->t_gsub.ptr, n_groups, h->stream));
    int64_t n_sub = 0;
    HIPCHK(h, hipMemcpyAsync(&n_sub, (int64_t*)h->t_gsub.ptr + n_groups, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipEventRecord(h->ev_i1, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->t_n_sub = n_sub;
    HIPCHK(h, h->t_tiles.reserve((size_t)n_sub * kTileSubBytes));
    HIPCHK(h, hipEventRecord(h->ev_r0, h->stream));
    HIPCHK(h, launch_relayout(h->d_events, (const int64_t*)h->tidx.v_start, (const uint32_t*)h->tidx.v_len, h->tidx.n_vrows,
                              (const int64_t*)h->t_gsub.ptr, n_sub, (uint4*)h->t_tiles.ptr, h->stream));
    HIPCHK(h, hipEventRecord(h->ev_r1, h->stream));
    if (std::getenv("SURGE_DBG_PRINT"))
      std::fprintf(stderr, "[surge dbg] tiles %p (%lld subtiles) v_len %p v_info %p v_dest %p g_sub %p state %p events %p\n", h->t_tiles.ptr,
                   (long long)n_sub, h->tidx.v_len, h->tidx.v_info, h->tidx.v_seg, h->t_gsub.ptr, (void*)h->d_state, (const void*)h->d_events);
    h->tiled_valid = true;
    h->index_timed = h->relayout_timed = true;
    h->index_algo = SURGE_ALGO_TILED;
  } else {
    return SURGE_OK;  // nothing was built
  }
  index_scratch_release(h);  // (hipFree waits for the build's kernels)
  return SURGE_OK;
}

// v2 over the tile-major copy (rows = whole aggregates, never cut): same launch shape as the v1 tiled fold
int32_t run_slots_tiled(surge_replay_handle* h, FoldParams& p) {
  int subs = 2;
  if (const char* v = std::getenv("SURGE_REPLAY_TILED_SUBS")) subs = std::atoi(v) == 1 ? 1 : 2;
  const auto& ci = h->tidx;
  {
    const int32_t rcd = dispenser_begin(h, p);
    if (rcd != SURGE_OK) return rcd;
  }
  p.n_seg = h->an.n_empty > 0 ? h->n_nz : h->n_agg;
  const int64_t groups = (ci.n_vrows + kWave - 1) / kWave;
  int64_t per_cu = subs == 1 ? 8 : 6;
  if (const char* v = std::getenv("SURGE_REPLAY_TILED_WAVES")) per_cu = std::atoi(v) > 0 ? std::atoi(v) : per_cu;
  const int64_t slots = (int64_t)h->n_cus * per_cu;
  const int64_t n_waves = groups < slots ? groups : slots;
  TileTable t;
  t.tiles = (const uint4*)h->t_tiles.ptr; t.g_sub0 = (const int64_t*)h->t_gsub.ptr; t.v_len = (const uint32_t*)ci.v_len;
  t.v_info = (const uint32_t*)ci.v_info; t.v_dest = (const int64_t*)ci.v_seg; t.n_vrows = ci.n_vrows; t.side = nullptr;
  hipEvent_t e0, e1;
  const int32_t rc = next_fold_events(h, &e0, &e1);
  if (rc != SURGE_OK) return rc;
  HIPCHK(h, hipEventRecord(e0, h->stream));
  HIPCHK(h, launch_fold_slots_tiled(p, *(const SlotParams*)h->slot_params, h->spec, t, n_waves, subs, h->stream));
  HIPCHK(h, hipEventRecord(e1, h->stream));
  h->st.n_tasks = (int32_t)n_waves;
  return SURGE_OK;
}

}  // namespace

extern "C" {

int32_t surge_replay_kernel_info(surge_replay_handle* h, surge_replay_kernel_info_t* out) {
  if (!h || !out) return fail(h, SURGE_E_INVALID, "NULL argument");
  std::memset(out, 0, sizeof(*out));
  if (!h->v2) {  // v1: the flat kernel (K3 appends, AUTO on logs of few long rows) is the one compiled per op table
    DeviceGuard g(h->device);
    FoldParams p;
    fill_params(h, p);
    (void)flat_spec(h, p);
  }
  out->specialised = h->v2 ? (h->spec ? 1 : 0) : (h->spec1 ? 1 : 0);
  out->compile_ms = h->v2 ? h->spec_compile_ms : h->spec1_compile_ms;
  std::string d = h->v2 ? h->spec_why
                        : (h->spec1 ? h->spec1_why : "v1 schema, ahead-of-time kernels interpret the op table: " + h->spec1_why);
  if (!h->v2 && h->lanes1_tried) d = "lane kernels " + (h->lanes1 ? h->lanes1_why : "ahead of time (" + h->lanes1_why.substr(0, 60) + ")") + "; " + d;
  std::snprintf(out->detail, sizeof(out->detail), "%s", d.c_str());
  return SURGE_OK;
}

int32_t surge_replay_compile_schema(const surge_replay_schema* schema, const char* arch, void* code_out, int64_t capacity, int64_t* code_bytes) {
  if (!schema || !arch || !code_bytes) return fail(nullptr, SURGE_E_INVALID, "NULL argument");
  *code_bytes = 0;
  {
    const int32_t rc = validate_schema(schema);
    if (rc != SURGE_OK) return rc;
  }
  FoldParams p;
  fill_params(*schema, p);
  const std::string src = v1_spec_source(p.table, V1_FLAT);
  if (src.empty()) return fail(nullptr, SURGE_E_UNSUPPORTED, "the op table holds words the specialised build cannot express");
  std::vector<char> code;
  std::string log;
  double ms = 0.0;
  if (const char* v = std::getenv("SURGE_REPLAY_RTC_LANES")) {
    if (std::atoi(v) != 0) {  // the lane kernels' program too (its code object goes to the disk cache, not to the caller)
      std::vector<char> lanes;
      if (!rtc_compile(v1_spec_source(p.table, V1_LANES), arch, &lanes, &log, &ms)) return fail(nullptr, SURGE_E_UNSUPPORTED, log);
    }
  }
  if (!rtc_compile(src, arch, &code, &log, &ms)) return fail(nullptr, SURGE_E_UNSUPPORTED, log);
  *code_bytes = (int64_t)code.size();
  if (code_out) {
    if (capacity < (int64_t)code.size()) return fail(nullptr, SURGE_E_INVALID, "code_out is too small (see *code_bytes)");
    std::memcpy(code_out, code.data(), code.size());
  }
  return SURGE_OK;
}

int32_t surge_replay_compile_schema_v2(const surge_replay_schema_v2* sc, const char* arch, void* code_out, int64_t capacity,
                                       int64_t* code_bytes) {
  if (!sc || !arch || !code_bytes) return fail(nullptr, SURGE_E_INVALID, "NULL argument");
  *code_bytes = 0;
  {
    const int32_t rc = validate_schema_v2(sc);
    if (rc != SURGE_OK) return rc;
  }
  alignas(16) unsigned char spb[kSlotParamsBytes] = {};
  slot_params_from_schema(*sc, (SlotParams*)spb);
  std::vector<char> code;
  std::string log;
  double ms = 0.0;
  if (!rtc_compile(slots_spec_source(*(const SlotParams*)spb), arch, &code, &log, &ms)) return fail(nullptr, SURGE_E_UNSUPPORTED, log);
  *code_bytes = (int64_t)code.size();
  if (code_out) {
    if (capacity < (int64_t)code.size()) return fail(nullptr, SURGE_E_INVALID, "code_out is too small (see *code_bytes)");
    std::memcpy(code_out, code.data(), code.size());
  }
  return SURGE_OK;
}

int32_t surge_replay_prepare(surge_replay_handle* h, int32_t algo) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  DeviceGuard g(h->device);
  FoldPlan pl;
  const int32_t rc = plan_fold(h, algo, pl);
  if (rc != SURGE_OK) return rc;
  if (h->v2 && pl.use != SURGE_ALGO_TILED) return SURGE_OK;  // the slot kernel's length order is built by its first fold
  if (!h->v2 && (pl.use == SURGE_ALGO_SORTED || pl.use == SURGE_ALGO_CHUNKED || pl.use == SURGE_ALGO_ROWS)) {
    FoldParams p;
    fill_params(h, p);
    (void)lane_spec(h, p);  // the kernels for this op table (hiprtc, or the code-object cache on disk)
  }
  return ensure_index(h, pl);
}

int32_t surge_replay_layout_info(surge_replay_handle* h, surge_replay_layout_info_t* out) {
  if (!h || !out) return fail(h, SURGE_E_INVALID, "NULL argument");
  if (!h->bound) return fail(h, SURGE_E_STATE, "layout_info before load_csr/bind_device_csr");
  DeviceGuard g(h->device);
  std::memset(out, 0, sizeof(*out));
  out->algo = h->index_algo;
  if (h->index_algo == SURGE_ALGO_CHUNKED) {
    out->virtual_rows = h->cidx.n_vrows;
    out->cut_aggregates = h->cidx.n_cut_rows;
    out->chunk_events = h->cidx.T;
  } else if (h->index_algo == SURGE_ALGO_TILED) {
    out->virtual_rows = h->tidx.n_vrows;
    out->cut_aggregates = h->tidx.n_cut_rows;
    out->chunk_events = h->tidx.T;
    out->tiled_bytes = h->t_n_sub * kTileSubBytes;
    out->padding_events = h->t_n_sub * (kTileSubBytes / 16) - (h->an.last - h->an.first);
  } else if (h->index_algo == SURGE_ALGO_SORTED) {
    out->virtual_rows = h->an.n_empty > 0 ? h->n_nz : h->n_agg;
  }
  if (h->index_timed) {
    HIPCHK(h, hipEventSynchronize(h->ev_i1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev_i0, h->ev_i1));
    out->index_build_ms = ms;
  }
  if (h->relayout_timed) {
    HIPCHK(h, hipEventSynchronize(h->ev_r1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev_r0, h->ev_r1));
    out->relayout_ms = ms;
  }
  return SURGE_OK;
}

int32_t surge_replay_index_order(surge_replay_handle* h, int32_t algo, int64_t* order_out, int64_t capacity, int64_t* n_out) {
  if (!h || !n_out) return fail(h, SURGE_E_INVALID, "NULL argument");
  *n_out = 0;
  if (!h->bound) return fail(h, SURGE_E_STATE, "index_order before load_csr/bind_device_csr");
  if (capacity < 0 || (capacity > 0 && !order_out)) return fail(h, SURGE_E_INVALID, "bad capacity / buffer");
  DeviceGuard g(h->device);
  const void* src = nullptr;
  int64_t n = 0;
  if (algo == SURGE_ALGO_SORTED && h->perm_valid && !h->v2) {
    src = h->perm.ptr;
    n = h->an.n_empty > 0 ? h->n_nz : h->n_agg;
  } else if (algo == SURGE_ALGO_CHUNKED && h->cidx.T != 0) {
    src = h->cidx.v_start;
    n = h->cidx.n_vrows;
  } else {
    return fail(h, SURGE_E_STATE, "the bound log has no index of that kind (surge_replay_prepare / fold with SURGE_ALGO_SORTED or _CHUNKED first)");
  }
  *n_out = n;
  const int64_t take = n < capacity ? n : capacity;
  if (take > 0) {
    HIPCHK(h, hipMemcpyAsync(order_out, src, (size_t)take * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return SURGE_OK;
}

int32_t surge_replay_fold(surge_replay_handle* h, int32_t algo) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  DeviceGuard g(h->device);
  FoldPlan pl;
  {
    const int32_t rc = plan_fold(h, algo, pl);
    if (rc != SURGE_OK) return rc;
  }
  if (h->v2) {
    if (pl.use == SURGE_ALGO_TILED) {
      const int32_t rc = ensure_index(h, pl);
      if (rc != SURGE_OK) return rc;
    }
    return fold_slots_bound(h, pl.use == SURGE_ALGO_TILED);
  }
  const int64_t span = pl.span;
  const int32_t use = pl.use;
  {
    const int32_t rc = ensure_index(h, pl);  // once per bound log (part of its index, like the empty-segment compaction)
    if (rc != SURGE_OK) return rc;
  }

  FoldParams p;
  fill_params(h, p);
  p.events = h->d_events;
  p.n_events = h->n_events;
  p.init = h->d_init;
  p.out = h->d_state;

  HIPCHK(h, hipEventRecord(h->ev_total0, h->stream));
  h->st.n_tasks = 0;
  if (h->n_agg > 0 && span > 0) {
    if (use == SURGE_ALGO_ROWS) {
      const int le = env_lane_events("SURGE_REPLAY_LE_ROWS", 8);
      const int64_t L = h->an.len0;
      // a task = G groups of 64 aggregates, about kTaskBytes of events
      // Measured on MI355X: this access pattern runs fastest as ONE resident generation of waves (no
      // re-dispatch, every wave streams from start to end): G groups of 64 aggregates per wave so that
      // the grid just fits the chip's wave slots (CUs x 16 waves at 8 KiB tiles, x 9 at 16 KiB tiles).
      const int64_t groups = (h->n_agg + kWave - 1) / kWave;
      const int64_t slots = (int64_t)h->n_cus * (le == 8 ? 16 : (le == 16 ? 9 : 4));
      int64_t G = (groups + slots - 1) / slots;
      if (const char* v = std::getenv("SURGE_REPLAY_ROWS_GROUPS")) G = std::atoi(v);
      if (G < 1) G = 1;
      const int64_t per_task = G * kWave;
      const int64_t n_tasks = (h->n_agg + per_task - 1) / per_task;
      p.n_seg = h->n_agg;
      p.fixed_len = L;
      p.segs_per_task = per_task;
      hipEvent_t e0, e1;
      const int32_t rc = next_fold_events(h, &e0, &e1);
      if (rc != SURGE_OK) return rc;
      const V1Kernels* lanes = lane_spec(h, p);
      HIPCHK(h, hipEventRecord(e0, h->stream));
      HIPCHK(h, launch_fold_rows(p, lanes, n_tasks, le, h->stream));
      HIPCHK(h, hipEventRecord(e1, h->stream));
      h->st.n_tasks = (int32_t)n_tasks;
    } else if (use == SURGE_ALGO_FIXED) {
      const int le = env_lane_events("SURGE_REPLAY_LE_FIXED", 16);
      const int64_t L = h->an.len0;
      const int64_t task_events = choose_task_events(span, le);
      int64_t G = task_events / L;
      if (G < 1) G = 1;
      const int64_t n_tasks = (h->n_agg + G - 1) / G;
      p.n_seg = h->n_agg;
      p.fixed_len = L;
      p.segs_per_task = G;
      hipEvent_t e0, e1;
      const int32_t rc = next_fold_events(h, &e0, &e1);
      if (rc != SURGE_OK) return rc;
      HIPCHK(h, hipEventRecord(e0, h->stream));
      HIPCHK(h, launch_fold_fixed(p, n_tasks, le, h->stream));
      HIPCHK(h, hipEventRecord(e1, h->stream));
      h->st.n_tasks = (int32_t)n_tasks;
    } else if (use == SURGE_ALGO_SHORT) {
      p.seg_off = h->d_seg_off;  // every aggregate is a row, the empty ones too
      p.n_seg = h->n_agg;
      hipEvent_t e0, e1;
      const int32_t rc = next_fold_events(h, &e0, &e1);
      if (rc != SURGE_OK) return rc;
      HIPCHK(h, hipEventRecord(e0, h->stream));
      HIPCHK(h, launch_fold_short(p, h->stream));
      HIPCHK(h, hipEventRecord(e1, h->stream));
      h->st.n_tasks = (int32_t)((h->n_agg + 63) / 64);
    } else if (use == SURGE_ALGO_SORTED) {
      const int le = env_lane_events("SURGE_REPLAY_LE_SORTED", 16);
      const int64_t* off = h->an.n_empty > 0 ? (const int64_t*)h->nz_off.ptr : h->d_seg_off;
      const int64_t n_seg = h->an.n_empty > 0 ? h->n_nz : h->n_agg;
      if (h->an.n_empty > 0) p.out_map = (const int64_t*)h->nz_map.ptr;
      p.seg_off = off;
      p.plan = (const int64_t*)h->perm.ptr;
      {
        const int32_t rcd = dispenser_begin(h, p);
        if (rcd != SURGE_OK) return rcd;
      }
      p.n_seg = n_seg;
      const int64_t groups = (n_seg + kWave - 1) / kWave;
      // resident waves per CU = min(LDS, registers): 8 KiB tiles 12 (136 VGPRs), 16 KiB tiles 8 (18.6 KB LDS), 32 KiB tiles 4
      int64_t per_cu = le == 8 ? 12 : (le == 16 ? 8 : 4);
      if (const char* v = std::getenv("SURGE_REPLAY_SORTED_WAVES")) per_cu = std::atoi(v) > 0 ? std::atoi(v) : per_cu;  // (experiments)
      const int64_t slots = (int64_t)h->n_cus * per_cu;
      const int64_t n_waves = groups < slots ? groups : slots;
      hipEvent_t e0, e1;
      const int32_t rc = next_fold_events(h, &e0, &e1);
      if (rc != SURGE_OK) return rc;
      const V1Kernels* lanes = lane_spec(h, p);
      HIPCHK(h, hipEventRecord(e0, h->stream));
      // round 5: the walk that fetches the next group's first tile during this group's last one (fold_sorted_pf_kernel);
      // SURGE_REPLAY_SORTED_KERNEL=plain keeps fold_sorted_kernel for a same-box comparison, and 32-event lanes are its only
      static const bool plain = [] { const char* v = std::getenv("SURGE_REPLAY_SORTED_KERNEL"); return v && std::strcmp(v, "plain") == 0; }();
      if (plain || (le == 32 && !lanes)) HIPCHK(h, launch_fold_sorted(p, n_waves, le, h->stream));
      else HIPCHK(h, launch_fold_sorted_pf(p, lanes, n_waves, le, h->stream));
      HIPCHK(h, hipEventRecord(e1, h->stream));
      h->st.n_tasks = (int32_t)n_waves;
    } else if (use == SURGE_ALGO_CHUNKED) {
      const int le = env_lane_events("SURGE_REPLAY_LE_CHUNKED", 16) == 8 ? 8 : 16;
      const auto& ci = h->cidx;
      {
        const int32_t rcd = dispenser_begin(h, p);
        if (rcd != SURGE_OK) return rcd;
      }
      p.n_seg = h->an.n_empty > 0 ? h->n_nz : h->n_agg;
      const int64_t groups = (ci.n_vrows + kWave - 1) / kWave;
      // resident waves per CU: 16 KiB tiles 8 (2 per SIMD, 8 x 18.7 KB of LDS), 8 KiB tiles 12 (3 per SIMD)
      const int64_t slots = (int64_t)h->n_cus * (le == 8 ? 12 : 8);
      const int64_t n_waves = groups < slots ? groups : slots;
      hipEvent_t e0, e1;
      const int32_t rc = next_fold_events(h, &e0, &e1);
      if (rc != SURGE_OK) return rc;
      const V1Kernels* lanes = lane_spec(h, p);
      HIPCHK(h, hipEventRecord(e0, h->stream));  // the stitch kernel is timed with the fold: it is part of it
      HIPCHK(h, launch_fold_chunked(p, (const int64_t*)ci.v_start, (const uint32_t*)ci.v_len, (const uint32_t*)ci.v_info,
                                    (const int64_t*)ci.v_seg, ci.n_vrows, (uint32_t*)ci.v_side, (const int64_t*)ci.r_slot0,
                                    (const uint32_t*)ci.r_c, (const int64_t*)ci.r_out, ci.n_cut_rows, lanes, n_waves, le, h->stream));
      HIPCHK(h, hipEventRecord(e1, h->stream));
      h->st.n_tasks = (int32_t)n_waves;
    } else if (use == SURGE_ALGO_TILED) {
      int subs = 2;  // subtiles (8 events per lane) per step: 16 KiB in flight per wave
      if (const char* v = std::getenv("SURGE_REPLAY_TILED_SUBS")) subs = std::atoi(v) == 1 ? 1 : 2;
      const auto& ci = h->tidx;
      {
        const int32_t rcd = dispenser_begin(h, p);
        if (rcd != SURGE_OK) return rcd;
      }
      p.n_seg = h->an.n_empty > 0 ? h->n_nz : h->n_agg;
      const int64_t groups = (ci.n_vrows + kWave - 1) / kWave;
      // resident waves per CU.  Measured (round 3, same handle, Zipf(1..4096) logs of 9 / 30 / 74 GB and config C2): with
      // 16 KiB steps 6 waves per CU beat 8 and 9 by 0.3-7 % and 4 by 0-5 %; 8 KiB steps are 0.5-4 % behind at any count
      int64_t per_cu = subs == 1 ? 8 : 6;
      if (const char* v = std::getenv("SURGE_REPLAY_TILED_WAVES")) per_cu = std::atoi(v) > 0 ? std::atoi(v) : per_cu;
      const int64_t slots = (int64_t)h->n_cus * per_cu;
      const int64_t n_waves = groups < slots ? groups : slots;
      hipEvent_t e0, e1;
      const int32_t rc = next_fold_events(h, &e0, &e1);
      if (rc != SURGE_OK) return rc;
      HIPCHK(h, hipEventRecord(e0, h->stream));  // the stitch kernel is timed with the fold: it is part of it
      HIPCHK(h, launch_fold_tiled(p, (const uint4*)h->t_tiles.ptr, (const int64_t*)h->t_gsub.ptr,
                                  (const uint32_t*)ci.v_len, (const uint32_t*)ci.v_info, (const int64_t*)ci.v_seg, ci.n_vrows,
                                  (uint32_t*)ci.v_side, n_waves, subs, h->stream));
      HIPCHK(h, launch_chunk_stitch(p, (const uint32_t*)ci.v_side, (const int64_t*)ci.r_slot0, (const uint32_t*)ci.r_c,
                                    (const int64_t*)ci.r_out, ci.n_cut_rows, h->stream));
      HIPCHK(h, hipEventRecord(e1, h->stream));
      h->st.n_tasks = (int32_t)n_waves;
    } else if (h->an.n_empty > 0) {
      p.out_map = (const int64_t*)h->nz_map.ptr;
      const int32_t rc = run_flat(h, p, (const int64_t*)h->nz_off.ptr, h->n_nz, span);
      if (rc != SURGE_OK) return rc;
    } else {
      const int32_t rc = run_flat(h, p, h->d_seg_off, h->n_agg, span);
      if (rc != SURGE_OK) return rc;
    }
  } else {
    hipEvent_t e0, e1;
    const int32_t rc = next_fold_events(h, &e0, &e1);
    if (rc != SURGE_OK) return rc;
    HIPCHK(h, hipEventRecord(e0, h->stream));
    HIPCHK(h, hipEventRecord(e1, h->stream));
  }
  if ((h->an.n_empty > 0 && !(use == SURGE_ALGO_SHORT && span > 0)) || span == 0)
    HIPCHK(h, launch_fill_empty(h->d_seg_off, h->n_agg, h->d_init, h->d_state, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_total1, h->stream));
  h->timing_valid = true;
  h->st.last_algo = use;
  h->st.n_folds += 1;
  h->st.n_poisoned = -1;
  h->fold_epoch.fetch_add(1);
  return SURGE_OK;
}

int32_t surge_replay_append_fold_device(surge_replay_handle* h, const int64_t* d_group_agg, const int64_t* d_group_off,
                                        int64_t n_groups, const void* d_events, int64_t n_events) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound) return fail(h, SURGE_E_STATE, "append_fold before load_csr/bind_device_csr");
  if (n_groups < 0 || n_events < 0) return fail(h, SURGE_E_INVALID, "negative size");
  if (n_groups == 0 || n_events == 0) return SURGE_OK;
  if (!d_group_agg || !d_group_off || !d_events) return fail(h, SURGE_E_INVALID, "NULL batch buffer");
  if ((uintptr_t)d_events & 15) return fail(h, SURGE_E_INVALID, "events must be 16-byte aligned");
  DeviceGuard g(h->device);
  FoldParams p;
  fill_params(h, p);
  p.events = (const uint4*)d_events;
  p.n_events = n_events;
  p.init = h->d_state;  // fold onto the resident state, in place
  p.out = h->d_state;
  p.out_map = d_group_agg;
  HIPCHK(h, hipEventRecord(h->ev_total0, h->stream));
  if (h->v2) std::memset(p.table, 0, sizeof(p.table));
  const int32_t rc = h->v2 ? run_slots(h, p, d_group_off, n_groups, false) : run_flat(h, p, d_group_off, n_groups, n_events);
  if (rc != SURGE_OK) return rc;
  if (h->v2) h->perm_valid = false;  // the length order of the micro-batch replaced the bound log's
  HIPCHK(h, hipEventRecord(h->ev_total1, h->stream));
  h->timing_valid = true;
  h->st.last_algo = h->v2 ? SURGE_ALGO_SLOTS : SURGE_ALGO_FLAT;
  h->st.n_folds += 1;
  h->st.n_poisoned = -1;
  h->fold_epoch.fetch_add(1);
  return SURGE_OK;
}

int32_t surge_replay_append_fold(surge_replay_handle* h, const int64_t* group_agg, const int64_t* group_off,
                                 int64_t n_groups, const void* events, int64_t n_events) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound) return fail(h, SURGE_E_STATE, "append_fold before load_csr/bind_device_csr");
  if (n_groups < 0 || n_events < 0) return fail(h, SURGE_E_INVALID, "negative size");
  if (n_groups == 0 || n_events == 0) return SURGE_OK;
  if (!group_agg || !group_off || !events) return fail(h, SURGE_E_INVALID, "NULL batch buffer");
  if (group_off[0] != 0 || group_off[n_groups] != n_events)
    return fail(h, SURGE_E_INVALID, "group_off must span [0, n_events]");
  for (int64_t gidx = 0; gidx < n_groups; ++gidx) {
    if (group_off[gidx + 1] <= group_off[gidx]) return fail(h, SURGE_E_INVALID, "batch groups must be non-empty and ordered");
    if (group_agg[gidx] < 0 || group_agg[gidx] >= h->n_agg) return fail(h, SURGE_E_RANGE, "group_agg out of range");
  }
  try {
    // an aggregate may appear in one group only: two groups would race on the same resident state
    std::vector<int64_t> seen(group_agg, group_agg + n_groups);
    std::sort(seen.begin(), seen.end());
    if (std::adjacent_find(seen.begin(), seen.end()) != seen.end())
      return fail(h, SURGE_E_INVALID, "an aggregate appears in more than one group of the batch");
  } catch (const std::bad_alloc&) {
    return fail(h, SURGE_E_NOMEM, "out of host memory while validating the micro-batch");
  }
  DeviceGuard g(h->device);
  HIPCHK(h, h->batch_group_agg.reserve_roomy((size_t)n_groups * 8));
  HIPCHK(h, h->batch_group_off.reserve_roomy((size_t)(n_groups + 1) * 8));
  HIPCHK(h, h->batch_events.reserve_roomy((size_t)n_events * 16));
  HIPCHK(h, hipEventRecord(h->ev_h0, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->batch_group_agg.ptr, group_agg, (size_t)n_groups * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->batch_group_off.ptr, group_off, (size_t)(n_groups + 1) * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->batch_events.ptr, events, (size_t)n_events * 16, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_h1, h->stream));
  h->h2d_valid = true;
  return surge_replay_append_fold_device(h, (const int64_t*)h->batch_group_agg.ptr, (const int64_t*)h->batch_group_off.ptr,
                                         n_groups, h->batch_events.ptr, n_events);
}

int32_t surge_replay_append_events_device(surge_replay_handle* h, const int64_t* d_agg_idx, const void* d_events, int64_t n_events) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound) return fail(h, SURGE_E_STATE, "append_events before load_csr/bind_device_csr");
  if (n_events < 0) return fail(h, SURGE_E_INVALID, "negative size");
  if (n_events == 0) return SURGE_OK;
  if (!d_agg_idx || !d_events) return fail(h, SURGE_E_INVALID, "NULL batch buffer");
  if (n_events > 0xffffffffll) return fail(h, SURGE_E_UNSUPPORTED, "micro-batches are limited to 2^32 - 1 events");
  if (h->n_agg > 0xffffffffll) return fail(h, SURGE_E_UNSUPPORTED, "the device group-by needs fewer than 2^32 aggregates");
  if ((uintptr_t)d_events & 15) return fail(h, SURGE_E_INVALID, "events must be 16-byte aligned");
  DeviceGuard g(h->device);
  const uint32_t n = (uint32_t)n_events;
  unsigned bits = 1;
  while (bits < 32 && (h->n_agg >> bits) != 0) ++bits;
  size_t temp = 0;
  HIPCHK(h, groupby_temp_bytes(n, bits, &temp));
  HIPCHK(h, h->gb_temp.reserve_roomy(temp));
  HIPCHK(h, h->gb_u32.reserve_roomy((size_t)n * 4 * 6));
  HIPCHK(h, h->gb_flags.reserve(16));
  HIPCHK(h, h->batch_group_agg.reserve_roomy((size_t)n * 8));
  HIPCHK(h, h->batch_group_off.reserve_roomy((size_t)(n + 1) * 8));
  HIPCHK(h, h->batch_events.reserve_roomy((size_t)n * 16));
  uint32_t* u = (uint32_t*)h->gb_u32.ptr;
  if (!h->host_flags) {
    HIPCHK(h, hipHostMalloc((void**)&h->host_flags, 16, hipHostMallocDefault));
    std::memset(h->host_flags, 0, 16);
    HIPCHK(h, hipMemsetAsync(h->gb_flags.ptr, 0, 16, h->stream));  // the sticky "skipped batches" word starts at 0
  }
  HIPCHK(h, launch_groupby(d_agg_idx, (const uint4*)d_events, n, h->n_agg, bits, h->gb_temp.ptr, temp, u, u + n, u + 2 * (size_t)n,
                           u + 3 * (size_t)n, u + 4 * (size_t)n, u + 5 * (size_t)n, (uint4*)h->batch_events.ptr,
                           (int64_t*)h->batch_group_agg.ptr, (int64_t*)h->batch_group_off.ptr, (uint32_t*)h->gb_flags.ptr, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->host_flags, h->gb_flags.ptr, 12, hipMemcpyDeviceToHost, h->stream));
  if (h->v2) {
    // the slot kernel's launch (length sort of the groups, one lane per group) is sized on the host: wait for the count
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int32_t rs = report_skipped_batches(h);
    if (rs != SURGE_OK) return rs;
    if (h->host_flags[0] == 0u) return SURGE_OK;
    return surge_replay_append_fold_device(h, (const int64_t*)h->batch_group_agg.ptr, (const int64_t*)h->batch_group_off.ptr,
                                           (int64_t)h->host_flags[0], h->batch_events.ptr, n_events);
  }
  // v1: no host round trip — the plan kernel reads the group count where the group-by left it, the fold's grid depends on
  // the event count only, and a batch with a bad index has zero groups (reported at the next synchronisation point)
  FoldParams p;
  fill_params(h, p);
  p.events = (const uint4*)h->batch_events.ptr;
  p.n_events = n_events;
  p.init = h->d_state;  // fold onto the resident state, in place
  p.out = h->d_state;
  p.out_map = (const int64_t*)h->batch_group_agg.ptr;
  HIPCHK(h, hipEventRecord(h->ev_total0, h->stream));
  {
    const int le = env_lane_events("SURGE_REPLAY_LE_FLAT", 16);
    const int64_t task_events = choose_task_events(n_events, le);
    const int64_t n_tasks = (n_events + task_events - 1) / task_events;
    HIPCHK(h, h->plan.reserve_roomy((size_t)(n_tasks + 1) * 8));
    HIPCHK(h, launch_plan_dev((const int64_t*)h->batch_group_off.ptr, (const uint32_t*)h->gb_flags.ptr, task_events, n_tasks,
                              (int64_t*)h->plan.ptr, h->stream));
    p.seg_off = (const int64_t*)h->batch_group_off.ptr;
    p.plan = (const int64_t*)h->plan.ptr;
    p.n_seg = 0;  // FLAT takes its segments from the plan
    hipEvent_t e0, e1;
    const int32_t rc = next_fold_events(h, &e0, &e1);
    if (rc != SURGE_OK) return rc;
    const V1Kernels* spec = flat_spec(h, p);
    HIPCHK(h, hipEventRecord(e0, h->stream));
    HIPCHK(h, launch_fold_flat(p, spec, n_tasks, le, h->stream));
    HIPCHK(h, hipEventRecord(e1, h->stream));
    h->st.n_tasks = (int32_t)n_tasks;
  }
  HIPCHK(h, hipEventRecord(h->ev_total1, h->stream));
  h->timing_valid = true;
  h->st.last_algo = SURGE_ALGO_FLAT;
  h->st.n_folds += 1;
  h->st.n_poisoned = -1;
  h->fold_epoch.fetch_add(1);
  return SURGE_OK;
}

int32_t surge_replay_append_events(surge_replay_handle* h, const int64_t* agg_idx, const void* events, int64_t n_events) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound) return fail(h, SURGE_E_STATE, "append_events before load_csr/bind_device_csr");
  if (n_events < 0) return fail(h, SURGE_E_INVALID, "negative size");
  if (n_events == 0) return SURGE_OK;
  if (!agg_idx || !events) return fail(h, SURGE_E_INVALID, "NULL batch buffer");
  if (n_events > 0xffffffffll) return fail(h, SURGE_E_UNSUPPORTED, "micro-batches are limited to 2^32 - 1 events");
  DeviceGuard g(h->device);
  // the indices are on the host here: check them before anything is enqueued (immediate SURGE_E_RANGE, batch not applied)
  for (int64_t i = 0; i < n_events; ++i)
    if (agg_idx[i] < 0 || agg_idx[i] >= h->n_agg) return fail(h, SURGE_E_RANGE, "agg_idx out of range");
  // Host buffers (pageable: a JNI direct buffer, a numpy array) go through pinned staging so the H2D copy runs at PCIe
  // speed; two staging areas alternate, so filling the next batch overlaps the copy and the fold of the previous one and
  // the host never waits for the whole stream (double-buffered H2D).  Grouping happens on the device.
  const size_t need = (size_t)n_events * 24;
  const int k = h->pinned_next;
  h->pinned_next ^= 1;
  if (!h->ev_staged[k]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_staged[k], hipEventDisableTiming));
  if (h->staged_busy[k]) {  // the copies out of this area (two batches ago) must be done before it is overwritten
    HIPCHK(h, hipEventSynchronize(h->ev_staged[k]));
    h->staged_busy[k] = false;
  }
  if (need > h->pinned_cap[k]) {
    if (h->pinned[k]) (void)hipHostFree(h->pinned[k]);
    h->pinned[k] = nullptr;
    h->pinned_cap[k] = 0;
    const size_t cap = need < (4u << 20) ? (4u << 20) : need + need / 2;
    HIPCHK(h, hipHostMalloc(&h->pinned[k], cap, hipHostMallocDefault));
    h->pinned_cap[k] = cap;
  }
  // the device-side landing buffers are reused by every batch: stream order keeps a batch's copies behind the previous
  // batch's kernels; growing them must wait for those kernels
  if ((size_t)n_events * 8 > h->gb_agg_idx.cap || (size_t)n_events * 16 > h->gb_events.cap) HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, h->gb_agg_idx.reserve_roomy((size_t)n_events * 8));
  HIPCHK(h, h->gb_events.reserve_roomy((size_t)n_events * 16));
  std::memcpy(h->pinned[k], agg_idx, (size_t)n_events * 8);
  std::memcpy((char*)h->pinned[k] + (size_t)n_events * 8, events, (size_t)n_events * 16);
  HIPCHK(h, hipEventRecord(h->ev_h0, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->gb_agg_idx.ptr, h->pinned[k], (size_t)n_events * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->gb_events.ptr, (char*)h->pinned[k] + (size_t)n_events * 8, (size_t)n_events * 16, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_h1, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_staged[k], h->stream));
  h->staged_busy[k] = true;
  h->h2d_valid = true;
  return surge_replay_append_events_device(h, (const int64_t*)h->gb_agg_idx.ptr, h->gb_events.ptr, n_events);
}

static int32_t stage_grow(surge_replay_handle* h, int64_t want) {
  if (want <= h->stage_cap) return SURGE_OK;
  if (want > 0xffffffffll) return fail(h, SURGE_E_UNSUPPORTED, "the staging log holds fewer than 2^32 events per pack");
  int64_t cap = h->stage_cap * 2 > want ? h->stage_cap * 2 : want;
  cap = cap < (1 << 16) ? (1 << 16) : (cap > 0xffffffffll ? 0xffffffffll : cap);
  void *nk = nullptr, *ne = nullptr;
  HIPCHK(h, hipMalloc(&nk, (size_t)cap * 4));
  hipError_t e = hipMalloc(&ne, (size_t)cap * 16);
  if (e == hipSuccess && h->staged_n > 0) {
    e = hipMemcpyAsync(nk, h->stage_keys.ptr, (size_t)h->staged_n * 4, hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ne, h->stage_events.ptr, (size_t)h->staged_n * 16, hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  }
  if (e != hipSuccess) {
    (void)hipFree(nk);
    if (ne) (void)hipFree(ne);
    return fail_hip(h, e, "growing the staging log");
  }
  h->stage_keys.release();
  h->stage_events.release();
  h->stage_keys.ptr = nk; h->stage_keys.cap = (size_t)cap * 4;
  h->stage_events.ptr = ne; h->stage_events.cap = (size_t)cap * 16;
  h->stage_cap = cap;
  return SURGE_OK;
}

int32_t surge_replay_stage_reserve(surge_replay_handle* h, int64_t n_events) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (n_events < 0) return fail(h, SURGE_E_INVALID, "negative size");
  DeviceGuard g(h->device);
  return stage_grow(h, n_events);
}

int32_t surge_replay_stage_events_device(surge_replay_handle* h, const int64_t* d_agg_idx, const void* d_events, int64_t n_events) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (h->v2) return fail(h, SURGE_E_UNSUPPORTED, "the packer serves v1 handles");
  if (n_events < 0) return fail(h, SURGE_E_INVALID, "negative size");
  if (n_events == 0) return SURGE_OK;
  if (!d_agg_idx || !d_events) return fail(h, SURGE_E_INVALID, "NULL batch buffer");
  DeviceGuard g(h->device);
  {
    const int32_t rc = stage_grow(h, h->staged_n + n_events);
    if (rc != SURGE_OK) return rc;
  }
  HIPCHK(h, launch_pack_stage(d_agg_idx, (uint32_t)n_events, (uint32_t*)h->stage_keys.ptr + h->staged_n, h->stream));
  HIPCHK(h, hipMemcpyAsync((char*)h->stage_events.ptr + (size_t)h->staged_n * 16, d_events, (size_t)n_events * 16, hipMemcpyDeviceToDevice, h->stream));
  h->staged_n += n_events;
  return SURGE_OK;
}

int32_t surge_replay_staged(surge_replay_handle* h, int64_t* n_events_out) {
  if (!h || !n_events_out) return fail(h, SURGE_E_INVALID, "NULL argument");
  *n_events_out = h->staged_n;
  return SURGE_OK;
}

int32_t surge_replay_pack_staged(surge_replay_handle* h, int64_t n_agg) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (h->v2) return fail(h, SURGE_E_UNSUPPORTED, "the packer serves v1 handles");
  if (n_agg < 0 || n_agg > 0xfffffffell) return fail(h, SURGE_E_INVALID, "n_agg out of range");
  DeviceGuard g(h->device);
  const uint32_t n = (uint32_t)h->staged_n;
  unsigned bits = 1;
  while (bits < 32 && ((uint64_t)(n_agg > 0 ? n_agg : 1) >> bits) != 0) ++bits;  // the key bits an index below n_agg needs
  size_t temp = 0;
  HIPCHK(h, pack_temp_bytes(n > 0 ? n : 1, bits, &temp));
  DevBuf scratch, seg, evs;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t rows = n > 0 ? n : 1;
  const size_t o_kb = up(temp), o_va = o_kb + up(rows * 4), o_vb = o_va + up(rows * 4), o_bad = o_vb + up(rows * 4), total = o_bad + 256;
  hipError_t e = scratch.reserve(total);
  if (e == hipSuccess) e = seg.reserve((size_t)(n_agg + 1) * 8);
  if (e == hipSuccess) e = evs.reserve(rows * 16);
  char* sb = (char*)scratch.ptr;
  if (e == hipSuccess)
    e = launch_pack((const uint32_t*)h->stage_keys.ptr, (const uint4*)h->stage_events.ptr, n, n_agg, bits, sb, temp, (uint32_t*)(sb + o_kb), (uint32_t*)(sb + o_va),
                    (uint32_t*)(sb + o_vb), (int64_t*)seg.ptr, (uint4*)evs.ptr, (uint32_t*)(sb + o_bad), h->stream);
  uint32_t bad = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&bad, sb + o_bad, 4, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  scratch.release();
  if (e != hipSuccess) {
    seg.release();
    evs.release();
    return fail_hip(h, e, "packing the staged events");
  }
  if (bad) {
    seg.release();
    evs.release();
    return fail(h, SURGE_E_RANGE, "a staged event names an aggregate index >= n_agg (nothing bound, the staging log kept)");
  }
  // the packed log becomes the handle's own bound log
  h->bound = false;
  h->own_seg_off.release();
  h->own_events.release();
  h->own_init.release();
  h->own_seg_off = seg;
  h->own_events = evs;
  seg.ptr = nullptr; seg.cap = 0; evs.ptr = nullptr; evs.cap = 0;
  h->stage_keys.release();
  h->stage_events.release();
  h->staged_n = h->stage_cap = 0;
  return surge_replay_bind_device_csr(h, (const int64_t*)h->own_seg_off.ptr, n_agg, h->own_events.ptr, (int64_t)n, nullptr, nullptr);
}

int32_t surge_replay_bound_log(surge_replay_handle* h, const int64_t** d_seg_off, const void** d_events, int64_t* n_agg, int64_t* n_events) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound) return fail(h, SURGE_E_STATE, "bound_log before load_csr/bind_device_csr/pack_staged");
  if (d_seg_off) *d_seg_off = h->d_seg_off;
  if (d_events) *d_events = h->d_events;
  if (n_agg) *n_agg = h->n_agg;
  if (n_events) *n_events = h->n_events;
  return SURGE_OK;
}

int32_t surge_replay_snapshot(surge_replay_handle* h, void* states_out, uint8_t* present_out) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound) return fail(h, SURGE_E_STATE, "snapshot before load_csr/bind_device_csr");
  if (h->st.n_folds == 0) return fail(h, SURGE_E_STATE, "snapshot before fold");
  DeviceGuard g(h->device);
  std::unique_lock<std::shared_mutex> lk(h->mu);
  const int64_t epoch = h->fold_epoch.load();
  const size_t bytes = (size_t)h->n_agg * 64;
  try {
    h->mirror.resize(bytes);
  } catch (const std::bad_alloc&) {
    return fail(h, SURGE_E_NOMEM, "out of host memory for the snapshot mirror");
  }
  if (bytes) {
    HIPCHK(h, hipMemcpyAsync(h->mirror.data(), h->d_state, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  h->mirror_epoch = epoch;
  int64_t poisoned = 0;
  for (int64_t a = 0; a < h->n_agg; ++a) {
    uint32_t fl;
    std::memcpy(&fl, h->mirror.data() + a * 64 + 36, 4);
    if (present_out) present_out[a] = (uint8_t)(fl & SURGE_STATE_PRESENT);
    poisoned += (fl & SURGE_STATE_POISONED) ? 1 : 0;
  }
  h->st.n_poisoned = poisoned;
  if (states_out && bytes) std::memcpy(states_out, h->mirror.data(), bytes);
  return SURGE_OK;
}

int32_t surge_replay_get(surge_replay_handle* h, int64_t agg_idx, void* state64_out, uint8_t* present_out) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!state64_out) return fail(h, SURGE_E_INVALID, "state64_out is NULL");
  if (!h->bound || h->st.n_folds == 0) return fail(h, SURGE_E_STATE, "get before fold");
  if (agg_idx < 0 || agg_idx >= h->n_agg) return fail(h, SURGE_E_RANGE, "aggregate index out of range");
  bool served = false;
  {
    std::shared_lock<std::shared_mutex> lk(h->mu);
    if (h->mirror_epoch == h->fold_epoch.load()) {
      std::memcpy(state64_out, h->mirror.data() + agg_idx * 64, 64);
      served = true;
    }
  }
  if (!served) {  // no mirror for this fold epoch: one device read at a time
    std::unique_lock<std::shared_mutex> lk(h->mu);
    DeviceGuard g(h->device);
    HIPCHK(h, hipMemcpyAsync(state64_out, h->d_state + agg_idx * 4, 64, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  if (present_out) {
    uint32_t fl;
    std::memcpy(&fl, (const uint8_t*)state64_out + 36, 4);
    *present_out = (uint8_t)(fl & SURGE_STATE_PRESENT);
  }
  return SURGE_OK;
}

int32_t surge_replay_gather(surge_replay_handle* h, const int64_t* agg_idx, int64_t n, void* states_out) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound || h->st.n_folds == 0) return fail(h, SURGE_E_STATE, "gather before fold");
  if (n < 0) return fail(h, SURGE_E_INVALID, "negative size");
  if (n == 0) return SURGE_OK;
  if (!agg_idx || !states_out) return fail(h, SURGE_E_INVALID, "NULL buffer");
  for (int64_t i = 0; i < n; ++i)
    if (agg_idx[i] < 0 || agg_idx[i] >= h->n_agg) return fail(h, SURGE_E_RANGE, "aggregate index out of range");
  DeviceGuard g(h->device);
  HIPCHK(h, h->gather_idx.reserve((size_t)n * 8));
  HIPCHK(h, h->gather_out.reserve((size_t)n * 64));
  HIPCHK(h, hipMemcpyAsync(h->gather_idx.ptr, agg_idx, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, launch_gather_states(h->d_state, (const int64_t*)h->gather_idx.ptr, n, (uint4*)h->gather_out.ptr, h->stream));
  HIPCHK(h, hipMemcpyAsync(states_out, h->gather_out.ptr, (size_t)n * 64, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return SURGE_OK;
}

static int32_t encode_states(surge_replay_handle* h, const surge_json_template* tmpl, const uint8_t* d_keys_utf8,
                             const int64_t* d_key_off, uint8_t* d_out, int64_t out_capacity, int64_t* d_out_off,
                             int64_t* total_bytes_out, uint32_t envelope) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound || h->st.n_folds == 0) return fail(h, SURGE_E_STATE, "encode_json before fold");
  if (!tmpl || !d_key_off || !d_out_off || !total_bytes_out) return fail(h, SURGE_E_INVALID, "NULL argument");
  if (tmpl->n_parts == 0 || tmpl->n_parts > SURGE_JSON_MAX_PARTS) return fail(h, SURGE_E_INVALID, "template.n_parts out of range");
  bool uses_f64 = false;
  for (uint32_t i = 0; i < tmpl->n_parts; ++i) {
    const auto& pt = tmpl->part[i];
    if (pt.kind > SURGE_JP_STR) return fail(h, SURGE_E_UNSUPPORTED, "unknown template part kind");
    if (pt.kind == SURGE_JP_LITERAL && (pt.lit_off > 256 || pt.lit_len > 256 - pt.lit_off)) return fail(h, SURGE_E_INVALID, "literal out of range");
    if (pt.kind == SURGE_JP_STR) {
      if (pt.field_offset >= SURGE_JSON_STRING_COLUMNS || !h->json_side.str_off[pt.field_offset])
        return fail(h, SURGE_E_INVALID, "SURGE_JP_STR names a string column that was not set (surge_replay_set_encode_strings)");
    } else if (pt.kind >= SURGE_JP_I32) {
      const uint32_t width = (pt.kind == SURGE_JP_I64 || pt.kind == SURGE_JP_F64) ? 8u : 4u;
      if (pt.field_offset + width > 64u || pt.field_offset % width) return fail(h, SURGE_E_INVALID, "field outside the 64-byte state or misaligned");
    }
    uses_f64 = uses_f64 || pt.kind == SURGE_JP_F64;
  }
  *total_bytes_out = 0;
  if (h->n_agg == 0) return SURGE_OK;
  DeviceGuard g(h->device);
  if (uses_f64 && !h->json_side.f64) {  // the power-of-5 tables of the Double text: one 10.7 KB copy per handle
    HIPCHK(h, h->f64_tables.reserve(sizeof(F64Tables)));
    HIPCHK(h, hipMemcpy(h->f64_tables.ptr, f64_tables_host(), sizeof(F64Tables), hipMemcpyHostToDevice));
    h->json_side.f64 = (const F64Tables*)h->f64_tables.ptr;
  }
  if (!h->json_side.not_a_number) {
    HIPCHK(h, h->nan_count.reserve(8));
    h->json_side.not_a_number = (unsigned long long*)h->nan_count.ptr;
  }
  HIPCHK(h, hipMemsetAsync(h->nan_count.ptr, 0, 8, h->stream));
  const int64_t nb = (h->n_agg + 1023) / 1024;
  HIPCHK(h, h->scan_totals.reserve((size_t)(nb + 1) * 8));
  HIPCHK(h, launch_json_encode(*tmpl, h->d_state, h->n_agg, d_keys_utf8, d_key_off, d_out_off, (int64_t*)h->scan_totals.ptr,
                               d_out, false, envelope, h->encode_filter, h->json_side, h->stream));
  int64_t total = 0;
  unsigned long long not_numbers = 0;
  HIPCHK(h, hipMemcpyAsync(&total, (int64_t*)h->scan_totals.ptr + nb, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(&not_numbers, h->nan_count.ptr, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpyAsync(d_out_off + h->n_agg, &total, 8, hipMemcpyHostToDevice, h->stream));
  *total_bytes_out = total;
  if (total > out_capacity) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return fail(h, SURGE_E_RANGE, "output buffer too small for the encoded snapshot");
  }
  if (total > 0 && !d_out) return fail(h, SURGE_E_INVALID, "d_out is NULL");
  HIPCHK(h, launch_json_encode(*tmpl, h->d_state, h->n_agg, d_keys_utf8, d_key_off, d_out_off, (int64_t*)h->scan_totals.ptr,
                               d_out, true, envelope, h->encode_filter, h->json_side, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (not_numbers)
    return fail(h, SURGE_E_UNSUPPORTED, std::to_string(not_numbers) + " aggregate(s) hold a NaN / infinite Double: no JSON number exists (the "
                                        "reference's writeState throws); they were encoded as zero bytes, everything else is valid");
  return SURGE_OK;
}

int32_t surge_replay_set_encode_strings(surge_replay_handle* h, int32_t column, const uint8_t* d_utf8, const int64_t* d_off) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (column < 0 || column >= SURGE_JSON_STRING_COLUMNS) return fail(h, SURGE_E_INVALID, "string column out of range");
  h->json_side.str[column] = d_utf8;
  h->json_side.str_off[column] = d_off;
  return SURGE_OK;
}

int32_t surge_replay_encode_json(surge_replay_handle* h, const surge_json_template* tmpl, const uint8_t* d_keys_utf8,
                                 const int64_t* d_key_off, uint8_t* d_out, int64_t out_capacity, int64_t* d_out_off,
                                 int64_t* total_bytes_out) {
  return encode_states(h, tmpl, d_keys_utf8, d_key_off, d_out, out_capacity, d_out_off, total_bytes_out, 0u);
}

int32_t surge_replay_encode_protobuf_state(surge_replay_handle* h, const surge_json_template* payload_tmpl,
                                           const uint8_t* d_keys_utf8, const int64_t* d_key_off, uint8_t* d_out,
                                           int64_t out_capacity, int64_t* d_out_off, int64_t* total_bytes_out) {
  return encode_states(h, payload_tmpl, d_keys_utf8, d_key_off, d_out, out_capacity, d_out_off, total_bytes_out, 1u);
}

int32_t surge_replay_snapshot_delta(surge_replay_handle* h, uint8_t* d_kind_out, int64_t* n_values_out, int64_t* n_tombstones_out,
                                    int32_t commit) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound || h->st.n_folds == 0) return fail(h, SURGE_E_STATE, "snapshot_delta before fold");
  if (!d_kind_out && h->n_agg > 0) return fail(h, SURGE_E_INVALID, "d_kind_out is NULL");
  DeviceGuard g(h->device);
  if (h->published_n < h->n_agg) {  // first use, or the resident state grew: new aggregates have never been published
    const size_t want = (size_t)h->n_agg * 64;
    if (want > h->published.cap) {
      void* fresh = nullptr;
      size_t cap = h->published.cap * 2 > want ? h->published.cap * 2 : want;
      HIPCHK(h, hipMalloc(&fresh, cap));
      if (h->published_n > 0)
        HIPCHK(h, hipMemcpyAsync(fresh, h->published.ptr, (size_t)h->published_n * 64, hipMemcpyDeviceToDevice, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      if (h->published.ptr) (void)hipFree(h->published.ptr);
      h->published.ptr = fresh;
      h->published.cap = cap;
    }
    HIPCHK(h, hipMemsetAsync((char*)h->published.ptr + (size_t)h->published_n * 64, 0, (size_t)(h->n_agg - h->published_n) * 64, h->stream));
    h->published_n = h->n_agg;
  }
  HIPCHK(h, h->poison_count.reserve(16));
  HIPCHK(h, launch_snapshot_delta(h->d_state, (uint4*)h->published.ptr, h->n_agg, d_kind_out, (unsigned long long*)h->poison_count.ptr,
                                  commit != 0, h->v2, h->stream));
  unsigned long long c[2] = {0, 0};
  HIPCHK(h, hipMemcpyAsync(c, h->poison_count.ptr, 16, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (n_values_out) *n_values_out = (int64_t)c[0];
  if (n_tombstones_out) *n_tombstones_out = (int64_t)c[1];
  h->delta_epoch = h->fold_epoch.load();
  h->delta_n = h->n_agg;
  return SURGE_OK;
}

int32_t surge_replay_snapshot_commit(surge_replay_handle* h, const uint8_t* d_kind) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound || h->published_n < h->n_agg) return fail(h, SURGE_E_STATE, "snapshot_commit without a preceding snapshot_delta");
  if (!d_kind && h->n_agg > 0) return fail(h, SURGE_E_INVALID, "d_kind is NULL");
  // the commit copies the CURRENT states of the reported aggregates into the baseline: after a fold / append / grow they
  // are no longer the states that were encoded, and a newer state would count as published without ever being emitted
  if (h->delta_epoch != h->fold_epoch.load() || h->delta_n != h->n_agg)
    return fail(h, SURGE_E_STATE, "snapshot_commit: the resident state changed since the snapshot_delta whose kinds these are "
                                  "(fold / append / grow in between); take a new delta");
  DeviceGuard g(h->device);
  HIPCHK(h, launch_snapshot_commit(h->d_state, (uint4*)h->published.ptr, h->n_agg, d_kind, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return SURGE_OK;
}

int32_t surge_replay_snapshot_invalidate(surge_replay_handle* h, const uint8_t* d_kind) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound || h->delta_n < 0 || h->published_n < h->delta_n) return fail(h, SURGE_E_STATE, "snapshot_invalidate without a preceding snapshot_delta");
  if (!d_kind && h->delta_n > 0) return fail(h, SURGE_E_INVALID, "d_kind is NULL");
  DeviceGuard g(h->device);
  // d_kind holds delta_n entries: the store may have grown (and folded) since; the aggregates added later have no
  // baseline to invalidate, and invalidating an older aggregate only makes the next delta report it again
  HIPCHK(h, launch_snapshot_invalidate((uint4*)h->published.ptr, h->delta_n, d_kind, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return SURGE_OK;
}

int32_t surge_replay_set_encode_filter(surge_replay_handle* h, const uint8_t* d_kind) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  h->encode_filter = d_kind;
  return SURGE_OK;
}

int32_t surge_replay_device_state(surge_replay_handle* h, void** d_states, int64_t* n_agg) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound) return fail(h, SURGE_E_STATE, "device_state before load_csr/bind_device_csr");
  if (d_states) *d_states = h->d_state;
  if (n_agg) *n_agg = h->n_agg;
  return SURGE_OK;
}

/* CPU variant of the shard map (R15).  This is product code (host-side routing needs it without a
 * GPU round trip), written independently of oracle/. */
static inline uint32_t rotl32_host(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

static int32_t partition_hash_host(const uint16_t* utf16, const int64_t* str_off, int64_t n, int32_t n_partitions,
                                   int32_t* part_out, bool up_to_colon) {
  if (n < 0 || n_partitions <= 0) return fail(nullptr, SURGE_E_INVALID, "bad n or n_partitions");
  if (n == 0) return SURGE_OK;
  if (!str_off || !part_out) return fail(nullptr, SURGE_E_INVALID, "NULL buffer");
  for (int64_t i = 0; i < n; ++i) {
    const int64_t b = str_off[i], e = str_off[i + 1];
    if (e < b) return fail(nullptr, SURGE_E_INVALID, "str_off is not monotone");
    int64_t len = e - b;
    if (up_to_colon) {  // PartitionStringUpToColon.partitionBy (KafkaPartitioner.scala:38-42)
      len = 0;
      while (b + len < e && utf16[b + len] != (uint16_t)':') ++len;
    }
    uint32_t hsh = 0xf7ca7fd2u;
    int64_t k = 0;
    for (; k + 1 < len; k += 2) {
      uint32_t d = ((uint32_t)utf16[b + k] << 16) + (uint32_t)utf16[b + k + 1];
      d *= 0xcc9e2d51u; d = rotl32_host(d, 15); d *= 0x1b873593u;
      hsh ^= d; hsh = rotl32_host(hsh, 13); hsh = hsh * 5u + 0xe6546b64u;
    }
    if (k < len) {
      uint32_t d = (uint32_t)utf16[b + k];
      d *= 0xcc9e2d51u; d = rotl32_host(d, 15); d *= 0x1b873593u;
      hsh ^= d;
    }
    hsh ^= (uint32_t)len;
    hsh ^= hsh >> 16; hsh *= 0x85ebca6bu; hsh ^= hsh >> 13; hsh *= 0xc2b2ae35u; hsh ^= hsh >> 16;
    const int32_t r = (int32_t)hsh % n_partitions;
    part_out[i] = r < 0 ? -r : r;
  }
  return SURGE_OK;
}

int32_t surge_replay_partition_hash(const uint16_t* utf16, const int64_t* str_off, int64_t n, int32_t n_partitions,
                                    int32_t* part_out) {
  return partition_hash_host(utf16, str_off, n, n_partitions, part_out, false);
}

int32_t surge_replay_partition_hash_up_to_colon(const uint16_t* utf16, const int64_t* str_off, int64_t n,
                                                int32_t n_partitions, int32_t* part_out) {
  return partition_hash_host(utf16, str_off, n, n_partitions, part_out, true);
}

static int32_t partition_hash_dev(surge_replay_handle* h, const uint16_t* d_utf16, const int64_t* d_str_off, int64_t n,
                                  int32_t n_partitions, int32_t* d_part_out, bool up_to_colon) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (n < 0 || n_partitions <= 0) return fail(h, SURGE_E_INVALID, "bad n or n_partitions");
  if (n == 0) return SURGE_OK;
  if (!d_str_off || !d_part_out) return fail(h, SURGE_E_INVALID, "NULL buffer");
  DeviceGuard g(h->device);
  HIPCHK(h, launch_partition_hash(d_utf16, d_str_off, n, n_partitions, d_part_out, up_to_colon, h->stream));
  return SURGE_OK;
}

int32_t surge_replay_partition_hash_device(surge_replay_handle* h, const uint16_t* d_utf16, const int64_t* d_str_off,
                                           int64_t n, int32_t n_partitions, int32_t* d_part_out) {
  return partition_hash_dev(h, d_utf16, d_str_off, n, n_partitions, d_part_out, false);
}

int32_t surge_replay_partition_hash_up_to_colon_device(surge_replay_handle* h, const uint16_t* d_utf16,
                                                       const int64_t* d_str_off, int64_t n, int32_t n_partitions,
                                                       int32_t* d_part_out) {
  return partition_hash_dev(h, d_utf16, d_str_off, n, n_partitions, d_part_out, true);
}

int32_t surge_replay_stats(surge_replay_handle* h, surge_replay_stats_t* out) {
  if (!h || !out) return fail(h, SURGE_E_INVALID, "NULL argument");
  DeviceGuard g(h->device);
  if (h->timing_valid) {
    HIPCHK(h, hipEventSynchronize(h->ev_total1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev_k0, h->ev_k1));
    h->st.last_fold_kernel_ms = ms;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev_total0, h->ev_total1));
    h->st.last_fold_total_ms = ms;
    const size_t n = h->folds_since_reset < kMaxTimedFolds ? h->folds_since_reset : kMaxTimedFolds;
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) {
      HIPCHK(h, hipEventElapsedTime(&ms, h->fold_events[i].first, h->fold_events[i].second));
      sum += ms;
    }
    h->st.sum_fold_kernel_ms = sum;
    h->st.timed_folds = (int64_t)n;
  }
  if (h->h2d_valid) {
    HIPCHK(h, hipEventSynchronize(h->ev_h1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev_h0, h->ev_h1));
    h->st.h2d_ms = ms;
  }
  if (h->bound && h->st.n_folds > 0 && h->st.n_poisoned < 0) {
    HIPCHK(h, h->poison_count.reserve(8));
    HIPCHK(h, launch_count_poisoned(h->d_state, h->n_agg, (unsigned long long*)h->poison_count.ptr, h->stream));
    unsigned long long c = 0;
    HIPCHK(h, hipMemcpyAsync(&c, h->poison_count.ptr, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->st.n_poisoned = (int64_t)c;
  }
  *out = h->st;
  return SURGE_OK;
}

int32_t surge_replay_fold_times(surge_replay_handle* h, double* ms_out, int64_t cap, int64_t* n_out) {
  if (!h || !n_out || cap < 0 || (!ms_out && cap > 0)) return fail(h, SURGE_E_INVALID, "bad argument");
  *n_out = 0;
  if (!h->timing_valid) return SURGE_OK;
  DeviceGuard g(h->device);
  HIPCHK(h, hipEventSynchronize(h->ev_total1));
  size_t n = h->folds_since_reset < kMaxTimedFolds ? h->folds_since_reset : kMaxTimedFolds;
  if ((int64_t)n > cap) n = (size_t)cap;
  for (size_t i = 0; i < n; ++i) {
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->fold_events[i].first, h->fold_events[i].second));
    ms_out[i] = ms;
  }
  *n_out = (int64_t)n;
  return SURGE_OK;
}

int32_t surge_replay_stats_reset(surge_replay_handle* h) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  DeviceGuard g(h->device);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->folds_since_reset = 0;
  h->timing_valid = false;
  h->st.sum_fold_kernel_ms = 0.0;
  h->st.timed_folds = 0;
  return SURGE_OK;
}

int32_t surge_replay_grow(surge_replay_handle* h, int64_t new_n_agg) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound) return fail(h, SURGE_E_STATE, "grow before load_csr/bind_device_csr");
  if (new_n_agg <= h->n_agg) return SURGE_OK;
  if (h->d_state != (uint4*)h->own_state.ptr)
    return fail(h, SURGE_E_UNSUPPORTED, "the state buffer belongs to the caller (bind_device_csr / set_state_out): grow it there");
  DeviceGuard g(h->device);
  std::unique_lock<std::shared_mutex> lk(h->mu);
  if ((size_t)new_n_agg * 64 > h->own_state.cap) {
    // amortised: at least double, so a stream of new aggregates reallocates O(log n) times
    size_t want = h->own_state.cap * 2;
    if (want < (size_t)new_n_agg * 64) want = (size_t)new_n_agg * 64;
    void* fresh = nullptr;
    HIPCHK(h, hipMalloc(&fresh, want));
    hipError_t e = hipMemcpyAsync(fresh, h->own_state.ptr, (size_t)h->n_agg * 64, hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
      (void)hipFree(fresh);
      return fail_hip(h, e, "copying the resident state");
    }
    (void)hipFree(h->own_state.ptr);
    h->own_state.ptr = fresh;
    h->own_state.cap = want;
    h->d_state = (uint4*)fresh;
  }
  // new aggregates are None (all-zero, the canonical encoding)
  HIPCHK(h, hipMemsetAsync((char*)h->d_state + (size_t)h->n_agg * 64, 0, (size_t)(new_n_agg - h->n_agg) * 64, h->stream));
  h->n_agg = new_n_agg;
  h->st.n_aggregates = new_n_agg;
  h->log_valid = false;
  h->fold_epoch.fetch_add(1);
  return SURGE_OK;
}

int32_t surge_replay_comm_unique_id(uint8_t id_out[SURGE_COMM_ID_BYTES]) {
  if (!id_out) return fail(nullptr, SURGE_E_INVALID, "id_out is NULL");
  std::string err;
  const int32_t rc = comm_unique_id(id_out, &err);
  return rc == SURGE_OK ? rc : fail(nullptr, rc, err);
}

int32_t surge_replay_comm_init(surge_replay_handle* h, int32_t rank, int32_t world, const uint8_t id[SURGE_COMM_ID_BYTES]) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!id) return fail(h, SURGE_E_INVALID, "id is NULL");
  if (h->comm) return fail(h, SURGE_E_STATE, "the handle already has a communicator (surge_replay_comm_destroy first)");
  DeviceGuard g(h->device);
  std::string err;
  const int32_t rc = comm_create(h->device, rank, world, id, &h->comm, &err);
  if (rc == SURGE_OK) h->comm_world = world;
  return rc == SURGE_OK ? rc : fail(h, rc, err);
}

int32_t surge_replay_comm_destroy(surge_replay_handle* h) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  DeviceGuard g(h->device);
  if (h->comm) comm_destroy(h->comm);
  h->comm = nullptr;
  return SURGE_OK;
}

int32_t surge_replay_comm_info(surge_replay_handle* h, int32_t* rank, int32_t* world, int32_t* rccl_version, const char** library) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->comm) return fail(h, SURGE_E_STATE, "no communicator: surge_replay_comm_init first");
  return comm_info(h->comm, rank, world, rccl_version, library);
}

int32_t surge_replay_comm_counts(surge_replay_handle* h, int64_t n_local, int64_t* counts_out, int64_t* max_count_out) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->comm) return fail(h, SURGE_E_STATE, "no communicator: surge_replay_comm_init first");
  if (n_local < 0) return fail(h, SURGE_E_INVALID, "negative size");
  DeviceGuard g(h->device);
  std::string err;
  const int32_t rc = comm_counts(h->comm, n_local, counts_out, max_count_out, true, &err);
  return rc == SURGE_OK ? rc : fail(h, rc, err);
}

int32_t surge_replay_allgather_snapshot(surge_replay_handle* h, const void* d_states, int64_t n_local, void* d_out,
                                        int64_t rows_per_rank, int32_t slot, int32_t mode) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->comm) return fail(h, SURGE_E_STATE, "no communicator: surge_replay_comm_init first");
  if (mode != SURGE_GATHER_P2P && mode != SURGE_GATHER_ALLGATHER && mode != SURGE_GATHER_P2P_RAW) return fail(h, SURGE_E_INVALID, "unknown gather mode");
  if (!d_states) {
    if (!h->bound) return fail(h, SURGE_E_STATE, "allgather_snapshot of the resident state before load_csr/bind_device_csr");
    if (n_local > h->n_agg) return fail(h, SURGE_E_RANGE, "n_local exceeds the resident state");
    d_states = h->d_state;
  }
  if (((uintptr_t)d_states & 7) || ((uintptr_t)d_out & 7)) return fail(h, SURGE_E_INVALID, "buffers must be 8-byte aligned");
  if (slot < 0 || slot > 1) return fail(h, SURGE_E_INVALID, "slot must be 0 or 1");
  DeviceGuard g(h->device);
  std::string err;
  if (!d_out) {  // the handle keeps the gathered snapshot (hosts without device pointers)
    int64_t mx = 0;
    const int32_t rc0 = comm_counts(h->comm, n_local, nullptr, &mx, false, &err);
    if (rc0 != SURGE_OK) return fail(h, rc0, err);
    rows_per_rank = mx;
    HIPCHK(h, hipStreamSynchronize(h->stream));  // a reallocation must not pull the buffer from under an earlier exchange
    {
      std::string e2;
      (void)comm_wait(h->comm, h->stream, slot, true, &e2);
    }
    HIPCHK(h, h->gathered[slot].reserve((size_t)h->comm_world * (size_t)(mx > 0 ? mx : 1) * 64));
    h->gathered_rows[slot] = mx;
    d_out = h->gathered[slot].ptr;
  }
  const int32_t rc = comm_allgather(h->comm, h->stream, d_states, n_local, d_out, rows_per_rank, slot, mode, !h->v2 && mode != SURGE_GATHER_P2P_RAW, &err);
  return rc == SURGE_OK ? rc : fail(h, rc, err);
}

int32_t surge_replay_allgather(surge_replay_handle* const* hs, int32_t n, const int64_t* n_local, void* const* d_out,
                               int64_t rows_per_rank, int32_t slot) {
  if (!hs || n < 1) return fail(nullptr, SURGE_E_INVALID, "no handles");
  for (int32_t r = 0; r < n; ++r)
    if (!hs[r]) return fail(nullptr, SURGE_E_INVALID, "a handle is NULL");
  surge_replay_handle* h0 = hs[0];
  if (slot < 0 || slot > 1) return fail(h0, SURGE_E_INVALID, "slot must be 0 or 1");
  std::vector<int64_t> counts((size_t)n);
  std::vector<CommState*> cs((size_t)n);
  std::vector<hipStream_t> streams((size_t)n);
  std::vector<const void*> src((size_t)n);
  std::vector<void*> dst((size_t)n);
  int64_t mx = 0;
  for (int32_t r = 0; r < n; ++r) {
    surge_replay_handle* h = hs[r];
    for (int32_t q = 0; q < r; ++q)
      if (hs[q] == h) return fail(h0, SURGE_E_INVALID, "a handle appears twice in the group");
    if (!h->bound) return fail(h0, SURGE_E_STATE, "allgather before load_csr/bind_device_csr on every handle");
    if (h->v2 != h0->v2) return fail(h0, SURGE_E_INVALID, "v1 and v2 handles cannot share a group");
    if (h->comm && !comm_is_local(h->comm)) return fail(h0, SURGE_E_STATE, "a handle holds an RCCL rank (surge_replay_comm_destroy first)");
    counts[(size_t)r] = n_local ? n_local[r] : h->n_agg;
    if (counts[(size_t)r] < 0 || counts[(size_t)r] > h->n_agg) return fail(h0, SURGE_E_RANGE, "n_local outside the resident state");
    mx = counts[(size_t)r] > mx ? counts[(size_t)r] : mx;
    if (d_out && (!d_out[r] || ((uintptr_t)d_out[r] & 7))) return fail(h0, SURGE_E_INVALID, "d_out entries must be non-NULL and 8-byte aligned");
  }
  if (!d_out) rows_per_rank = mx;
  std::string err;
  for (int32_t r = 0; r < n; ++r) {
    surge_replay_handle* h = hs[r];
    DeviceGuard g(h->device);
    int32_t cr = -1, cw = -1;
    if (h->comm) (void)comm_info(h->comm, &cr, &cw, nullptr, nullptr);
    if (h->comm && (cr != r || cw != n)) {  // the group changed shape
      comm_destroy(h->comm);
      h->comm = nullptr;
    }
    if (!h->comm) {
      const int32_t rc = comm_create_local(h->device, r, n, &h->comm, &err);
      if (rc != SURGE_OK) return fail(h0, rc, err);
      h->comm_world = n;
    }
    if (!d_out) {
      HIPCHK(h0, hipStreamSynchronize(h->stream));  // a reallocation must not pull the buffer from under an earlier exchange
      std::string e2;
      (void)comm_wait(h->comm, h->stream, slot, true, &e2);
      HIPCHK(h0, h->gathered[slot].reserve((size_t)n * (size_t)(mx > 0 ? mx : 1) * 64));
      h->gathered_rows[slot] = mx;
      dst[(size_t)r] = h->gathered[slot].ptr;
    } else {
      dst[(size_t)r] = d_out[r];
    }
    cs[(size_t)r] = h->comm;
    streams[(size_t)r] = h->stream;
    src[(size_t)r] = h->d_state;
  }
  const int32_t rc = comm_allgather_local(cs.data(), streams.data(), src.data(), counts.data(), dst.data(), rows_per_rank, n, slot, !h0->v2, &err);
  return rc == SURGE_OK ? rc : fail(h0, rc, err);
}

int32_t surge_replay_gathered(surge_replay_handle* h, int32_t slot, void** d_out, int64_t* rows_per_rank) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (slot < 0 || slot > 1) return fail(h, SURGE_E_INVALID, "slot must be 0 or 1");
  if (!h->gathered[slot].ptr) return fail(h, SURGE_E_STATE, "no handle-owned gathered snapshot in this slot (allgather_snapshot with d_out = NULL first)");
  if (d_out) *d_out = h->gathered[slot].ptr;
  if (rows_per_rank) *rows_per_rank = h->gathered_rows[slot];
  return SURGE_OK;
}

int32_t surge_replay_gathered_read(surge_replay_handle* h, int32_t slot, int32_t rank, int64_t first_row, int64_t n_rows,
                                   void* states_out) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->comm) return fail(h, SURGE_E_STATE, "no communicator: surge_replay_comm_init first");
  if (slot < 0 || slot > 1) return fail(h, SURGE_E_INVALID, "slot must be 0 or 1");
  if (!h->gathered[slot].ptr) return fail(h, SURGE_E_STATE, "no handle-owned gathered snapshot in this slot");
  if (rank < 0 || rank >= h->comm_world || first_row < 0 || n_rows < 0 || first_row + n_rows > h->gathered_rows[slot])
    return fail(h, SURGE_E_RANGE, "rank / rows outside the gathered snapshot");
  if (n_rows == 0) return SURGE_OK;
  if (!states_out) return fail(h, SURGE_E_INVALID, "states_out is NULL");
  DeviceGuard g(h->device);
  std::string err;
  const int32_t rc = comm_wait(h->comm, h->stream, slot, true, &err);
  if (rc != SURGE_OK) return fail(h, rc, err);
  const char* src = (const char*)h->gathered[slot].ptr + ((size_t)rank * (size_t)h->gathered_rows[slot] + (size_t)first_row) * 64;
  HIPCHK(h, hipMemcpy(states_out, src, (size_t)n_rows * 64, hipMemcpyDeviceToHost));
  return SURGE_OK;
}

int32_t surge_replay_comm_wait(surge_replay_handle* h, int32_t slot, int32_t host_sync) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->comm) return fail(h, SURGE_E_STATE, "no communicator: surge_replay_comm_init first");
  DeviceGuard g(h->device);
  std::string err;
  const int32_t rc = comm_wait(h->comm, h->stream, slot, host_sync != 0, &err);
  return rc == SURGE_OK ? rc : fail(h, rc, err);
}

int32_t surge_replay_set_state_out(surge_replay_handle* h, void* d_state_out) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (!h->bound) return fail(h, SURGE_E_STATE, "set_state_out before load_csr/bind_device_csr");
  if (!d_state_out || ((uintptr_t)d_state_out & 15)) return fail(h, SURGE_E_INVALID, "state buffer must be non-NULL and 16-byte aligned");
  h->d_state = (uint4*)d_state_out;
  h->fold_epoch.fetch_add(1);
  return SURGE_OK;
}

int32_t surge_replay_pack_states(surge_replay_handle* h, const void* d_states64, int64_t n, void* d_packed40, void* hip_stream) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (n < 0 || (n > 0 && (!d_states64 || !d_packed40))) return fail(h, SURGE_E_INVALID, "bad argument");
  if (((uintptr_t)d_states64 & 7) || ((uintptr_t)d_packed40 & 7)) return fail(h, SURGE_E_INVALID, "buffers must be 8-byte aligned");
  DeviceGuard g(h->device);
  HIPCHK(h, launch_pack_states(d_states64, n, d_packed40, false, hip_stream ? (hipStream_t)hip_stream : h->stream));
  return SURGE_OK;
}

int32_t surge_replay_unpack_states(surge_replay_handle* h, const void* d_packed40, int64_t n, void* d_states64, void* hip_stream) {
  if (!h) return fail(nullptr, SURGE_E_INVALID, "handle is NULL");
  if (n < 0 || (n > 0 && (!d_states64 || !d_packed40))) return fail(h, SURGE_E_INVALID, "bad argument");
  if (((uintptr_t)d_states64 & 7) || ((uintptr_t)d_packed40 & 7)) return fail(h, SURGE_E_INVALID, "buffers must be 8-byte aligned");
  DeviceGuard g(h->device);
  HIPCHK(h, launch_pack_states(d_packed40, n, d_states64, true, hip_stream ? (hipStream_t)hip_stream : h->stream));
  return SURGE_OK;
}

int32_t surge_replay_stream_probe(surge_replay_handle* h, const void* d_src, int64_t n_bytes, double* ms_out) {
  if (!h || !d_src || !ms_out) return fail(h, SURGE_E_INVALID, "NULL argument");
  if (n_bytes < 16 || (n_bytes & 15) || ((uintptr_t)d_src & 15)) return fail(h, SURGE_E_INVALID, "n_bytes/pointer must be 16-byte multiples");
  DeviceGuard g(h->device);
  HIPCHK(h, h->poison_count.reserve(8));
  if (std::getenv("SURGE_DBG_PROBE_TILES") && h->tiled_valid) {  // experiments: stream the handle's own tile-major copy
    d_src = h->t_tiles.ptr;
    n_bytes = h->t_n_sub * (int64_t)kTileSubBytes;
  }
  float best = 0.f;
  for (int variant = 0; variant < 6; ++variant) {  // plain / non-temporal loads, the LDS-DMA tile stream at 9 / 6 / 4 waves per CU, register tiles: report the fastest
    HIPCHK(h, hipEventRecord(h->ev_h0, h->stream));
    HIPCHK(h, launch_stream_probe((const uint4*)d_src, n_bytes / 16, (uint32_t*)h->poison_count.ptr, variant, h->stream));
    HIPCHK(h, hipEventRecord(h->ev_h1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev_h1));
    float t = 0.f;
    HIPCHK(h, hipEventElapsedTime(&t, h->ev_h0, h->ev_h1));
    if (variant == 0 || t < best) best = t;
  }
  const float ms = best;
  *ms_out = ms;
  h->h2d_valid = false;  // ev_h0/ev_h1 were reused
  return SURGE_OK;
}

}  // extern "C"
