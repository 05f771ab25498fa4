// replay_internal.h — shared between the HIP kernels and the host engine.
// Not part of the C ABI (that is include/surge_replay.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/surge_replay.h"
#include "fold_layout.h"

namespace surge {

struct CsrAnalysis {
  int32_t bad;              // a negative segment length was seen
  int32_t nonuniform;       // lengths differ
  int64_t n_empty;
  int64_t len0;             // length of segment 0
  int64_t max_len;
  int64_t first, last;      // seg_off[0], seg_off[n]
};

// Launch wrappers (fold_kernels.hip).  All asynchronous on `stream`.
hipError_t launch_fold_fixed(const FoldParams& p, int64_t n_tasks, int lane_events, hipStream_t stream);
// The flat kernel compiled for one v1 op table (hiprtc; fold_kernels.hip).  A V1Kernels object is shared by every handle
// of the process with the same table on the same device; nullptr = the ahead-of-time kernel that reads the table from LDS.
enum { V1_FLAT = 0, V1_LANES = 1 };  // the two run-time compiled programs of a v1 op table (fold_kernels.hip)
struct V1Kernels {
  hipModule_t module = nullptr;
  hipFunction_t flat8 = nullptr, flat16 = nullptr;                                  // V1_FLAT
  hipFunction_t sorted8 = nullptr, sorted16 = nullptr, sorted32 = nullptr, chunked8 = nullptr, chunked16 = nullptr, rows8 = nullptr, rows16 = nullptr;  // V1_LANES
  int device = 0;
  double compile_ms = 0.0;
};
std::string v1_spec_source(const uint32_t (*table)[kTableWords], int kind);  // the program handed to hiprtc ("" = not expressible)
void v1_kernels_acquire(const uint32_t (*table)[kTableWords], int device, int kind, V1Kernels** out, double* compile_ms, std::string* why);
hipError_t launch_v1_lane(hipFunction_t fn, const void* args, size_t args_bytes, int64_t grid, unsigned lds, hipStream_t stream);
hipError_t launch_fold_flat(const FoldParams& p, const V1Kernels* spec, int64_t n_tasks, int lane_events, hipStream_t stream);
// lanes: the V1_LANES kernels compiled for the handle's op table, or nullptr = the ahead-of-time kernels (op table in LDS)
hipError_t launch_fold_rows(const FoldParams& p, const V1Kernels* lanes, int64_t n_tasks, int lane_events, hipStream_t stream);
hipError_t launch_fold_sorted(const FoldParams& p, int64_t n_waves, int lane_events, hipStream_t stream);
hipError_t launch_fold_short(const FoldParams& p, hipStream_t stream);  // K1s: one lane per aggregate of a log of many short rows (p.seg_off over ALL aggregates, empty ones included)
// the same fold, pipelined across groups (fold_chunked.hip: the chunked kernel's walk over whole aggregates); 8 or 16 events per lane
hipError_t launch_fold_sorted_pf(const FoldParams& p, const V1Kernels* lanes, int64_t n_waves, int lane_events, hipStream_t stream);
// ---- index_kernels.hip: the per-log indexes (length order, chunk table), built with rocPRIM sorts / scans -----------
struct IndexScratch {  // engine-owned device scratch, sized for the rows being ordered
  void* temp;          // the counting sort's histograms (counting) or rocPRIM temporary storage
  size_t temp_bytes;
  uint32_t *keys_a, *keys_b;  // n x u32 each (keys_b: rocPRIM path only)
  int64_t *vals_a, *vals_b;   // n x i64 each (vals_a: rocPRIM path only; launch_sort_by_length writes its result to `perm` instead of vals_b)
  bool counting;       // keys <= max_key < kCountSortMaxBins: the hand-written counting sort (length_sort.hip), else rocPRIM's radix sort
  uint32_t max_key;
  int n_cus;
};
hipError_t index_temp_bytes(int64_t n, size_t* bytes);  // index_radix.hip: rocPRIM's radix sort (rows of 8192 events and more)
hipError_t launch_radix_sort_pairs_desc(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const int64_t* vals_in,
                                        int64_t* vals_out, int64_t n, hipStream_t stream);
// ---- length_sort.hip: the stable descending counting sort of the indexes, and the exclusive scans of the chunk counts ----------
constexpr int kCountSortMaxBins = 8192;
size_t count_sort_scratch_bytes(int64_t n, uint32_t max_key, int n_cus);
hipError_t launch_count_sort_desc(const uint32_t* keys, const int64_t* off, int64_t n, uint32_t max_key, int n_cus, void* scratch, int64_t* perm,
                                  hipStream_t stream);
size_t scan_i64_scratch_bytes(int64_t n, int k_arrays);
hipError_t launch_exclusive_scans_i64(int64_t* v, int64_t n, int64_t stride, int k_arrays, void* scratch, hipStream_t stream);
hipError_t launch_sort_by_length(const int64_t* off, int64_t n_seg, const IndexScratch& sc, int64_t* perm, hipStream_t stream);
// CHUNKED / TILED: the chunk table of the kernel-facing CSR.  cnt: 3 x (n_seg + 1) int64.
hipError_t launch_chunk_count(const int64_t* off, int64_t n_seg, uint32_t T, bool align, int64_t* cnt, void* scan_scratch, hipStream_t stream);
hipError_t launch_chunk_table(const int64_t* off, int64_t n_seg, const int64_t* out_map, uint32_t T, bool align, const int64_t* cnt,
                              int64_t n_vrows, const IndexScratch& sc, int64_t* u_start, uint32_t* u_len, uint32_t* u_info, int64_t* u_dest,
                              int64_t* v_start, uint32_t* v_len, uint32_t* v_info, int64_t* v_dest, int64_t* r_slot0, uint32_t* r_c,
                              int64_t* r_out, hipStream_t stream);
hipError_t launch_fold_chunked(const FoldParams& p, const int64_t* v_start, const uint32_t* v_len, const uint32_t* v_info,
                               const int64_t* v_dest, int64_t n_vrows, uint32_t* side, const int64_t* r_slot0, const uint32_t* r_c,
                               const int64_t* r_out, int64_t n_cut, const V1Kernels* lanes, int64_t n_waves, int lane_events, hipStream_t stream);
hipError_t launch_chunk_stitch(const FoldParams& p, const uint32_t* side, const int64_t* r_slot0, const uint32_t* r_c, const int64_t* r_out,
                               int64_t n_cut, hipStream_t stream);
// TILED (fold_tiled.hip): the chunk table's virtual rows copied once into group-major / tile-major order, then the fold
// over that copy.  g_sub: n_groups + 1 int64 (subtiles per group, scanned in place into offsets by the caller).
constexpr int kTileSubBytes = 8192;
hipError_t launch_tile_index(const uint32_t* v_len, int64_t n_vrows, int64_t* g_sub, hipStream_t stream);
hipError_t launch_relayout(const uint4* events, const int64_t* v_start, const uint32_t* v_len, int64_t n_vrows, const int64_t* g_sub0,
                           int64_t n_sub_total, uint4* tiles, hipStream_t stream);
hipError_t launch_fold_tiled(const FoldParams& p, const uint4* tiles, const int64_t* g_sub0, const uint32_t* v_len,
                             const uint32_t* v_info, const int64_t* v_dest, int64_t n_vrows, uint32_t* side, int64_t n_waves, int subs,
                             hipStream_t stream);
hipError_t launch_exclusive_scan_i64(int64_t* v, int64_t n, hipStream_t stream);
hipError_t launch_plan(const int64_t* off, int64_t n_seg, int64_t task_events, int64_t n_tasks,
                       int64_t* plan, hipStream_t stream);
hipError_t launch_plan_dev(const int64_t* off, const uint32_t* d_n_seg, int64_t task_events, int64_t n_tasks, int64_t* plan,
                           hipStream_t stream);
hipError_t launch_analyze_csr(const int64_t* off, int64_t n_seg, CsrAnalysis* d_result, hipStream_t stream);
hipError_t launch_fill_empty(const int64_t* off, int64_t n_seg, const uint4* init, uint4* out,
                             hipStream_t stream);
// Stable compaction of non-empty segments: nz_off[n_nz+1], nz_map[n_nz].  d_block_counts is scratch of
// ceil(n_seg/1024)+1 int64.
hipError_t launch_compact_nonempty(const int64_t* off, int64_t n_seg, int64_t* d_block_counts,
                                   int64_t* nz_off, int64_t* nz_map, hipStream_t stream);
hipError_t launch_partition_hash(const uint16_t* utf16, const int64_t* str_off, int64_t n,
                                 int32_t n_partitions, int32_t* part_out, bool up_to_colon, hipStream_t stream);
hipError_t launch_stream_probe(const uint4* src, int64_t n_vec, uint32_t* sink, int variant, hipStream_t stream);
struct F64Tables;
struct JsonSide {
  const F64Tables* f64;                              // device copy of the Double-text tables (f64_text.h); needed by SURGE_JP_F64
  const uint8_t* str[SURGE_JSON_STRING_COLUMNS];     // side string columns (SURGE_JP_STR), UTF-8
  const int64_t* str_off[SURGE_JSON_STRING_COLUMNS];
  unsigned long long* not_a_number;                  // += aggregates skipped because a Double of theirs is NaN / infinite
};
hipError_t launch_json_encode(const surge_json_template& tmpl, const uint4* states, int64_t n, const uint8_t* keys,
                              const int64_t* key_off, int64_t* d_len_off, int64_t* d_totals, uint8_t* out, bool write_pass,
                              uint32_t envelope, const uint8_t* filter, const JsonSide& side, hipStream_t stream);
// kind[a] in SURGE_SNAP_*; d_counts: two u64 {values, tombstones}; commit: published := states where kind != SKIP
hipError_t launch_snapshot_invalidate(uint4* published, int64_t n, const uint8_t* kind, hipStream_t stream);
hipError_t launch_snapshot_commit(const uint4* states, uint4* published, int64_t n, const uint8_t* kind, hipStream_t stream);
hipError_t launch_snapshot_delta(const uint4* states, uint4* published, int64_t n, uint8_t* kind, unsigned long long* d_counts,
                                 bool commit, bool full64, hipStream_t stream);

// ---- fold_slots.hip: the fold of ABI v2 slot schemas ------------------------------------------------------------
struct SlotParams;
struct SlotKernels;
constexpr size_t kSlotParamsBytes = 512;  // >= sizeof(SlotParams): the engine keeps it as opaque storage
void slot_params_from_schema(const surge_replay_schema_v2& sc, SlotParams* out);
hipError_t launch_fold_slots(const FoldParams& p, const SlotParams& sp, const SlotKernels* spec, int64_t n_waves, int lane_events,
                             hipStream_t stream);
// Schema-specialised build of the slot kernels (hiprtc; rtc.cpp + fold_slots.hip).  A SlotKernels object is shared by
// every handle of the process with the same schema on the same device; nullptr = the interpreter.
bool rtc_compile(const std::string& source, const char* arch, std::vector<char>* code, std::string* log, double* ms);
const char* rtc_library_path();
std::string slots_spec_source(const SlotParams& sp);  // the program handed to hiprtc for this schema
// never fails the caller: on any problem *out stays nullptr and *why says what happened
void slot_kernels_acquire(const surge_replay_schema_v2& sc, const SlotParams& sp, int device, SlotKernels** out, double* compile_ms,
                          std::string* why);
hipError_t launch_fold_slots_tiled(const FoldParams& p, const SlotParams& sp, const SlotKernels* spec, const TileTable& t, int64_t n_waves,
                                   int subs, hipStream_t stream);
// unpack == false: in = n x 64 B, out = n x 40 B; unpack == true: in = n x 40 B, out = n x 64 B
hipError_t launch_pack_states(const void* in, int64_t n, void* out, bool unpack, hipStream_t stream);
hipError_t launch_gather_states(const uint4* states, const int64_t* idx, int64_t n, uint4* out, hipStream_t stream);
hipError_t launch_count_poisoned(const uint4* states, int64_t n, unsigned long long* d_count,
                                 hipStream_t stream);


// ---- stream_kernels.hip: K3 micro-batch group-by on the device ------------------------------------------------
hipError_t groupby_temp_bytes(uint32_t n, unsigned key_bits, size_t* bytes);
hipError_t launch_groupby(const int64_t* d_agg_idx, const uint4* d_events, uint32_t n, int64_t n_agg, unsigned key_bits, void* d_temp,
                          size_t temp_bytes, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, uint32_t* head,
                          uint32_t* gid, uint4* d_sorted_events, int64_t* d_group_agg, int64_t* d_group_off, uint32_t* d_flags,
                          hipStream_t stream);

// the packer (surge_replay_pack_staged): staged (aggregate, event) pairs in topic order -> CSR
hipError_t launch_pack_stage(const int64_t* d_agg_idx, uint32_t n, uint32_t* keys, hipStream_t stream);
hipError_t pack_temp_bytes(uint32_t n, unsigned key_bits, size_t* bytes);
hipError_t launch_pack(const uint32_t* keys_a, const uint4* staged_events, uint32_t n, int64_t n_agg, unsigned key_bits, void* d_temp, size_t temp_bytes,
                       uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int64_t* seg_off, uint4* out, uint32_t* d_bad, hipStream_t stream);

// ---- comm.hip: the snapshot exchange over RCCL (dlopen'ed) ----------------------------------------------------
struct CommState;
int32_t comm_unique_id(uint8_t* id_out, std::string* err);
int32_t comm_create(int device, int rank, int world, const uint8_t* id, CommState** out, std::string* err);
void comm_destroy(CommState* c);
int32_t comm_info(const CommState* c, int32_t* rank, int32_t* world, int32_t* version, const char** library);
// force: run the (collective) exchange even when counts for this n_local are cached — the public, documented-collective call
int32_t comm_counts(CommState* c, int64_t n_local, int64_t* counts_out, int64_t* max_count_out, bool force, std::string* err);
int32_t comm_allgather(CommState* c, hipStream_t compute, const void* d_states, int64_t n_local, void* d_out,
                       int64_t out_rows_per_rank, int slot, int mode, bool packed, std::string* err);
int32_t comm_wait(CommState* c, hipStream_t compute, int slot, bool host, std::string* err);
// in-process groups (one host process drives every rank; peer copies instead of RCCL)
int32_t comm_create_local(int device, int rank, int world, CommState** out, std::string* err);
bool comm_is_local(const CommState* c);
int32_t comm_allgather_local(CommState* const* cs, const hipStream_t* compute, const void* const* d_states, const int64_t* n_local,
                             void* const* d_out, int64_t out_rows_per_rank, int world, int slot, bool packed, std::string* err);

}  // namespace surge
