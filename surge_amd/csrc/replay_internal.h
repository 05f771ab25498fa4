// replay_internal.h — shared between the HIP kernels and the host engine.
// Not part of the C ABI (that is include/surge_replay.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/surge_replay.h"

namespace surge {

// Tile geometry: one wave = 64 lanes x LE consecutive events per tile (LE = 8 or 16), staged through
// LDS with direct global->LDS loads; see Geo<LE> in fold_kernels.hip.
constexpr int kWave = 64;
constexpr int kTaskBytes = 256 * 1024;                    // a wave task streams about this many bytes of events
constexpr int kTableEntries = 18;                         // 16 event types + [16] unknown type (poison) + [17] null (padding) event
constexpr int kTableWords = 16;                           // 64 B of pre-expanded masks per event type
constexpr int kTableStride = 20;                          // dwords between entries in LDS (80 B: conflict-free b128 reads)
// The 16 type entries at a stride of 20 dwords tile the 64 LDS banks exactly (20 e mod 64 hits every multiple of 4 once),
// so a 17th / 18th entry must share banks with one of them.  [16] (unknown type) sits at 320 = bank 0 with type 0; the
// null (padding) event, which every partial tile is full of, is moved off bank 20 (type 1, the commonest event of the
// Counter model) onto bank 44, shared with type 15.
constexpr int kNullEntryOff = 17 * kTableStride + 24;     // dword offset of the null entry [17] in LDS
constexpr int kTableLdsDwords = kNullEntryOff + 16;

// Per-type op table, pre-expanded on the host from the ABI descriptor so the kernel applies an event
// with VALU mask arithmetic only (no per-event decode, no compares, no branches).  Every word is an
// all-ones / all-zero mask except TW_EVC (0 or 1).
enum {
  TW_CNT_NZ = 0,   // count += / -= arg
  TW_CNT_NEG = 1,  // ... negated (SUB)
  TW_CNT_SET = 2,  // count := arg
  TW_VER_SET = 3,  // version := seq
  TW_SUM_NZ = 4,   // sum64 += / -= (long) arg
  TW_SUM_NEG = 5,
  TW_BAL_SET = 6,  // balance := value
  TW_EVC = 7,      // event_count += this (0 / 1)
  TW_POISON = 8,   // handleEvent throws
  TW_DELETE = 9,   // result is None
  TW_MATERIALIZES = 10,  // class MATERIALIZE or CREATE: result is always Some
  TW_NOT_REQUIRE = 11,   // applies to None as well
  TW_CREATE = 12,  // resets to defaults even when Some
  TW_MIN = 13,
  TW_MAX = 14,
  TW_FLAGS = 15,   // presence pre-pass: bit0 poison, bit16 delete, bit1 materializes (OR-ed in at << j)
};
constexpr int kTargetTasks = 16384;                       // enough tasks to fill the chip several times over

struct FoldParams {
  const uint4* events;      // 16 B records
  int64_t n_events;         // length of the events buffer (loads are clamped to it)
  const int64_t* seg_off;   // kernel-facing CSR offsets, strictly increasing (FLAT); unused for FIXED
  const int64_t* plan;      // FLAT: n_tasks+1 segment indices; task k owns segments [plan[k], plan[k+1])
                            // SORTED: perm[n_seg], kernel-facing segment ids by descending length
  unsigned long long* counter;  // SORTED: group dispenser, zeroed before every launch
  const int64_t* out_map;   // nullable: segment rank -> aggregate index (compacted CSR / micro-batch groups)
  const uint4* init;        // nullable: prior snapshot, 64 B per aggregate
  uint4* out;               // 64 B per aggregate
  int64_t n_seg;            // kernel-facing segment count
  int64_t fixed_len;        // FIXED: events per segment (multiple of 16)
  int64_t segs_per_task;    // FIXED: segments per wave task
  uint32_t table[kTableEntries][kTableWords];  // see TF_* above; unused slots and [16] = poison
  int32_t d_count, d_version;
  int64_t d_sum;
  uint64_t d_balance;
  int32_t d_min, d_max;
  uint32_t d_evcount;
};

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
hipError_t launch_fold_flat(const FoldParams& p, int64_t n_tasks, int lane_events, hipStream_t stream);
hipError_t launch_fold_rows(const FoldParams& p, int64_t n_tasks, int lane_events, hipStream_t stream);
hipError_t launch_fold_sorted(const FoldParams& p, int64_t n_waves, int lane_events, hipStream_t stream);
constexpr int kSortBucketsHost = 65536;
hipError_t launch_sort_by_length(const int64_t* off, int64_t n_seg, unsigned long long* d_hist, int64_t* perm,
                                 hipStream_t stream);
// CHUNKED (fold_chunked.hip): chunk table of the kernel-facing CSR, then the fold over it + the stitch kernel
constexpr int kChunkBucketsHost = 65536;
hipError_t launch_chunk_count(const int64_t* off, int64_t n_seg, uint32_t T, unsigned long long* d_hist,
                              unsigned long long* d_total, unsigned long long* d_ctr, hipStream_t stream);
hipError_t launch_chunk_scatter(const int64_t* off, int64_t n_seg, const int64_t* out_map, uint32_t T,
                                unsigned long long* d_cursor, unsigned long long* d_ctr, int64_t* v_start, uint32_t* v_len,
                                uint32_t* v_info, int64_t* v_dest, int64_t* r_slot0, uint32_t* r_c, int64_t* r_out,
                                hipStream_t stream);
hipError_t launch_fold_chunked(const FoldParams& p, const int64_t* v_start, const uint32_t* v_len, const uint32_t* v_info,
                               const int64_t* v_dest, int64_t n_vrows, uint32_t* side, const int64_t* r_slot0, const uint32_t* r_c,
                               const int64_t* r_out, int64_t n_cut, int64_t n_waves, int lane_events, hipStream_t stream);
hipError_t launch_plan(const int64_t* off, int64_t n_seg, int64_t task_events, int64_t n_tasks,
                       int64_t* plan, hipStream_t stream);
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
hipError_t launch_json_encode(const surge_json_template& tmpl, const uint4* states, int64_t n, const uint8_t* keys,
                              const int64_t* key_off, int64_t* d_len_off, int64_t* d_totals, uint8_t* out, bool write_pass,
                              uint32_t envelope, const uint8_t* filter, hipStream_t stream);
// kind[a] in SURGE_SNAP_*; d_counts: two u64 {values, tombstones}; commit: published := states where kind != SKIP
hipError_t launch_snapshot_delta(const uint4* states, uint4* published, int64_t n, uint8_t* kind, unsigned long long* d_counts,
                                 bool commit, bool full64, hipStream_t stream);

// ---- fold_slots.hip: the fold of ABI v2 slot schemas ------------------------------------------------------------
struct SlotParams;
constexpr size_t kSlotParamsBytes = 512;  // >= sizeof(SlotParams): the engine keeps it as opaque storage
void slot_params_from_schema(const surge_replay_schema_v2& sc, SlotParams* out);
hipError_t launch_fold_slots(const FoldParams& p, const SlotParams& sp, int64_t n_waves, int lane_events, hipStream_t stream);
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

// ---- comm.hip: the snapshot exchange over RCCL (dlopen'ed) ----------------------------------------------------
struct CommState;
int32_t comm_unique_id(uint8_t* id_out, std::string* err);
int32_t comm_create(int device, int rank, int world, const uint8_t* id, CommState** out, std::string* err);
void comm_destroy(CommState* c);
int32_t comm_info(const CommState* c, int32_t* rank, int32_t* world, int32_t* version, const char** library);
int32_t comm_counts(CommState* c, int64_t n_local, int64_t* counts_out, int64_t* max_count_out, std::string* err);
int32_t comm_allgather(CommState* c, hipStream_t compute, const void* d_states, int64_t n_local, void* d_out,
                       int64_t out_rows_per_rank, int slot, int mode, bool packed, std::string* err);
int32_t comm_wait(CommState* c, hipStream_t compute, int slot, bool host, std::string* err);

}  // namespace surge
