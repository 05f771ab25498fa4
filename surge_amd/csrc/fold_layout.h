// fold_layout.h — the contract between the fold kernels and the host engine that feeds them: tile geometry, the LDS op
// table and the kernel parameter block.  Anything here changes what the kernels do (bench.py hashes this file with the
// kernel sources to decide whether a committed rocprof traffic figure still describes the code).
#pragma once

// Also compiled at run time (hiprtc, fold_slots.hip's schema-specialised kernels): hiprtc has no system headers and
// brings the HIP device runtime with it, so the includes are skipped there and the fixed-width types come from here.
#ifdef __HIPCC_RTC__
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long int64_t;
typedef unsigned long uint64_t;
#else
#include <stdint.h>

#include <hip/hip_runtime.h>
#endif

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
  TW_NOT_REQUIRE = 10,  // applies to None as well
  TW_CREATE = 11,  // resets to defaults even when Some
  TW_MIN = 12,
  TW_MAX = 13,
  // Words [0, 14) are what the per-event walk reads (three ds_read_b128 + one ds_read_b64): every loaded register is
  // used.  With a dead word inside a b128 the register allocator reuses its destination at once and hipcc has to put an
  // s_waitcnt lgkmcnt(0) right behind the prefetch of the NEXT event's entry (write-after-write on an outstanding LDS
  // load) — one exposed LDS round trip per event (round 3: visible in the ISA of every walk).
  TW_MATERIALIZES = 14,  // flat kernel's presence pre-pass only: class MATERIALIZE or CREATE, the result is always Some
  TW_FLAGS = 15,   // flat kernel's presence pre-pass only: bit0 poison, bit16 delete (OR-ed in at << j)
};
constexpr int kTableWalkWords = 14;
constexpr int kTargetTasks = 8192;                        // enough tasks to fill the chip (2048 resident waves) four times over; 16384
                                                          // cost 3-13 % on logs below 1 GB: per-task start-up (plan, offsets, op table, first tile)

struct FoldParams {
  const uint4* events;      // 16 B records
  int64_t n_events;         // length of the events buffer (loads are clamped to it)
  const int64_t* seg_off;   // kernel-facing CSR offsets, strictly increasing (FLAT); unused for FIXED
  const int64_t* plan;      // FLAT: n_tasks+1 segment indices; task k owns segments [plan[k], plan[k+1])
                            // SORTED: perm[n_seg], kernel-facing segment ids by descending length
  unsigned long long* counter;  // SORTED / CHUNKED / SLOTS: {group tickets, waves done}: zero before the first launch; the
                                // last wave to leave a launch zeroes both again (no memset node in front of every fold)
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

// The tile-major copy of a bound log (fold_tiled.hip builds it; fold_tiled.hip and the slot kernels fold it).
struct TileTable {
  const uint4* tiles;        // the tile-major log
  const int64_t* g_sub0;     // n_groups + 1: first subtile of every group
  const uint32_t* v_len;     // per virtual row: events
  const uint32_t* v_info;    // VI_*
  const int64_t* v_dest;     // aggregate index (state array) or side-buffer slot
  int64_t n_vrows;
  uint32_t* side;
};
constexpr int kSubEvents = 8;                       // events per lane in one subtile
constexpr int kSubBytes = kWave * kSubEvents * 16;  // 8 KiB

}  // namespace surge
