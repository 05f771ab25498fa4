// comm.hip — the path's one exchange step behind the C ABI: all-gather(v) of the final snapshot over RCCL / xGMI
// (SURVEY §8b `surge_replay_allgather`, §8e).  One process per GPU, one communicator rank per handle.
//
// Why grouped send/recv and not ncclAllGather by default: xGMI on an MI355X node is a point-to-point full mesh
// (7 links per GPU), so one ncclSend/ncclRecv pair per peer inside ONE group puts exactly one peer's shard on each
// link, all links busy at once (40 B x 1.25 M aggregates = 50 MB per link ~ 0.35 ms at ~150 GB/s); a ring all-gather
// pushes all N-1 shards through one link pair.  Shards differ in size (aggregates shard by partition key), hence
// "all-gather-v": counts are exchanged once per shard size, buffers are max-padded.  States travel in the 40-byte
// wire form (the 24-byte reserved tail of a state is always zero) and are expanded on arrival.
//
// librccl is dlopen'ed at the first comm call, not linked: the fold, the point reads and every single-GPU host work
// without it, and inside a process that already carries an RCCL (PyTorch bundles one) that same copy is used
// (RTLD_NOLOAD first) instead of a second runtime.  SURGE_RCCL_LIBRARY names the library to use instead (it wins over a
// copy already in the process; tests/rccl_stub uses it to put two ranks on one GPU).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "replay_internal.h"

namespace surge {

namespace {

struct RcclApi {
  void* lib = nullptr;
  std::string path;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
};

std::mutex g_api_mu;
RcclApi g_api;
bool g_api_tried = false;
std::string g_api_err;

template <class F>
bool sym(void* lib, const char* name, F* out, std::string* err) {
  *out = (F)dlsym(lib, name);
  if (!*out) {
    *err = std::string("librccl lacks ") + name;
    return false;
  }
  return true;
}

const RcclApi* rccl(std::string* err) {
  std::lock_guard<std::mutex> lk(g_api_mu);
  if (g_api.lib) return &g_api;
  if (g_api_tried) {
    *err = g_api_err;
    return nullptr;
  }
  g_api_tried = true;
  void* lib = nullptr;
  std::string used;
  if (const char* v = std::getenv("SURGE_RCCL_LIBRARY")) {
    // an explicit choice wins over whatever the process already carries (RTLD_LOCAL: its symbols stay out of the global scope)
    lib = dlopen(v, RTLD_NOW | RTLD_LOCAL);
    used = v;
  } else {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !lib; ++pass)  // pass 0: a copy this process already loaded (PyTorch bundles one)
      for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (lib) { used = n; break; }
      }
  }
  if (!lib) {
    const char* why = dlerror();
    g_api_err = std::string("cannot load librccl (set SURGE_RCCL_LIBRARY): ") + (why ? why : "not found");
    *err = g_api_err;
    return nullptr;
  }
  RcclApi a;
  a.lib = lib;
  a.path = used;
  std::string e;
  const bool ok = sym(lib, "ncclGetUniqueId", &a.GetUniqueId, &e) && sym(lib, "ncclCommInitRank", &a.CommInitRank, &e) &&
                  sym(lib, "ncclCommDestroy", &a.CommDestroy, &e) && sym(lib, "ncclGroupStart", &a.GroupStart, &e) &&
                  sym(lib, "ncclGroupEnd", &a.GroupEnd, &e) && sym(lib, "ncclSend", &a.Send, &e) &&
                  sym(lib, "ncclRecv", &a.Recv, &e) && sym(lib, "ncclAllGather", &a.AllGather, &e) &&
                  sym(lib, "ncclGetErrorString", &a.GetErrorString, &e) && sym(lib, "ncclGetVersion", &a.GetVersion, &e);
  if (!ok) {
    g_api_err = e;
    *err = e;
    return nullptr;
  }
  g_api = a;
  return &g_api;
}

}  // namespace

struct CommState {
  const RcclApi* api = nullptr;
  ncclComm_t comm = nullptr;
  int device = 0, rank = 0, world = 1;
  hipStream_t side = nullptr;          // the exchange runs here, beside the fold's stream
  hipEvent_t ready = nullptr;          // "the states to publish are final" (recorded on the fold's stream)
  hipEvent_t done[2] = {nullptr, nullptr};  // exchange of slot s finished (recorded on the side stream)
  hipEvent_t staged = nullptr;         // in-process groups: this rank's wire form is complete (peers copy after it)
  bool local = false;                  // rank of an in-process group (no RCCL communicator; peer copies)
  bool peers_enabled = false;          // in-process groups: peer access to the other ranks' devices has been requested
  bool launched[2] = {false, false};
  std::vector<int64_t> counts;         // states per rank of the current shard sizes
  int64_t counts_for = -1;             // n_local the counts were exchanged for
  int64_t max_count = 0;
  bool wire_dirty = false;             // shard sizes changed: rows beyond the new counts may hold an older exchange
  void* d_wire_local = nullptr;        // n_local x 40 B
  void* d_wire_all = nullptr;          // world x max_count x 40 B
  void* d_counts = nullptr;            // world x int64
  size_t wire_local_cap = 0, wire_all_cap = 0;
};

namespace {

int32_t comm_fail(std::string* err, int32_t code, const std::string& m) {
  if (err) *err = m;
  return code;
}

#define COMM_HIP(call)                                                                              \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) return comm_fail(err, SURGE_E_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define COMM_NCCL(c, call)                                                                          \
  do {                                                                                              \
    ncclResult_t r_ = (call);                                                                       \
    if (r_ != ncclSuccess)                                                                          \
      return comm_fail(err, SURGE_E_COMM, std::string(#call) + ": " + (c)->api->GetErrorString(r_)); \
  } while (0)

int32_t reserve_dev(void** p, size_t* cap, size_t bytes, std::string* err) {
  if (bytes <= *cap) return SURGE_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  COMM_HIP(hipMalloc(p, bytes ? bytes : 16));
  COMM_HIP(hipMemset(*p, 0, bytes ? bytes : 16));  // padding rows of the wire buffer stay zero = None
  *cap = bytes ? bytes : 16;
  return SURGE_OK;
}

}  // namespace

int32_t comm_unique_id(uint8_t* id_out, std::string* err) {
  const RcclApi* a = rccl(err);
  if (!a) return SURGE_E_COMM;
  ncclUniqueId id;
  const ncclResult_t r = a->GetUniqueId(&id);
  if (r != ncclSuccess) return comm_fail(err, SURGE_E_COMM, std::string("ncclGetUniqueId: ") + a->GetErrorString(r));
  static_assert(sizeof(id) == SURGE_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  std::memcpy(id_out, &id, sizeof(id));
  return SURGE_OK;
}

void comm_destroy(CommState* c) {
  if (!c) return;
  if (c->side) (void)hipStreamSynchronize(c->side);
  if (c->comm) (void)c->api->CommDestroy(c->comm);
  for (hipEvent_t e : {c->ready, c->done[0], c->done[1], c->staged})
    if (e) (void)hipEventDestroy(e);
  if (c->side) (void)hipStreamDestroy(c->side);
  for (void* p : {c->d_wire_local, c->d_wire_all, c->d_counts})
    if (p) (void)hipFree(p);
  delete c;
}

int32_t comm_create(int device, int rank, int world, const uint8_t* id, CommState** out, std::string* err) {
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return comm_fail(err, SURGE_E_INVALID, "rank / world out of range");
  const RcclApi* a = rccl(err);
  if (!a) return SURGE_E_COMM;
  CommState* c = new (std::nothrow) CommState();
  if (!c) return comm_fail(err, SURGE_E_NOMEM, "out of host memory");
  c->api = a;
  c->device = device;
  c->rank = rank;
  c->world = world;
  auto bail = [&](int32_t rc) {
    comm_destroy(c);
    return rc;
  };
  hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[0], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[1], hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc(&c->d_counts, (size_t)world * 8);
  if (e != hipSuccess) return bail(comm_fail(err, SURGE_E_DEVICE, std::string("comm resources: ") + hipGetErrorString(e)));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  const ncclResult_t r = a->CommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    return bail(comm_fail(err, SURGE_E_COMM, std::string("ncclCommInitRank: ") + a->GetErrorString(r)));
  }
  c->counts.assign((size_t)world, 0);
  *out = c;
  return SURGE_OK;
}

int32_t comm_info(const CommState* c, int32_t* rank, int32_t* world, int32_t* version, const char** library) {
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (version) {
    int v = 0;
    if (c->api) (void)c->api->GetVersion(&v);
    *version = v;
  }
  if (library) *library = c->api ? c->api->path.c_str() : "in-process peer copies";
  return SURGE_OK;
}

// counts[r] = states rank r contributes.  The exchange is a COLLECTIVE (all-gather + host sync), so whether it runs must
// never depend on one rank's local state alone: it runs when the host asks for it on every rank (surge_replay_comm_counts:
// force) and, implicitly, in a rank's very first exchange (no counts yet — every rank of a fresh communicator is in that
// state).  A rank whose shard size differs from the one the cached counts were exchanged for gets SURGE_E_STATE instead of
// starting a collective its peers are not in (shards can change unevenly: surge_replay_grow, append_* after a grow) —
// round-2 review: a skipped / unskipped all-gather on one side only is an operation mismatch, i.e. a hang.
static int32_t exchange_counts(CommState* c, int64_t n_local, bool force, std::string* err) {
  if (!force) {
    if (c->counts_for == n_local) return SURGE_OK;
    if (c->counts_for >= 0)
      return comm_fail(err, SURGE_E_STATE,
                       "this rank's shard size changed since the shard sizes were exchanged (" + std::to_string(c->counts_for) + " -> " +
                           std::to_string(n_local) + "): call surge_replay_comm_counts on EVERY rank before the next exchange");
  }
  int64_t mine = n_local;
  COMM_HIP(hipMemcpyAsync((char*)c->d_counts + (size_t)c->rank * 8, &mine, 8, hipMemcpyHostToDevice, c->side));
  if (c->world > 1)
    COMM_NCCL(c, c->api->AllGather((char*)c->d_counts + (size_t)c->rank * 8, c->d_counts, 8, ncclUint8, c->comm, c->side));
  COMM_HIP(hipMemcpyAsync(c->counts.data(), c->d_counts, (size_t)c->world * 8, hipMemcpyDeviceToHost, c->side));
  COMM_HIP(hipStreamSynchronize(c->side));
  c->max_count = 0;
  for (int64_t n : c->counts) {
    if (n < 0) return comm_fail(err, SURGE_E_COMM, "a rank reported a negative shard size");
    c->max_count = n > c->max_count ? n : c->max_count;
  }
  c->counts_for = n_local;
  c->wire_dirty = true;
  return SURGE_OK;
}

int32_t comm_counts(CommState* c, int64_t n_local, int64_t* counts_out, int64_t* max_count_out, bool force, std::string* err) {
  if (c->local) {  // an in-process group learns its shard sizes in surge_replay_allgather
    if (c->counts_for < 0) return comm_fail(err, SURGE_E_STATE, "in-process group: shard sizes are set by surge_replay_allgather");
  } else {
    const int32_t rc = exchange_counts(c, n_local, force, err);
    if (rc != SURGE_OK) return rc;
  }
  if (counts_out) std::memcpy(counts_out, c->counts.data(), (size_t)c->world * 8);
  if (max_count_out) *max_count_out = c->max_count;
  return SURGE_OK;
}

// d_out[r * max_count + i] := rank r's state i (64 B); rows i >= counts[r] are None (zero).  Asynchronous: the side
// stream first waits for everything enqueued so far on `compute`, and records done[slot] at the end.
int32_t comm_allgather(CommState* c, hipStream_t compute, const void* d_states, int64_t n_local, void* d_out,
                       int64_t out_rows_per_rank, int slot, int mode, bool packed, std::string* err) {
  if (slot < 0 || slot > 1) return comm_fail(err, SURGE_E_INVALID, "slot must be 0 or 1");
  if (n_local < 0 || (n_local > 0 && !d_states) || !d_out) return comm_fail(err, SURGE_E_INVALID, "bad argument");
  if (c->local) return comm_fail(err, SURGE_E_STATE, "rank of an in-process group: exchange with surge_replay_allgather");
  int32_t rc = exchange_counts(c, n_local, false, err);
  if (rc != SURGE_OK) return rc;
  if (out_rows_per_rank < c->max_count)
    return comm_fail(err, SURGE_E_RANGE, "d_out holds fewer rows per rank than the largest shard (see surge_replay_comm_counts)");
  const int64_t M = c->max_count;
  if (!packed) {
    // v2 slot schemas use all 64 bytes of a state, and SURGE_GATHER_P2P_RAW asks for it: shards travel as they are,
    // straight between the state arrays (grouped per-peer send / recv; rows between a rank's count and the largest
    // shard are zeroed so they read as None; rows beyond the largest shard of a wider d_out stay untouched)
    COMM_HIP(hipEventRecord(c->ready, compute));
    COMM_HIP(hipStreamWaitEvent(c->side, c->ready, 0));
    char* out = (char*)d_out;
    const size_t pitch = (size_t)out_rows_per_rank * 64;
    for (int r = 0; r < c->world; ++r)  // rows between a rank's count and the largest shard read as None
      if (c->counts[(size_t)r] < M)
        COMM_HIP(hipMemsetAsync(out + (size_t)r * pitch + (size_t)c->counts[(size_t)r] * 64, 0, (size_t)(M - c->counts[(size_t)r]) * 64, c->side));
    if (n_local > 0) COMM_HIP(hipMemcpyAsync(out + (size_t)c->rank * pitch, d_states, (size_t)n_local * 64, hipMemcpyDeviceToDevice, c->side));
    if (c->world > 1) {
      COMM_NCCL(c, c->api->GroupStart());
      ncclResult_t r = ncclSuccess;
      for (int d = 1; d < c->world && r == ncclSuccess; ++d) {
        const int to = (c->rank + d) % c->world, frm = (c->rank - d + c->world) % c->world;
        const size_t sb = (size_t)n_local * 64, rb = (size_t)c->counts[(size_t)frm] * 64;
        if (sb) r = c->api->Send(d_states, sb, ncclUint8, to, c->comm, c->side);
        if (r == ncclSuccess && rb) r = c->api->Recv(out + (size_t)frm * pitch, rb, ncclUint8, frm, c->comm, c->side);
      }
      const ncclResult_t g = c->api->GroupEnd();
      if (r != ncclSuccess) return comm_fail(err, SURGE_E_COMM, std::string("ncclSend/ncclRecv: ") + c->api->GetErrorString(r));
      if (g != ncclSuccess) return comm_fail(err, SURGE_E_COMM, std::string("ncclGroupEnd: ") + c->api->GetErrorString(g));
    }
    COMM_HIP(hipEventRecord(c->done[slot], c->side));
    c->launched[slot] = true;
    return SURGE_OK;
  }
  rc = reserve_dev(&c->d_wire_local, &c->wire_local_cap, (size_t)(M > 0 ? M : 1) * SURGE_PACKED_STATE_SIZE, err);
  if (rc != SURGE_OK) return rc;
  rc = reserve_dev(&c->d_wire_all, &c->wire_all_cap, (size_t)c->world * (size_t)(M > 0 ? M : 1) * SURGE_PACKED_STATE_SIZE, err);
  if (rc != SURGE_OK) return rc;

  COMM_HIP(hipEventRecord(c->ready, compute));
  COMM_HIP(hipStreamWaitEvent(c->side, c->ready, 0));
  if (c->wire_dirty) {  // padding rows must read as None
    COMM_HIP(hipMemsetAsync(c->d_wire_local, 0, c->wire_local_cap, c->side));
    COMM_HIP(hipMemsetAsync(c->d_wire_all, 0, c->wire_all_cap, c->side));
    c->wire_dirty = false;
  }
  COMM_HIP(launch_pack_states(d_states, n_local, c->d_wire_local, false, c->side));
  char* all = (char*)c->d_wire_all;
  const size_t stride = (size_t)M * SURGE_PACKED_STATE_SIZE;
  if (mode == SURGE_GATHER_ALLGATHER && c->world > 1) {
    // library collective over max-padded shards (ring / tree chosen by RCCL)
    COMM_NCCL(c, c->api->AllGather(c->d_wire_local, all, stride, ncclUint8, c->comm, c->side));
  } else {
    if (n_local > 0)
      COMM_HIP(hipMemcpyAsync(all + (size_t)c->rank * stride, c->d_wire_local, (size_t)n_local * SURGE_PACKED_STATE_SIZE,
                              hipMemcpyDeviceToDevice, c->side));
    if (c->world > 1) {
      COMM_NCCL(c, c->api->GroupStart());
      ncclResult_t r = ncclSuccess;
      for (int d = 1; d < c->world && r == ncclSuccess; ++d) {  // skewed peer order: in step d rank r talks to r+d / r-d
        const int to = (c->rank + d) % c->world, frm = (c->rank - d + c->world) % c->world;
        const size_t sb = (size_t)n_local * SURGE_PACKED_STATE_SIZE, rb = (size_t)c->counts[(size_t)frm] * SURGE_PACKED_STATE_SIZE;
        if (sb) r = c->api->Send(c->d_wire_local, sb, ncclUint8, to, c->comm, c->side);
        if (r == ncclSuccess && rb) r = c->api->Recv(all + (size_t)frm * stride, rb, ncclUint8, frm, c->comm, c->side);
      }
      const ncclResult_t g = c->api->GroupEnd();
      if (r != ncclSuccess) return comm_fail(err, SURGE_E_COMM, std::string("ncclSend/ncclRecv: ") + c->api->GetErrorString(r));
      if (g != ncclSuccess) return comm_fail(err, SURGE_E_COMM, std::string("ncclGroupEnd: ") + c->api->GetErrorString(g));
    }
  }
  // expand world x M packed rows to 64 bytes (padding rows are zero = None); rows beyond M of a wider d_out stay untouched
  if (out_rows_per_rank == M) {
    COMM_HIP(launch_pack_states(all, (int64_t)c->world * M, d_out, true, c->side));
  } else {
    for (int r = 0; r < c->world; ++r)
      COMM_HIP(launch_pack_states(all + (size_t)r * stride, M, (char*)d_out + (size_t)r * (size_t)out_rows_per_rank * 64, true, c->side));
  }
  COMM_HIP(hipEventRecord(c->done[slot], c->side));
  c->launched[slot] = true;
  return SURGE_OK;
}

// ---- in-process groups: one host process drives every rank, shards move as peer copies (xGMI between GPUs) --------
int32_t comm_create_local(int device, int rank, int world, CommState** out, std::string* err) {
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return comm_fail(err, SURGE_E_INVALID, "rank / world out of range");
  CommState* c = new (std::nothrow) CommState();
  if (!c) return comm_fail(err, SURGE_E_NOMEM, "out of host memory");
  c->local = true;
  c->device = device;
  c->rank = rank;
  c->world = world;
  hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  for (hipEvent_t* ev : {&c->ready, &c->done[0], &c->done[1], &c->staged})
    if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    comm_destroy(c);
    return comm_fail(err, SURGE_E_DEVICE, std::string("comm resources: ") + hipGetErrorString(e));
  }
  c->counts.assign((size_t)world, 0);
  *out = c;
  return SURGE_OK;
}

bool comm_is_local(const CommState* c) { return c->local; }

// Every rank at once.  Rank r stages its shard in wire form on its own side stream (40-byte packed rows, or the raw 64
// bytes of a v2 state); every destination then pulls the n staged shards with (peer) copies on ITS side stream — with
// the ranks on different GPUs each copy crosses exactly one xGMI link and all links of a destination run at once — and
// expands them into d_out[d].  Nothing here blocks the host.
int32_t comm_allgather_local(CommState* const* cs, const hipStream_t* compute, const void* const* d_states, const int64_t* n_local,
                             void* const* d_out, int64_t out_rows_per_rank, int world, int slot, bool packed, std::string* err) {
  if (slot < 0 || slot > 1) return comm_fail(err, SURGE_E_INVALID, "slot must be 0 or 1");
  int64_t M = 0;
  for (int r = 0; r < world; ++r) {
    if (!cs[r] || !cs[r]->local || cs[r]->world != world || cs[r]->rank != r) return comm_fail(err, SURGE_E_STATE, "not rank r of this in-process group");
    if (n_local[r] < 0 || (n_local[r] > 0 && !d_states[r]) || !d_out[r]) return comm_fail(err, SURGE_E_INVALID, "bad argument");
    M = n_local[r] > M ? n_local[r] : M;
  }
  if (out_rows_per_rank < M) return comm_fail(err, SURGE_E_RANGE, "d_out holds fewer rows per rank than the largest shard");
  const size_t W = packed ? (size_t)SURGE_PACKED_STATE_SIZE : 64;
  const size_t stride = (size_t)M * W;
  int prev_dev = 0;
  COMM_HIP(hipGetDevice(&prev_dev));
  struct Restore {
    int d;
    ~Restore() { (void)hipSetDevice(d); }
  } restore{prev_dev};

  for (int r = 0; r < world; ++r) {  // stage
    CommState* c = cs[r];
    COMM_HIP(hipSetDevice(c->device));
    bool same = c->counts_for == n_local[r];
    for (int q = 0; q < world && same; ++q) same = c->counts[(size_t)q] == n_local[q];
    if (!same) {
      for (int q = 0; q < world; ++q) c->counts[(size_t)q] = n_local[q];
      c->counts_for = n_local[r];
      c->max_count = M;
      c->wire_dirty = true;
    }
    int32_t rc = reserve_dev(&c->d_wire_local, &c->wire_local_cap, (size_t)(M > 0 ? M : 1) * 64, err);
    if (rc == SURGE_OK && packed) rc = reserve_dev(&c->d_wire_all, &c->wire_all_cap, (size_t)world * (size_t)(M > 0 ? M : 1) * W, err);
    if (rc != SURGE_OK) return rc;
    COMM_HIP(hipEventRecord(c->ready, compute[r]));
    COMM_HIP(hipStreamWaitEvent(c->side, c->ready, 0));
    // the peers may still be pulling the previous exchange out of this rank's staging buffer
    for (int q = 0; q < world; ++q)
      for (int s = 0; s < 2; ++s)
        if (q != r && cs[q]->launched[s]) COMM_HIP(hipStreamWaitEvent(c->side, cs[q]->done[s], 0));
    if (c->wire_dirty && packed) COMM_HIP(hipMemsetAsync(c->d_wire_all, 0, c->wire_all_cap, c->side));
    c->wire_dirty = false;
    if (n_local[r] > 0) {
      if (packed)
        COMM_HIP(launch_pack_states(d_states[r], n_local[r], c->d_wire_local, false, c->side));
      else
        COMM_HIP(hipMemcpyAsync(c->d_wire_local, d_states[r], (size_t)n_local[r] * 64, hipMemcpyDeviceToDevice, c->side));
    }
    COMM_HIP(hipEventRecord(c->staged, c->side));
  }
  for (int d = 0; d < world; ++d) {  // pull + expand
    CommState* c = cs[d];
    COMM_HIP(hipSetDevice(c->device));
    if (!c->peers_enabled) {  // direct xGMI reads of the peers' staging buffers (hipMemcpyPeerAsync stages through the host otherwise)
      for (int r = 0; r < world; ++r) {
        int can = 0;
        if (cs[r]->device != c->device && hipDeviceCanAccessPeer(&can, c->device, cs[r]->device) == hipSuccess && can) {
          const hipError_t pe = hipDeviceEnablePeerAccess(cs[r]->device, 0);
          if (pe != hipSuccess) (void)hipGetLastError();  // already enabled (by the host, by another handle): fine
        }
      }
      c->peers_enabled = true;
    }
    const size_t pitch = (size_t)out_rows_per_rank * 64;
    if (!packed)  // rows between a rank's count and the largest shard read as None
      for (int r = 0; r < world; ++r)
        if (n_local[r] < M)
          COMM_HIP(hipMemsetAsync((char*)d_out[d] + (size_t)r * pitch + (size_t)n_local[r] * 64, 0, (size_t)(M - n_local[r]) * 64, c->side));
    for (int k = 0; k < world; ++k) {
      const int r = (d + k) % world;  // skewed: at step k every destination reads a different source
      const size_t bytes = (size_t)n_local[r] * W;
      if (r != d) COMM_HIP(hipStreamWaitEvent(c->side, cs[r]->staged, 0));
      if (!bytes) continue;
      char* dst = packed ? (char*)c->d_wire_all + (size_t)r * stride : (char*)d_out[d] + (size_t)r * pitch;
      if (cs[r]->device == c->device)
        COMM_HIP(hipMemcpyAsync(dst, cs[r]->d_wire_local, bytes, hipMemcpyDeviceToDevice, c->side));
      else
        COMM_HIP(hipMemcpyPeerAsync(dst, c->device, cs[r]->d_wire_local, cs[r]->device, bytes, c->side));
    }
    if (packed) {
      if (out_rows_per_rank == M) {
        COMM_HIP(launch_pack_states(c->d_wire_all, (int64_t)world * M, d_out[d], true, c->side));
      } else {
        for (int r = 0; r < world; ++r)
          COMM_HIP(launch_pack_states((char*)c->d_wire_all + (size_t)r * stride, M, (char*)d_out[d] + (size_t)r * pitch, true, c->side));
      }
    }
    COMM_HIP(hipEventRecord(c->done[slot], c->side));
    c->launched[slot] = true;
  }
  return SURGE_OK;
}

int32_t comm_wait(CommState* c, hipStream_t compute, int slot, bool host, std::string* err) {
  if (slot < 0 || slot > 1) return comm_fail(err, SURGE_E_INVALID, "slot must be 0 or 1");
  if (!c->launched[slot]) return SURGE_OK;
  if (host)
    COMM_HIP(hipEventSynchronize(c->done[slot]));
  else
    COMM_HIP(hipStreamWaitEvent(compute, c->done[slot], 0));
  return SURGE_OK;
}

}  // namespace surge
