// libcerbos_hip.so - MI355X (gfx950 / CDNA4) batched decision engine for the Cerbos
// CheckResources hot path.  Hand-written HIP; no MFMA (branchy integer / string-id work,
// HBM- and VALU-issue bound).
//
// Kernels
//   cbh_resolve_globs_kernel : one lane per batch-local string; bit-parallel glob NFA
//                              (LDS-staged transition tables) -> match bits per dimension.
//                              Replaces gobwas glob.Match per query
//                              (internal/ruletable/index/glob_dimension.go:62-95).
//   cbh_check_kernel*        : one lane per CheckInput (its actions as a bit mask), wave-uniform
//                              table walk (cbh_check_wave.h); restates
//                              ruletable.(*RuleTable).check (internal/ruletable/check.go:97-460),
//                              Index.Query + appendRolePolicyDenies
//                              (internal/ruletable/index/index.go:214-530), GetAllScopes
//                              (internal/ruletable/ruletable.go:848-882) over the flat table image.
//
// Host side: the C ABI of include/cerbos_hip.h.
//   * a table = one replica of the image per device of the engine (broadcast once: RCCL, or peer copies);
//   * cbh_check_batch = the reference's fan-out (engine.go:309-338) as contiguous request ranges over the
//     devices, each range pipelined in chunks over three streams (upload / kernels / download overlap);
//   * tables are reference counted, so a released table drains its in-flight batches (manager.go:86-124).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <map>
#include <tuple>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cbh_kernels.h"
#include "cbh_image.h"
#include "cbh_wire_host.h"

// ======================================================================== host code
// (skipped in the device compilation pass, where the device structs carry address-space qualifiers)
#if !defined(__HIP_DEVICE_COMPILE__)

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return -1; }
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(_e)); } while (0)

// ---- engine-wide state -------------------------------------------------------------------------------------
struct Rccl {   // the few RCCL entry points the image broadcast needs, bound at first use (no link-time dependency:
                // a single-GPU deployment never loads the collective library)
  void* lib = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  std::vector<void*> comms;
  bool tried = false, ok = false;
};
static struct Engine {
  std::mutex mu;
  bool inited = false;
  std::vector<int> devices;
  Rccl rccl;
} g_eng;

// CBH_TRACE=1: one line per one-shot call on stderr (path taken, phase times) - measurement aid
static bool trace_on() { static const bool on = getenv("CBH_TRACE") != nullptr; return on; }
// ... and, for the sliced road (cbh_wire_check_pb), the phases of every slice's thread: marks of (label, microseconds since the call began)
struct WireMarks { std::chrono::steady_clock::time_point t0; std::vector<std::pair<const char*, double>> v; };
static thread_local WireMarks* tl_marks = nullptr;
static inline void wmark(const char* label) {
  if (tl_marks) tl_marks->v.push_back({label, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tl_marks->t0).count()});
}
static double now_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
// CBH_SPIN=1: wait for a stream by polling hipStreamQuery instead of hipStreamSynchronize
static hipError_t stream_wait(hipStream_t s) {
  static const bool spin = [] { const char* e = getenv("CBH_SPIN"); return e && *e == '1'; }();
  if (!spin) return hipStreamSynchronize(s);
  for (;;) { const hipError_t e = hipStreamQuery(s); if (e != hipErrorNotReady) return e; }
}
static const size_t SMALL_BATCH_BYTES = (size_t)1 << 20;   // inputs under this: one staged copy each way
static const u32 SHARD_MIN_REQUESTS = 8192;                 // a device is given at least this many requests
static const int N_STREAMS = 3;

extern "C" const char* cbh_last_error(void) { return g_err.c_str(); }
extern "C" uint32_t cbh_abi_version(void) { return CBH_ABI_VERSION; }

extern "C" int cbh_init(const cbh_config* cfg) {
  if (cfg && cfg->abi_version != CBH_ABI_VERSION) return fail("ABI version mismatch");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return fail("no HIP device available: the decision engine requires an MI355X (no CPU fallback)");
  std::vector<int> devs;
  if (cfg && cfg->n_devices) {
    if (cfg->n_devices > CBH_MAX_DEVICES) return fail("too many devices");
    for (u32 i = 0; i < cfg->n_devices; ++i) {
      if (cfg->devices[i] < 0 || cfg->devices[i] >= n) return fail("invalid device ordinal");
      devs.push_back(cfg->devices[i]);
    }
  } else {
    for (int i = 0; i < n && i < (int)CBH_MAX_DEVICES; ++i) devs.push_back(i);
    if (!cfg) devs.resize(1);   // no configuration: the first device only
  }
  for (int d : devs) { HIPCHK(hipSetDevice(d)); HIPCHK(hipFree(nullptr)); }   // create the contexts now, not under a request
  // peer access between the engine's devices (image broadcast by peer copy, should RCCL be unavailable)
  for (int a : devs) for (int b : devs) if (a != b) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) { (void)hipSetDevice(a); (void)hipDeviceEnablePeerAccess(b, 0); (void)hipGetLastError(); }
  }
  HIPCHK(hipSetDevice(devs[0]));
  std::lock_guard<std::mutex> lk(g_eng.mu);
  g_eng.devices = devs;
  g_eng.inited = true;
  return 0;
}

extern "C" void cbh_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_eng.mu);
  Rccl& r = g_eng.rccl;
  if (r.ok) for (void* c : r.comms) if (c) (void)r.CommDestroy(c);
  r.comms.clear(); r.ok = false; r.tried = false;
  if (r.lib) { dlclose(r.lib); r.lib = nullptr; }
  g_eng.inited = false;
  g_eng.devices.clear();
}
extern "C" uint32_t cbh_num_devices(void) { std::lock_guard<std::mutex> lk(g_eng.mu); return (uint32_t)g_eng.devices.size(); }
extern "C" int32_t cbh_device_ordinal(uint32_t i) { std::lock_guard<std::mutex> lk(g_eng.mu); return i < g_eng.devices.size() ? g_eng.devices[i] : -1; }

extern "C" void* cbh_alloc_pinned(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); g_err = "hipHostMalloc failed"; return nullptr; }
  return p;
}
extern "C" void cbh_free_pinned(void* p) { if (p) (void)hipHostFree(p); }

// ---- tables -----------------------------------------------------------------------------------------------
struct OneShot {   // what one cbh_check_batch call owns on one device while it runs
  hipStream_t s[N_STREAMS] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_setup = nullptr;
  hipEvent_t ev_piece[N_STREAMS] = {nullptr, nullptr, nullptr};   // the pieces of a slab upload (run_range)
  uint8_t* h = nullptr; size_t h_cap = 0;   // pinned staging block (small batches)
  uint8_t* d = nullptr; size_t d_cap = 0;   // device block
};

struct Replica {   // the table on one device
  int device = 0;
  void* image = nullptr; bool owns_image = true;
  TableDev dev{};
  hipStream_t stream = nullptr;   // resident path (and the image broadcast)
  // Resident batches are dealt round-robin to a few streams - a batch keeps the one it was uploaded on, so everything that
  // touches it stays ordered - and launches of consecutive batches overlap: the dispatch ramp of one fills the CUs the
  // drain of the one before leaves idle (~40 % of a 15 us launch is ramp + drain, profiles/r02_cycles_flat_C2.txt).
  static constexpr int MAX_RESIDENT_STREAMS = 8;
  hipStream_t rstreams[MAX_RESIDENT_STREAMS] = {};
  std::atomic<int> n_rstreams{1};
  std::atomic<uint32_t> next_rstream{0};
  // Kernel timing: a ring of event sets so that launches queue back to back; the host only waits
  // when it laps the ring.  ev[0..1] bracket the glob-resolve kernel, ev[2..3] the decision kernel.
  static constexpr int RING = 32;
  struct Slot { hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; bool pending = false; bool resolved = false; };
  Slot ring[RING];
  uint64_t next_slot = 0, launches = 0;
  double check_ms_sum = 0, resolve_ms_sum = 0; uint64_t timed = 0;
  std::mutex mu;
  std::mutex pool_mu;
  std::vector<std::pair<void*, size_t>> pool_free;   // idle device blocks of released batches
  // One-shot calls run on their own small set of contexts so that calls from different threads overlap.
  static constexpr int MAX_ONESHOT = 8;
  std::mutex ctx_mu;
  std::condition_variable ctx_cv;
  std::vector<OneShot*> ctx_idle;
  int ctx_count = 0;
  // the device flattener's per-table data (cbh_wire_host.h WireIndexHost), uploaded at load
  u64* w_tix = nullptr; u32* w_scope_of_sid = nullptr; WireCol* w_cols = nullptr; u8* w_col_keys = nullptr;
  u32* w_name_off = nullptr; u8* w_name_bytes = nullptr;
  // A batch of cbh_wire_flatten has a stream to itself while it lives (the call synchronises it several times: on a shared
  // stream every caller would wait for every other caller's work); idle ones are kept, at most MAX_WIRE_STREAMS are made.
  static constexpr int MAX_WIRE_STREAMS = 64;
  std::mutex wstream_mu; std::vector<hipStream_t> wstreams_idle, wstreams_all; int wstreams_made = 0;
  // ... and a page-locked staging block (the call's offsets, defaults and statistics cross PCIe from / to it: a pageable source makes
  // hipMemcpyAsync a blocking, staged copy)
  std::vector<std::pair<void*, size_t>> wpinned_idle;
  // The link's two directions, a stream each.  A stream's copies go to ONE copy engine, whatever their direction: with every slice
  // of a call uploading and downloading on its own stream the link carried one copy at a time (tools/pcie_duplex2.hip: 4 x (12 MB up,
  // 10 MB down) 1.55 ms a stream per slice, 0.94 ms with one upload stream and one download stream - the link is full duplex, the
  // engines are per stream).  The bulk copies of the wire road go here; a batch's own stream waits for / is waited for by events.
  hipStream_t up_stream = nullptr, down_stream = nullptr; bool link_streams_tried = false;
  std::vector<hipEvent_t> wevents_idle;
};

struct cbh_table {
  std::vector<Replica*> reps;
  size_t image_len = 0;
  std::vector<uint32_t> meta;
  std::atomic<int> refs{1};
  const char* bcast = "none";
  WireIndexHost wire;   // cbh_wire_flatten
};

struct cbh_device_batch {
  cbh_table* table = nullptr;
  Replica* rep = nullptr;
  BatchDev dev{};
  OutDev out{};
  hipStream_t stream = nullptr;   // the replica's resident stream this batch lives on
  KernelArgs* d_args = nullptr;   // device copy of the launch arguments
  KernelArgs last_args;           // what d_args currently holds
  bool have_args = false;
  u32 max_actions = 0, max_roles = 0;   // largest CBH_RQ_ACT_CNT / ROLE_CNT of the batch: select the kernel
  u32 wide_lo = 0, wide_hi = 0;         // BatchShape::wide_lo / wide_hi
  bool plain_tags = false;              // BatchShape::plain_tags
  std::vector<std::pair<void*, size_t>> allocs;   // (block, capacity) taken from the replica's pool
  // a batch the device flattened (cbh_wire_flatten): where the response's strings sit in the messages
  bool wire = false; bool own_wire_stream = false; u32* w_in_span = nullptr; u32* w_act_span = nullptr;
  void* w_pinned = nullptr; size_t w_pinned_cap = 0, w_pin_out_at = 0;
  ptrdiff_t w_pinned_delta = 0;   // device address of the page-locked block minus its host address (hipHostGetDevicePointer; 0 where both agree)
  hipEvent_t w_ev[2] = {nullptr, nullptr};   // (the link streams) upload landed / the outputs are written; download landed
  const u32* w_req_input = nullptr;   // the request words in INPUT order (dev.req_u32 may be the grouped copy)
  u32 trail_groups = 0; u32* trail_grp = nullptr;   // cbh_batch_set_trail: groups of out.eff_pol, the requests' groups
  const u32* w_inv = nullptr;         // grouped by route: input -> position of its per-request results; else null
  u64* w_edr_input = nullptr;         // scratch of cbh_result_download: the derived-role masks back in input order
  const u64* w_moff = nullptr; u32 w_dver_off = 0, w_dver_len = 0;   // (the device assembler reads the messages again)
  bool w_total_known = false; uint64_t w_total = 0; uint32_t w_out_errors = 0;   // cbh_wire_outputs ran its size / scan launches for this batch's current results
  u32* w_sizes = nullptr; u64* w_wavesum = nullptr; u64* w_waveoff = nullptr; WireOutStats* w_ostats = nullptr; u64* w_out_off = nullptr; u8* w_out_flags = nullptr;
};

static void replica_destroy(Replica* r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  for (int i = 1; i < Replica::MAX_RESIDENT_STREAMS; ++i) if (r->rstreams[i]) { (void)hipStreamSynchronize(r->rstreams[i]); (void)hipStreamDestroy(r->rstreams[i]); }
  if (r->stream) { (void)hipStreamSynchronize(r->stream); (void)hipStreamDestroy(r->stream); }
  for (auto& sl : r->ring) for (auto& e : sl.ev) if (e) (void)hipEventDestroy(e);
  for (hipStream_t ws : r->wstreams_all) { (void)hipStreamSynchronize(ws); (void)hipStreamDestroy(ws); }
  for (auto& pb : r->wpinned_idle) (void)hipHostFree(pb.first);
  for (hipStream_t ls : {r->up_stream, r->down_stream}) if (ls) { (void)hipStreamSynchronize(ls); (void)hipStreamDestroy(ls); }
  for (hipEvent_t e : r->wevents_idle) (void)hipEventDestroy(e);
  if (r->image && r->owns_image) (void)hipFree(r->image);
  for (void* p : {(void*)r->w_tix, (void*)r->w_scope_of_sid, (void*)r->w_cols, (void*)r->w_col_keys, (void*)r->w_name_off, (void*)r->w_name_bytes}) if (p) (void)hipFree(p);
  for (auto& a : r->pool_free) (void)hipFree(a.first);
  for (auto* c : r->ctx_idle) {
    for (auto& s : c->s) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    if (c->ev_setup) (void)hipEventDestroy(c->ev_setup);
    for (auto& e : c->ev_piece) if (e) (void)hipEventDestroy(e);
    if (c->h) (void)hipHostFree(c->h);
    if (c->d) (void)hipFree(c->d);
    delete c;
  }
  delete r;
}
static void table_destroy(cbh_table* t) {
  for (Replica* r : t->reps) replica_destroy(r);
  delete t;
}
// every entry point that works on a table holds a reference while it runs: the table outlives a
// concurrent cbh_table_release (the last reference frees it)
struct TableRef {
  cbh_table* t;
  explicit TableRef(cbh_table* t_) : t(t_) { t->refs.fetch_add(1, std::memory_order_relaxed); }
  ~TableRef() { if (t->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) table_destroy(t); }
};
extern "C" void cbh_table_retain(cbh_table* t) { if (t) t->refs.fetch_add(1, std::memory_order_relaxed); }
extern "C" void cbh_table_release(cbh_table* t) {
  if (t && t->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) table_destroy(t);
}

static int replica_finish(Replica* r) {
  HIPCHK(hipSetDevice(r->device));
  HIPCHK(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
  static const int n_streams = [] { const char* e = getenv("CBH_RESIDENT_STREAMS"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > Replica::MAX_RESIDENT_STREAMS ? Replica::MAX_RESIDENT_STREAMS : v); }();
  r->n_rstreams = n_streams;   // (default; cbh_table_set_resident_streams changes it for the batches uploaded afterwards)
  r->rstreams[0] = r->stream;
  for (int i = 1; i < Replica::MAX_RESIDENT_STREAMS; ++i) HIPCHK(hipStreamCreateWithFlags(&r->rstreams[i], hipStreamNonBlocking));
  for (auto& sl : r->ring) for (auto& e : sl.ev) HIPCHK(hipEventCreate(&e));
  return 0;
}

// the device flattener's view of the table (cbh_wire.h): built once from the image, a copy on every replica
static int wire_index_install(cbh_table* t, const uint8_t* host_image, size_t len) {
  if (const char* e = cbh_wire_index_build(t->wire, host_image, len, t->meta)) return fail(e);
  const WireIndexHost& w = t->wire;
  for (Replica* r : t->reps) {
    HIPCHK(hipSetDevice(r->device));
    HIPCHK(hipMalloc((void**)&r->w_tix, w.tix.size() * 8)); HIPCHK(hipMemcpy(r->w_tix, w.tix.data(), w.tix.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&r->w_scope_of_sid, w.scope_of_sid.size() * 4)); HIPCHK(hipMemcpy(r->w_scope_of_sid, w.scope_of_sid.data(), w.scope_of_sid.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&r->w_cols, w.cols.size() * sizeof(WireCol))); HIPCHK(hipMemcpy(r->w_cols, w.cols.data(), w.cols.size() * sizeof(WireCol), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&r->w_col_keys, w.col_keys.size())); HIPCHK(hipMemcpy(r->w_col_keys, w.col_keys.data(), w.col_keys.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&r->w_name_off, w.name_off.size() * 4)); HIPCHK(hipMemcpy(r->w_name_off, w.name_off.data(), w.name_off.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&r->w_name_bytes, w.name_bytes.size())); HIPCHK(hipMemcpy(r->w_name_bytes, w.name_bytes.data(), w.name_bytes.size(), hipMemcpyHostToDevice));
  }
  return 0;
}

static bool rccl_bind(Rccl& r, const std::vector<int>& devs) {   // under g_eng.mu
  if (r.tried) return r.ok;
  r.tried = true;
  const char* off = getenv("CBH_BCAST");
  if (off && !strcmp(off, "peer")) return false;
  for (size_t i = 0; i < devs.size(); ++i) for (size_t j = i + 1; j < devs.size(); ++j) if (devs[i] == devs[j]) return false;   // one rank per GPU
  r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!r.lib) return false;
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
  r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
  if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Broadcast) return false;
  r.comms.assign(devs.size(), nullptr);
  if (r.CommInitAll(r.comms.data(), (int)devs.size(), devs.data()) != 0) { r.comms.clear(); return false; }
  r.ok = true;
  return true;
}

// the image sits on reps[0]; put a copy on every other replica
static int broadcast_image(cbh_table* t) {
  const size_t len = t->image_len;
  for (size_t i = 1; i < t->reps.size(); ++i) {
    HIPCHK(hipSetDevice(t->reps[i]->device));
    HIPCHK(hipMalloc(&t->reps[i]->image, len));
  }
  bool done = false;
  {
    std::lock_guard<std::mutex> lk(g_eng.mu);
    Rccl& r = g_eng.rccl;
    if (rccl_bind(r, g_eng.devices) && r.comms.size() == t->reps.size()) {
      // one-time RCCL broadcast over xGMI from the first device (ncclUint8 = 1)
      bool ok = r.GroupStart() == 0;
      for (size_t i = 0; ok && i < t->reps.size(); ++i) {
        ok = hipSetDevice(t->reps[i]->device) == hipSuccess &&
             r.Broadcast(t->reps[0]->image, t->reps[i]->image, len, 1, 0, r.comms[i], t->reps[i]->stream) == 0;
      }
      ok = (r.GroupEnd() == 0) && ok;
      for (size_t i = 0; i < t->reps.size(); ++i) { (void)hipSetDevice(t->reps[i]->device); ok = (hipStreamSynchronize(t->reps[i]->stream) == hipSuccess) && ok; }
      if (ok) { done = true; t->bcast = "rccl"; } else (void)hipGetLastError();
    }
  }
  if (!done) {
    for (size_t i = 1; i < t->reps.size(); ++i)
      HIPCHK(hipMemcpyPeer(t->reps[i]->image, t->reps[i]->device, t->reps[0]->image, t->reps[0]->device, len));
    t->bcast = "peer-copy";
  }
  return 0;
}

extern "C" int cbh_table_load(const void* blob, size_t len, cbh_table** out) {
  std::vector<int> devs;
  { std::lock_guard<std::mutex> lk(g_eng.mu); if (!g_eng.inited) return fail("cbh_init has not been called"); devs = g_eng.devices; }
  if (!blob || !out) return fail("null argument");
  cbh_table* t = new (std::nothrow) cbh_table();
  if (!t) return fail("out of memory");
  t->image_len = len;
  auto bail = [&]() { table_destroy(t); return -1; };
  for (int d : devs) { Replica* r = new (std::nothrow) Replica(); if (!r) { fail("out of memory"); return bail(); } r->device = d; t->reps.push_back(r); }
  for (Replica* r : t->reps) if (replica_finish(r) != 0) return bail();
  Replica* r0 = t->reps[0];
  if (hipSetDevice(r0->device) != hipSuccess || hipMalloc(&r0->image, len) != hipSuccess) { fail("hipMalloc(table image) failed"); return bail(); }
  if (hipMemcpy(r0->image, blob, len, hipMemcpyHostToDevice) != hipSuccess) { fail("hipMemcpy(table image) failed"); return bail(); }
  if (t->reps.size() > 1 && broadcast_image(t) != 0) return bail();
  for (Replica* r : t->reps) {
    const char* e = cbh_parse_image(r->dev, t->meta, static_cast<const uint8_t*>(r->image), static_cast<const uint8_t*>(blob), len);
    if (e) { fail(e); return bail(); }
  }
  if (wire_index_install(t, static_cast<const uint8_t*>(blob), len) != 0) return bail();
  *out = t;
  return 0;
}

extern "C" int cbh_table_adopt_device_image(void* device_image, size_t len, cbh_table** out) {
  int dev0;
  { std::lock_guard<std::mutex> lk(g_eng.mu); if (!g_eng.inited) return fail("cbh_init has not been called"); dev0 = g_eng.devices[0]; }
  if (!device_image || !out) return fail("null argument");
  HIPCHK(hipSetDevice(dev0));
  std::vector<uint8_t> host(len);
  HIPCHK(hipMemcpy(host.data(), device_image, len, hipMemcpyDeviceToHost));
  cbh_table* t = new (std::nothrow) cbh_table();
  if (!t) return fail("out of memory");
  Replica* r = new (std::nothrow) Replica();
  if (!r) { delete t; return fail("out of memory"); }
  r->device = dev0; r->image = device_image; r->owns_image = false;
  t->reps.push_back(r); t->image_len = len;
  const char* e = cbh_parse_image(r->dev, t->meta, static_cast<const uint8_t*>(r->image), host.data(), len);
  if (e) { table_destroy(t); return fail(e); }
  if (replica_finish(r) != 0) { table_destroy(t); return -1; }
  if (wire_index_install(t, host.data(), len) != 0) { table_destroy(t); return -1; }
  *out = t;
  return 0;
}

extern "C" const char* cbh_table_broadcast_kind(const cbh_table* t) { return t ? t->bcast : "none"; }
extern "C" uint32_t cbh_table_num_strings(const cbh_table* t) { return t ? t->meta[CBH_M_NSTRINGS] : 0; }
extern "C" uint32_t cbh_table_num_columns(const cbh_table* t) { return t ? t->meta[CBH_M_NCOLUMNS] : 0; }
extern "C" uint64_t cbh_table_device_bytes(const cbh_table* t) { return t ? t->image_len : 0; }
extern "C" void* cbh_table_device_ptr(const cbh_table* t) { return t ? t->reps[0]->image : nullptr; }

// ---- batch validation (O(n_requests), both entry points) ---------------------------------------------------
// Offsets and counts the kernels index device memory with must lie inside the arrays they index.  String ids
// need no host pass: the kernels only compare them, or bound them before using one as an index.
struct BatchShape {
  u32 max_actions = 0, max_roles = 0; bool ascending = true;
  u32 wide_lo = 0, wide_hi = 0;   // the requests with more than CBH_W2_NA actions or CBH_W2_NR roles lie in [wide_lo, wide_hi)
  // Do the attribute columns hold plain scalars only - no int / uint (cross-type numerics) and no list / map (deep
  // equality)?  Then no classified leaf can need the shared evaluator and the flat kernel without that call decides
  // the batch (cbh_check_flat.h).  One pass over the tag bytes, made only where the answer selects a kernel and only
  // when the first launch is being prepared - by then the uploads are enqueued and the pass runs beside them.
  const uint8_t* tags = nullptr; size_t n_tags = 0;   // nullptr: the answer cannot matter (no flat kernel for this table / shape)
  uint32_t sens_cols = 0; size_t n_req = 0;           // only these columns [bit c: tags + c * n_req] can send a classified leaf to the evaluator (CBH_M_SENS_COLS)
  mutable std::atomic<int> plain{-1};                 // -1 not looked at yet (racing threads compute the same answer)
  bool plain_tags() const {
    int v = plain.load(std::memory_order_relaxed);
    if (v < 0) {
      static const bool force_any = getenv("CBH_FLAT_ANY") != nullptr;   // measurement / test aid: always the variant with the call
      bool hit = false;
      if (tags) for (uint32_t c = 0; c < 32 && !hit; ++c) if ((sens_cols >> c) & 1u) hit = has_int_or_container_tag(tags + (size_t)c * n_req, n_req);
      v = (!force_any && !hit) ? 1 : 0;
      plain.store(v, std::memory_order_relaxed);
    }
    return v == 1;
  }
  // tags 2, 3 (int, uint) and 6, 7 (list, map) are exactly the bytes x with (x & 0xFA) == 0x02: eight at a time
  static bool has_int_or_container_tag(const uint8_t* p, size_t n) {
    size_t i = 0; uint64_t hit = 0;
    for (; i + 8 <= n; i += 8) {
      uint64_t w; std::memcpy(&w, p + i, 8);
      const uint64_t z = (w & 0xFAFAFAFAFAFAFAFAull) ^ 0x0202020202020202ull;              // a zero byte where a tag matched
      hit |= (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull;
    }
    for (; i < n; ++i) hit |= (uint64_t)((p[i] & 0xFAu) == 0x02u);
    return hit != 0;
  }
};
// the O(1) part of validate_batch: the arrays a batch of these counts needs are there
static int validate_header(const cbh_table* t, const cbh_batch* in) {
  if (in->n_columns != t->meta[CBH_M_NCOLUMNS]) return fail("cbh_batch.n_columns does not match the table's column schema");
  const size_t NR = in->n_requests;
  if (NR && !in->req_u32) return fail("cbh_batch: a required array is NULL");
  if (in->n_tuples && !in->tuple_action) return fail("cbh_batch: a required array is NULL");
  if (in->n_roles && !in->roles) return fail("cbh_batch: a required array is NULL");
  if (NR && in->n_columns && (!in->col_tag || !in->col_val)) return fail("cbh_batch: a required array is NULL");
  if (in->heap_len && (!in->heap_tag || !in->heap_val)) return fail("cbh_batch: a required array is NULL");
  if (in->n_strings && (!in->str_off || !in->str_flags)) return fail("cbh_batch: a required array is NULL");
  if (in->str_bytes_len && !in->str_bytes) return fail("cbh_batch: a required array is NULL");
  return 0;
}
static int validate_batch(const cbh_table* t, const cbh_batch* in, BatchShape& sh) {
  if (validate_header(t, in) != 0) return -1;
  const size_t NR = in->n_requests;
  const u32* role_off = in->req_u32 + (size_t)CBH_RQ_ROLE_OFF * NR; const u32* role_cnt = in->req_u32 + (size_t)CBH_RQ_ROLE_CNT * NR;
  const u32* act_off = in->req_u32 + (size_t)CBH_RQ_ACT_OFF * NR; const u32* act_cnt = in->req_u32 + (size_t)CBH_RQ_ACT_CNT * NR;
  // (three passes without loop-carried dependences other than max / or reductions: the compiler vectorises them - this scan
  // sits on the path of every one-shot call, 250 000 requests at the headline size)
  u32 maxa = 0, maxr = 0; u32 bad = 0;
  const u64 n_roles = in->n_roles, n_tuples = in->n_tuples;
  for (size_t r = 0; r < NR; ++r) {
    maxa = act_cnt[r] > maxa ? act_cnt[r] : maxa;
    maxr = role_cnt[r] > maxr ? role_cnt[r] : maxr;
    bad |= (u32)((u64)role_off[r] + role_cnt[r] > n_roles) | (u32)((u64)act_off[r] + act_cnt[r] > n_tuples);
  }
  u32 unordered = 0;
  for (size_t r = 1; r < NR; ++r) unordered |= (u32)((u64)act_off[r] < (u64)act_off[r - 1] + act_cnt[r - 1]);
  u32 wlo = 0xFFFFFFFFu, whi = 0;
  if (maxa > CBH_W2_NA || maxr > CBH_W2_NR)   // where the requests wider than the walk's base shape lie (rare: found in a pass of its own)
    for (size_t r = 0; r < NR; ++r)
      if (act_cnt[r] > CBH_W2_NA || role_cnt[r] > CBH_W2_NR) { if (wlo == 0xFFFFFFFFu) wlo = (u32)r; whi = (u32)r + 1; }
  if (maxa > CBH_MAX_ACTIONS_PER_REQUEST) return fail("cbh_batch: a request carries more than CBH_MAX_ACTIONS_PER_REQUEST actions");
  if (bad) return fail("cbh_batch: a request's role or action slice lies outside the batch");
  if (in->n_strings && in->str_off[in->n_strings] > in->str_bytes_len) return fail("cbh_batch: string offsets exceed str_bytes_len");
  sh.max_actions = maxa; sh.max_roles = maxr; sh.ascending = !unordered;
  sh.wide_lo = whi ? wlo : 0; sh.wide_hi = whi;
  sh.tags = nullptr; sh.n_tags = 0; sh.plain.store(-1, std::memory_order_relaxed);
  if (((t->meta[CBH_M_FLAGS] & CBH_MF_FLAT) && maxa <= 4 && maxr <= 4) ||
      ((t->meta[CBH_M_FLAGS] & CBH_MF_WALK2) && t->meta[CBH_M_GSLOTS_ALL] > t->meta[CBH_M_GSLOTS_GENERIC])) { sh.tags = in->col_tag; sh.n_tags = (size_t)in->n_columns * NR; }
  sh.sens_cols = t->meta[CBH_M_SENS_COLS]; sh.n_req = NR;
  if (in->n_columns < 32) sh.sens_cols &= (1u << in->n_columns) - 1u;
  return 0;
}

// ---- resident path ----------------------------------------------------------------------------------------
// Device buffers of batches come from a per-replica pool of power-of-two blocks: a small synchronous
// CheckResources round trip must not pay ~17 hipMalloc / hipFree pairs (each hipFree also
// synchronises the device).  Blocks go back to the pool on cbh_batch_release and to the driver when the
// table goes.
static int pool_alloc(cbh_device_batch* b, size_t bytes, void** out) {
  size_t cap = 256;
  while (cap < bytes) cap <<= 1;
  Replica* r = b->rep;
  {
    std::lock_guard<std::mutex> lk(r->pool_mu);
    for (size_t i = 0; i < r->pool_free.size(); ++i)
      if (r->pool_free[i].second == cap) {
        *out = r->pool_free[i].first;
        r->pool_free[i] = r->pool_free.back(); r->pool_free.pop_back();
        b->allocs.push_back({*out, cap});
        return 0;
      }
  }
  HIPCHK(hipMalloc(out, cap));
  b->allocs.push_back({*out, cap});
  return 0;
}

extern "C" void cbh_batch_release(cbh_device_batch* b) {
  if (!b) return;
  cbh_table* t = b->table;
  (void)hipSetDevice(b->rep->device); (void)hipStreamSynchronize(b->stream ? b->stream : b->rep->stream);
  // a copy of this batch's outputs may still run on the replica's SHARED download stream (cbh_wire_outputs left early on an
  // error between the copy's enqueue and its wait): its block must not go back to the pool - and its event not to another
  // batch - before the copy has landed.  The event was recorded behind the copy; an event never recorded is complete.
  if (b->w_ev[1]) { if (hipEventSynchronize(b->w_ev[1]) != hipSuccess) (void)hipGetLastError(); }
  {
    std::lock_guard<std::mutex> lk(b->rep->pool_mu);
    for (auto& a : b->allocs) b->rep->pool_free.push_back(a);
  }
  if (b->own_wire_stream || b->w_pinned || b->w_ev[0] || b->w_ev[1]) {
    std::lock_guard<std::mutex> lk(b->rep->wstream_mu);
    if (b->own_wire_stream) b->rep->wstreams_idle.push_back(b->stream);
    if (b->w_pinned) b->rep->wpinned_idle.push_back({b->w_pinned, b->w_pinned_cap});
    for (hipEvent_t e : b->w_ev) if (e) b->rep->wevents_idle.push_back(e);
  }
  delete b;
  cbh_table_release(t);   // the reference the batch held
}

template <typename T>
static int up(cbh_device_batch* b, const T*& dst, const T* src, size_t n, hipStream_t s) {
  dst = nullptr;
  size_t bytes = (n ? n : 1) * sizeof(T);
  void* p = nullptr;
  if (pool_alloc(b, bytes, &p) != 0) return -1;
  if (n) HIPCHK(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
  dst = static_cast<const T*>(p);
  return 0;
}
template <typename T>
static int dalloc(cbh_device_batch* b, T*& dst, size_t n) {
  void* p = nullptr;
  if (pool_alloc(b, (n ? n : 1) * sizeof(T), &p) != 0) return -1;
  dst = static_cast<T*>(p);
  return 0;
}

extern "C" int cbh_batch_upload_on(cbh_table* t, uint32_t device_index, const cbh_batch* in, cbh_device_batch** out) {
  if (!t || !in || !out) return fail("null argument");
  if (device_index >= t->reps.size()) return fail("device index out of range");
  BatchShape sh;
  if (validate_batch(t, in, sh) != 0) return -1;
  Replica* rep = t->reps[device_index];
  HIPCHK(hipSetDevice(rep->device));
  cbh_device_batch* b = new (std::nothrow) cbh_device_batch();
  if (!b) return fail("out of memory");
  cbh_table_retain(t);
  b->table = t; b->rep = rep; b->max_actions = sh.max_actions; b->max_roles = sh.max_roles; b->plain_tags = sh.plain_tags();
  b->wide_lo = sh.wide_lo; b->wide_hi = sh.wide_hi;
  BatchDev& d = b->dev;
  d.n_requests = in->n_requests; d.n_tuples = in->n_tuples; d.n_roles = in->n_roles;
  d.n_columns = in->n_columns; d.n_strings = in->n_strings; d.heap_len = in->heap_len;
  d.req_lo = 0; d.req_hi = in->n_requests;
  b->stream = rep->rstreams[rep->next_rstream.fetch_add(1, std::memory_order_relaxed) % (uint32_t)rep->n_rstreams.load(std::memory_order_relaxed)];
  hipStream_t s = b->stream;
  const size_t NR = in->n_requests;
  int rc = 0;
  rc |= up(b, d.req_u32, in->req_u32, (size_t)CBH_RQ_NFIELDS * NR, s);
  rc |= up(b, d.roles, in->roles, in->n_roles, s);
  d.tuple_req = nullptr;   // informational on the host side; no kernel reads it
  rc |= up(b, d.tuple_action, in->tuple_action, in->n_tuples, s);
  rc |= up(b, d.col_tag, in->col_tag, (size_t)in->n_columns * NR, s);
  rc |= up(b, d.col_val, in->col_val, (size_t)in->n_columns * NR, s);
  rc |= up(b, d.heap_tag, in->heap_tag, in->heap_len, s);
  rc |= up(b, d.heap_val, in->heap_val, in->heap_len, s);
  rc |= up(b, d.str_off, in->str_off, in->n_strings ? (size_t)in->n_strings + 1 : 0, s);
  rc |= up(b, d.str_bytes, in->str_bytes, in->str_bytes_len, s);
  rc |= up(b, d.str_flags, in->str_flags, in->n_strings, s);
  rc |= dalloc(b, d.gbits, (size_t)3 * in->n_strings);
  d.n_gwords = (rep->dev.flags & CBH_MF_WALK2) ? w2_gwords(rep->dev.gslots_generic, rep->dev.gslots_all, b->plain_tags) : 0;
  d.n_gslots = 0;   // per launch (launch_plan)
  if (d.n_gwords) rc |= dalloc(b, d.gres, (size_t)d.n_gwords * NR); else d.gres = nullptr;
  rc |= dalloc(b, b->out.effect, in->n_tuples);
  rc |= dalloc(b, b->out.policy, in->n_tuples);
  rc |= dalloc(b, b->out.scope, in->n_tuples);
  rc |= dalloc(b, b->out.status, in->n_tuples);
  rc |= dalloc(b, b->out.edr, NR);
  rc |= dalloc(b, b->d_args, 1);
  if (rc != 0) { cbh_batch_release(b); return -1; }
  // glob bits of the batch-local strings: all zero unless the table has automata to run (then
  // cbh_check_resident overwrites every word on each launch)
  if (in->n_strings && hipMemsetAsync(d.gbits, 0, (size_t)3 * in->n_strings * sizeof(u64), s) != hipSuccess) {
    cbh_batch_release(b); return fail("upload failed");
  }
  if (hipStreamSynchronize(s) != hipSuccess) { cbh_batch_release(b); return fail("upload failed"); }
  *out = b;
  return 0;
}
extern "C" int cbh_batch_upload(cbh_table* t, const cbh_batch* in, cbh_device_batch** out) { return cbh_batch_upload_on(t, 0, in, out); }

static void collect_slot(Replica* r, Replica::Slot& sl) {   // the slot's last event has completed
  if (!sl.pending) return;
  float a = 0, c = 0;
  if (sl.resolved && hipEventElapsedTime(&a, sl.ev[0], sl.ev[1]) != hipSuccess) a = 0;
  if (hipEventElapsedTime(&c, sl.ev[2], sl.ev[3]) == hipSuccess) {
    r->resolve_ms_sum += a; r->check_ms_sum += c; r->timed += 1;
  }
  sl.pending = false;
}
static void collect_times(Replica* r) {   // after the stream has been synchronised
  for (auto& sl : r->ring) collect_slot(r, sl);
}

// CBH_NO_FLAT=1 / CBH_NO_WALK2=1 / CBH_NO_WALK2_WIDE=1 (measurement aids): leave the flat kernels / cbh_walk2_kernel out of the choice
static CbhPlan plan_for(const TableDev& dev, u32 max_actions, u32 max_roles, bool plain_tags, u32 eval_flags) {
  static const bool no_flat = getenv("CBH_NO_FLAT") != nullptr, no_walk2 = getenv("CBH_NO_WALK2") != nullptr;
  static const bool no_walk2_wide = getenv("CBH_NO_WALK2_WIDE") != nullptr;   // (measurement aid: requests with five to eight roles on the general walk, as before the wider shape)
  const bool has_globs = (dev.nfa_words[0] | dev.nfa_words[1] | dev.nfa_words[2]) != 0 || (dev.flags & CBH_MF_HAS_ANY_PATTERN);
  static const bool force_staged = getenv("CBH_FORCE_STAGED") != nullptr;   // (tests: the staged record walk on tables of any size)
  return cbh_plan(dev.flags, dev.n_dr, has_globs, dev.gslots_generic, dev.gslots_all, max_actions, max_roles, plain_tags, eval_flags, no_flat, no_walk2,
                  force_staged ? 0xFFFFFFFFu : dev.max_bucket, no_walk2_wide, cbh_flat_use_masks(dev.segs, dev.max_bucket));
}
// (on by default since round 5: C5 11.8 -> 12.4 G decisions/s, C5W 7.61 -> 7.67, profiles/r05_presplit_ab.txt; CBH_PRE_SPLIT=0: the fused pre-pass)
static bool pre_split_on() { static const bool on = [] { const char* e = getenv("CBH_PRE_SPLIT"); return e ? atoi(e) != 0 : true; }(); return on; }
// Does the packed form of the column cache's tags (cbh_vm.h CBH_CC_DWORDS) let a CU hold more workgroups of `fn` than the wide one?
// The runtime's occupancy figure for the kernel at either LDS size, kept per (kernel, size).  CBH_PACKED_TAGS=0/1 (tests,
// measurement): never / always.
static bool packed_tags_pay(cbh_check_kernel_fn fn, u32 threads, size_t lds_wide, size_t lds_packed) {
  static const int forced = [] { const char* e = getenv("CBH_PACKED_TAGS"); return e ? atoi(e) : -1; }();
  if (forced >= 0) return forced != 0;
  if (lds_packed >= lds_wide) return false;
  static std::mutex mu;
  static std::map<std::tuple<const void*, u32, size_t>, int> memo;
  auto blocks = [&](size_t lds) {
    const auto key = std::make_tuple((const void*)fn, threads, lds);
    std::lock_guard<std::mutex> g(mu);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)fn, (int)threads, lds) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    memo.emplace(key, n);
    return n;
  };
  return blocks(lds_packed) > blocks(lds_wide);
}
// the launches that decide the requests [lo, hi) of `ka.b`; [wide_lo, wide_hi) = where the batch's requests wider than
// cbh_walk2_kernel's shape lie (empty: none)
static void launch_plan(const CbhPlan& pl, const TableDev& dev, KernelArgs ka, const KernelArgs* d_args, u32 lo, u32 hi, u32 wide_lo, u32 wide_hi,
                        size_t pad, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
  if (hi <= lo) return;
  ka.b.req_lo = lo; ka.b.req_hi = hi;
  ka.flags &= ~(u32)CBH_FI_MASK;
  const u32 n = hi - lo;
  // (timed launches: the start event rides on the first kernel of the plan, the stop event on the last - the figure is the
  // whole plan's, gaps between its kernels included)
  // `lds_of(packed)`: the launch's dynamic LDS with the column cache's tags in either form (cbh_vm.h CBH_CC_DWORDS); the packed form
  // where it lets a CU hold more workgroups of this kernel
  auto go = [&](cbh_check_kernel_fn fn, u32 grid, u32 threads, auto lds_of, const KernelArgs& a0, bool last) {
    const size_t wide = lds_of(false), packed = lds_of(true);
    const bool use_packed = packed_tags_pay(fn, threads, wide, packed);
    const size_t lds = use_packed ? packed : wide;
    KernelArgs a = a0;
    if (use_packed) a.flags |= CBH_FI_PACKED_TAGS;
    if (ev0 || (ev1 && last)) { hipExtLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds, s, ev0, last ? ev1 : nullptr, 0, a, d_args); ev0 = nullptr; }
    else hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds, s, a, d_args);
  };
  auto plan_lds = [&](bool pre, u32 na, size_t extra) {
    return [=, &pl, &dev, &ka](bool packed) { return cbh_plan_lds(pl, dev.flags, dev.max_depth, dev.n_scopes, dev.K, ka.b.n_columns, dev.inline_cols, dev.n_dr, pre, na, packed) + extra; };
  };
  if (pl.kind == 2) {
    const u32 wlo = std::max(lo, wide_lo), whi = std::min(hi, wide_hi);   // where the requests wider than the base shape lie
    if (pl.wide_kernel) {   // the few requests wider than the walk's shapes: the general walk, on the lanes the walks below leave alone
      KernelArgs kw = ka;
      kw.b.req_lo = wlo; kw.b.req_hi = whi;
      kw.flags |= (pl.walk_wide || pl.walk_awide) ? CBH_FI_ONLY_WIDER : CBH_FI_ONLY_WIDE;
      if (whi > wlo) go(pl.wide_kernel, (whi - wlo + CBH_BLOCK - 1) / CBH_BLOCK, CBH_BLOCK, [&](bool packed) { return cbh_general_lds(dev.flags, ka.b.n_columns, packed); }, kw, false);
    }
    if (pl.wide_kernel || pl.walk_wide || pl.walk_awide) ka.flags |= CBH_FI_SKIP_WIDE;
    ka.b.n_gwords = ka.b.gres ? pl.n_gwords : 0; ka.b.n_gslots = ka.b.gres ? pl.n_gslots : 0;
    // the requests with five to eight roles / nine to sixteen actions: the walk's wider forms (and their pre-passes), over their part of the batch
    for (int shape = 1; shape <= 2 && whi > wlo; ++shape) {
      if (!(shape == 1 ? pl.walk_wide : pl.walk_awide)) continue;
      KernelArgs kv = ka;
      kv.b.req_lo = wlo; kv.b.req_hi = whi;
      const u32 na = shape == 1 ? CBH_W2_NA : CBH_W2_AWIDE_NA;
      if (kv.b.n_gwords)
        go(shape == 1 ? cbh_walk2_pre_wide_kernel : cbh_walk2_pre_awide_kernel, (whi - wlo + CBH_BLOCK - 1) / CBH_BLOCK, CBH_BLOCK,
           plan_lds(true, na, 0), kv, false);
      go(shape == 1 ? (pl.trail ? cbh_walk2_wide_trail_kernel : cbh_walk2_wide_kernel) : (pl.trail ? cbh_walk2_awide_trail_kernel : cbh_walk2_awide_kernel),
         (whi - wlo + pl.threads - 1) / pl.threads, pl.threads, plan_lds(false, na, 0), kv, false);
    }
    if (ka.b.n_gwords && ka.b.site_cnt && ka.b.site_cap >= n && pre_split_on()) {
      // the evaluation sites in two kernels: who reaches which site (the walk's registers), then the sites' lists (the interpreter's)
      (void)hipMemsetAsync(ka.b.site_cnt, 0, (size_t)ka.b.n_gslots * 4, s);
      go(cbh_walk2_collect_kernel, (n + CBH_BLOCK - 1) / CBH_BLOCK, CBH_BLOCK,
         [&](bool packed) { return w2_lds_bytes(w2_layout(dev.inline_cols, false, dev.max_depth, dev.n_scopes, true, pl.n_gwords, dev.K, dev.n_dr, CBH_W2_NA, packed), 1u); }, ka, false);
      go(cbh_walk2_interp_kernel, ka.b.n_gslots * ((n + CBH_BLOCK - 1) / CBH_BLOCK), CBH_BLOCK, plan_lds(true, CBH_W2_NA, 0), ka, false);
    } else if (ka.b.n_gwords)   // the evaluation sites first: their results are what the walk reads
      go(cbh_walk2_pre_kernel, (n + CBH_BLOCK - 1) / CBH_BLOCK, CBH_BLOCK, plan_lds(true, CBH_W2_NA, 0), ka, false);
  }
  static const bool pre_only = getenv("CBH_PRE_ONLY") != nullptr;   // measurement aid (profiling build): the pre-pass alone
  if (pre_only && pl.kind == 2) return;
  go(pl.kernel, (n + pl.threads - 1) / pl.threads, pl.threads, plan_lds(false, CBH_W2_NA, pad), ka, true);
}
// CBH_LDS_PAD=<bytes> (measurement aid): extra dynamic LDS per workgroup of the resident launches, to hold the occupancy down
static size_t lds_pad() { static const size_t pad = [] { const char* e = getenv("CBH_LDS_PAD"); return e ? (size_t)atol(e) : (size_t)0; }(); return pad; }
static u32 nfa_maxw(const TableDev& d) { return std::max(std::max(d.nfa_words[0], d.nfa_words[1]), d.nfa_words[2]); }
static size_t check_lds_bytes(const BatchDev& d, u32 table_flags) {   // the column cache (tags in the wide form)
  const u32 ncc = d.n_columns < CBH_CACHE_COLS ? d.n_columns : CBH_CACHE_COLS;
  // ... and, for a table whose programs build lists, the lanes' arenas behind it (cbh_vm.h arena_vals)
  return (size_t)CBH_CC_DWORDS(ncc, false) * 4 + ((table_flags & CBH_MF_NEEDS_ARENA) ? (size_t)CBH_ARENA_ENTRIES * CBH_BLOCK * 9 : 0);
}

extern "C" int cbh_check_resident(cbh_table* t, cbh_device_batch* b, const cbh_params* p) {
  if (!t || !b || !p) return fail("null argument");
  if (b->table != t) return fail("batch was uploaded for a different table");
  Replica* rep = b->rep;
  std::lock_guard<std::mutex> lk(rep->mu);
  HIPCHK(hipSetDevice(rep->device));
  hipStream_t s = b->stream;
  b->w_total_known = false;   // (the sizes cbh_wire_outputs computed belong to the results this launch replaces)
  // The walk's pre-pass as a collector and an interpreter over per-site lists - the lists
  // live with the batch (slots x requests items).  Tables whose programs read runtime.effectiveDerivedRoles keep the fused pre-pass.
  if (pre_split_on() && !b->dev.site_cnt && b->dev.n_requests && (rep->dev.flags & CBH_MF_WALK2) && rep->dev.gslots_all &&
      !(rep->dev.flags & CBH_MF_USES_RUNTIME_EDR) && b->dev.gres) {
    u32* cnt = nullptr; u64* list = nullptr;
    if (dalloc(b, cnt, (size_t)rep->dev.gslots_all) != 0 || dalloc(b, list, (size_t)rep->dev.gslots_all * b->dev.n_requests) != 0) return -1;
    b->dev.site_cnt = cnt; b->dev.site_list = list; b->dev.site_cap = b->dev.n_requests;
  }
  // Kernel durations come from the dispatches' own begin / end timestamps (hipExtLaunchKernelGGL
  // with start / stop events: what rocprofv3's kernel trace reads too), not from event-record
  // packets placed around them, which would sit between back-to-back launches and add their own
  // latency to the figure.
  // Every fourth launch is timed (and the first few, so that a short run has a figure): a
  // timestamped dispatch costs the queue a little more than a plain one.
  const uint64_t launch_no = rep->launches++;
  const bool timed = launch_no < 4 || (launch_no & 3) == 0;
  Replica::Slot scratch_slot;
  Replica::Slot& sl = timed ? rep->ring[rep->next_slot++ % Replica::RING] : scratch_slot;
  if (timed && sl.pending) { HIPCHK(hipEventSynchronize(sl.ev[3])); collect_slot(rep, sl); }
  const BatchDev& d = b->dev;
  {
    // launch arguments live in device memory; re-sent only when they change (the kernel itself
    // writes every output word of every request, so nothing needs clearing between launches)
    KernelArgs ka;
    std::memset(&ka, 0, sizeof(ka));
    ka.t = rep->dev; ka.b = d; ka.o = b->out; ka.now_ns = p->now_ns; ka.flags = p->flags & ~(u32)CBH_FI_MASK;
    if (!b->have_args || std::memcmp(&ka, &b->last_args, sizeof(ka)) != 0) {
      b->last_args = ka; b->have_args = true;
      HIPCHK(hipMemcpyAsync(b->d_args, &b->last_args, sizeof(ka), hipMemcpyHostToDevice, s));
    }
  }
  // batch-local strings against the table's glob automata; a table without globs has nothing to
  // resolve (the bits were zeroed once at upload)
  const u32 maxw = nfa_maxw(rep->dev);
  sl.resolved = d.n_strings && maxw;
  if (sl.resolved) {
    const u32 grid = (d.n_strings + CBH_BLOCK - 1) / CBH_BLOCK;
    const size_t lds = (size_t)(2 + 512) * maxw * sizeof(u64);
    if (timed) hipExtLaunchKernelGGL(cbh_resolve_globs_kernel, dim3(grid), dim3(CBH_BLOCK), lds, s, sl.ev[0], sl.ev[1], 0, rep->dev, d);
    else hipLaunchKernelGGL(cbh_resolve_globs_kernel, dim3(grid), dim3(CBH_BLOCK), lds, s, rep->dev, d);
  }
  sl.pending = false;
  if (d.n_requests) {
    const CbhPlan pl = plan_for(rep->dev, b->max_actions, b->max_roles, b->plain_tags, p->flags);
    launch_plan(pl, rep->dev, b->last_args, (const KernelArgs*)b->d_args, 0, d.n_requests, b->wide_lo, b->wide_hi, lds_pad(), s, timed ? sl.ev[2] : nullptr, timed ? sl.ev[3] : nullptr);
    sl.pending = timed;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// A sweep: cbh_check_resident for each of `n` resident batches of the table, in order, in one call (what a server's dispatch loop
// does between two polls of its queue; saves the caller n - 1 crossings of the boundary).
extern "C" int cbh_check_resident_many(cbh_table* t, cbh_device_batch* const* bs, uint32_t n, const cbh_params* p) {
  if (!t || (!bs && n) || !p) return fail("null argument");
  for (uint32_t i = 0; i < n; ++i) if (cbh_check_resident(t, bs[i], p) != 0) return -1;
  return 0;
}

// How many of the replica's resident streams batches uploaded FROM NOW ON are dealt to (1 .. 4; a batch keeps its stream).
// 1 = every launch queues behind the one before it: the setting for timing one kernel by itself.
extern "C" int cbh_table_set_resident_streams(cbh_table* t, uint32_t n) {
  if (!t) return fail("null argument");
  if (n < 1 || n > (uint32_t)Replica::MAX_RESIDENT_STREAMS) return fail("resident streams: 1 .. 8");
  for (Replica* rep : t->reps) { rep->n_rstreams.store((int)n); rep->next_rstream.store(0); }
  return 0;
}
extern "C" uint32_t cbh_table_resident_streams(const cbh_table* t) { return t && !t->reps.empty() ? (uint32_t)t->reps[0]->n_rstreams.load() : 0u; }

// Which kernels cbh_check_resident launches for this batch (measurement aid: bench.py names them in its line).
extern "C" const char* cbh_plan_describe(cbh_table* t, cbh_device_batch* b, const cbh_params* p) {
  static thread_local std::string s;
  if (!t || !b || !p) return "";
  const CbhPlan pl = plan_for(b->rep->dev, b->max_actions, b->max_roles, b->plain_tags, p->flags & ~(u32)CBH_FI_MASK);
  // (the pre-pass's form: cbh_check_resident's own condition for giving the batch its site lists)
  const Replica* rep = b->rep;
  const bool pre_split = pre_split_on() && b->dev.n_requests && (rep->dev.flags & CBH_MF_WALK2) && rep->dev.gslots_all && !(rep->dev.flags & CBH_MF_USES_RUNTIME_EDR) && b->dev.gres;
  if (pl.kind == 2) s = std::string(pl.wide_kernel ? "cbh_check_kernel*(wide requests)+" : "") + (pl.walk_wide ? (pl.trail ? "cbh_walk2_wide_trail_kernel(5-8 roles)+" : "cbh_walk2_wide_kernel(5-8 roles)+") : "") + (pl.walk_awide ? (pl.trail ? "cbh_walk2_awide_trail_kernel(9-16 actions)+" : "cbh_walk2_awide_kernel(9-16 actions)+") : "") + (pl.n_gwords && b->dev.gres ? (pre_split ? "cbh_walk2_collect_kernel+cbh_walk2_interp_kernel+" : "cbh_walk2_pre_kernel+") : "") + (pl.trail ? "cbh_walk2_trail_kernel" : "cbh_walk2_kernel");
  else if (pl.kind == 1 && cbh_is_flat_trail_kernel(pl.kernel)) s = cbh_is_mask_kernel(pl.kernel) ? "cbh_check_flat_trail_kernel*_masks" : "cbh_check_flat_trail_kernel*";
  else if (pl.kind == 0 && pl.kernel == cbh_check_trail_kernel) s = "cbh_check_trail_kernel";
  else if (pl.kind == 1) s = pl.kernel == cbh_check_flat_kernel ? "cbh_check_flat_kernel" : pl.kernel == cbh_check_flat_kernel_dr ? "cbh_check_flat_kernel_dr" : pl.kernel == cbh_check_flat_kernel_any ? "cbh_check_flat_kernel_any"
                           : pl.kernel == cbh_check_flat_kernel_staged ? "cbh_check_flat_kernel_staged" : pl.kernel == cbh_check_flat_kernel_masks ? "cbh_check_flat_kernel_masks"
                           : pl.kernel == cbh_check_flat_kernel_any_masks ? "cbh_check_flat_kernel_any_masks" : "cbh_check_flat_kernel_any_staged";
  else s = "cbh_check_kernel*";
  return s.c_str();
}

extern "C" int cbh_synchronize(cbh_table* t) {
  if (!t) return fail("null argument");
  for (Replica* rep : t->reps) {
    std::lock_guard<std::mutex> lk(rep->mu);
    HIPCHK(hipSetDevice(rep->device));
    for (int i = 0; i < Replica::MAX_RESIDENT_STREAMS; ++i) if (rep->rstreams[i]) HIPCHK(hipStreamSynchronize(rep->rstreams[i]));
    { std::vector<hipStream_t> ws; { std::lock_guard<std::mutex> lw(rep->wstream_mu); ws = rep->wstreams_all; } for (hipStream_t x : ws) HIPCHK(hipStreamSynchronize(x)); }
    collect_times(rep);
  }
  return 0;
}

extern "C" int cbh_kernel_time_ms(cbh_table* t, float* check_ms, float* resolve_ms) {
  if (!t) return fail("null argument");
  double c = 0, r = 0; uint64_t n = 0;
  for (Replica* rep : t->reps) {
    std::lock_guard<std::mutex> lk(rep->mu);
    c += rep->check_ms_sum; r += rep->resolve_ms_sum; n += rep->timed;
    rep->check_ms_sum = rep->resolve_ms_sum = 0; rep->timed = 0;
  }
  if (n == 0) return fail("no timed launches yet");
  if (check_ms) *check_ms = (float)(c / (double)n);
  if (resolve_ms) *resolve_ms = (float)(r / (double)n);
  return 0;
}

extern "C" int cbh_result_download(cbh_table* t, cbh_device_batch* b, cbh_result* out) {
  if (!t || !b || !out) return fail("null argument");
  if (b->dev.n_tuples && !out->effect) return fail("cbh_result.effect is required");
  Replica* rep = b->rep;
  std::lock_guard<std::mutex> lk(rep->mu);
  HIPCHK(hipSetDevice(rep->device));
  hipStream_t s = b->stream;
  const BatchDev& d = b->dev;
  if (d.n_tuples) HIPCHK(hipMemcpyAsync(out->effect, b->out.effect, d.n_tuples, hipMemcpyDeviceToHost, s));
  if (out->policy && d.n_tuples) HIPCHK(hipMemcpyAsync(out->policy, b->out.policy, (size_t)d.n_tuples * 4, hipMemcpyDeviceToHost, s));
  if (out->scope && d.n_tuples) HIPCHK(hipMemcpyAsync(out->scope, b->out.scope, (size_t)d.n_tuples * 4, hipMemcpyDeviceToHost, s));
  if (out->status && d.n_tuples) HIPCHK(hipMemcpyAsync(out->status, b->out.status, d.n_tuples, hipMemcpyDeviceToHost, s));
  if (out->edr_mask && d.n_requests) {
    const u64* src = b->out.edr;
    if (b->w_inv) {   // a batch grouped by route: the masks follow their requests back to input order
      if (!b->w_edr_input && dalloc(b, b->w_edr_input, (size_t)d.n_requests) != 0) return -1;
      WireUnsortArgs ua; ua.edr_grouped = b->out.edr; ua.inv = b->w_inv; ua.edr_input = b->w_edr_input; ua.n = d.n_requests; ua.pad = 0;
      hipLaunchKernelGGL(cbh_wire_unsort_edr_kernel, dim3((d.n_requests + 255u) / 256u), dim3(256), 0, s, ua);
      src = b->w_edr_input;
    }
    HIPCHK(hipMemcpyAsync(out->edr_mask, src, (size_t)d.n_requests * 8, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  collect_times(rep);
  return 0;
}


// ---- engine.Check's second return value: the policies a call touched (AuditTrail.EffectivePolicies) ---------------------------
extern "C" uint32_t cbh_table_num_policies(const cbh_table* t) { return t ? t->wire.n_policies : 0; }
extern "C" int cbh_table_policy_key(const cbh_table* t, uint32_t i, const char** key, uint32_t* len) {
  if (!t || !key || !len) return fail("null argument");
  if (i >= t->wire.n_policies) return fail("policy index out of range");
  *key = reinterpret_cast<const char*>(t->wire.name_bytes.data()) + t->wire.name_off[i];
  *len = t->wire.name_off[i + 1] - t->wire.name_off[i];
  return 0;
}
// cbh_check_batch with the trail: the batch goes through the resident path of device 0 (upload, the general walk with
// CBH_F_WANT_EFFECTIVE_POLICIES, download) - the walk that iterates a request's roles one after the other as check.go:208-442
// does, so that "touched" means what it means there.
// The trail of a RESIDENT batch: cbh_batch_set_trail says which group (engine.Check call) every request of the batch belongs to and
// gives the batch its masks; from then on a cbh_check_resident with CBH_F_WANT_EFFECTIVE_POLICIES ORs into them, cbh_trail_download
// reads them (and cbh_batch_set_trail again clears them).  group_of_request: host memory, DEVICE order of the batch, NULL = one group.
extern "C" int cbh_batch_set_trail(cbh_table* t, cbh_device_batch* b, const uint32_t* group_of_request, uint32_t n_groups) {
  if (!t || !b) return fail("null argument");
  if (b->table != t) return fail("batch was uploaded for a different table");
  if (n_groups == 0) n_groups = 1;
  const u32 n = b->dev.n_requests;
  if (group_of_request) for (u32 r = 0; r < n; ++r) if (group_of_request[r] >= n_groups) return fail("cbh_batch_set_trail: group index out of range");
  Replica* rep = b->rep;
  HIPCHK(hipSetDevice(rep->device));
  hipStream_t s = b->stream;
  const u32 words = (t->wire.n_policies + 31u) / 32u;
  const size_t ep_n = (size_t)n_groups * (words ? words : 1u);
  if (!b->out.eff_pol || b->trail_groups != n_groups) {
    u32* d_ep = nullptr;
    if (dalloc(b, d_ep, ep_n) != 0) return -1;
    b->out.eff_pol = d_ep; b->out.ep_words = words; b->trail_groups = n_groups;
  }
  HIPCHK(hipMemsetAsync(b->out.eff_pol, 0, ep_n * 4, s));
  if (group_of_request && n) {
    if (!b->trail_grp && dalloc(b, b->trail_grp, (size_t)n) != 0) return -1;   // (kept: a batch is asked again and again)
    HIPCHK(hipMemcpyAsync(b->trail_grp, group_of_request, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));   // (a pageable source)
  }
  b->dev.ep_group = (group_of_request && n) ? b->trail_grp : nullptr;
  return 0;
}
extern "C" int cbh_trail_download(cbh_table* t, cbh_device_batch* b, uint32_t* effective_policies) {
  if (!t || !b || !effective_policies) return fail("null argument");
  if (!b->out.eff_pol) return fail("cbh_trail_download: the batch has no trail (cbh_batch_set_trail)");
  HIPCHK(hipSetDevice(b->rep->device));
  if (b->out.ep_words) HIPCHK(hipMemcpyAsync(effective_policies, b->out.eff_pol, (size_t)b->trail_groups * b->out.ep_words * 4, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

extern "C" int cbh_check_batch_trail(cbh_table* t, const cbh_batch* in, const cbh_params* p, cbh_result* out, const uint32_t* group_of_request,
                                     uint32_t n_groups, uint32_t* effective_policies) {
  if (!t || !in || !p || !out || !effective_policies) return fail("null argument");
  cbh_device_batch* b = nullptr;
  if (cbh_batch_upload_on(t, 0, in, &b) != 0) return -1;
  struct Release { cbh_device_batch* b; ~Release() { cbh_batch_release(b); } } release{b};
  if (cbh_batch_set_trail(t, b, group_of_request, n_groups) != 0) return -1;
  cbh_params q = *p;
  q.flags |= CBH_F_WANT_EFFECTIVE_POLICIES;
  if (cbh_check_resident(t, b, &q) != 0) return -1;
  if (cbh_result_download(t, b, out) != 0) return -1;
  return cbh_trail_download(t, b, effective_policies);
}

#ifndef CBH_WIRE_LDS_DEFAULT
#define CBH_WIRE_LDS_DEFAULT 1
#endif
// ---- device-side ingest: serialized CheckInputs -> a resident batch, flattened by the GPU (cbh_wire.h) ------------------
// H2D of the raw bytes + offsets, count + scan launches, one small D2H (totals, shape), the fill launch, one small D2H
// (what it needed, what it could not take).  The batch is then an ordinary resident batch: cbh_check_resident,
// cbh_result_download - results in INPUT order (no routing sort on this path: nothing to undo).
// bytes of dynamic LDS for a wave that wants `want` bytes: a power of two between 4 and 48 KB, 0 = the kernel works in place.
// CBH_WIRE_LDS (measurement aid): 0 nothing staged, 1 the assembler's outputs only, 2 the flattener's messages too.
static int wire_lds_mode() { static const int m = [] { const char* e = getenv("CBH_WIRE_LDS"); return e ? atoi(e) : CBH_WIRE_LDS_DEFAULT; }(); return m; }
static u32 wire_lds_cap(size_t want, int needs_mode) {
  if (wire_lds_mode() < needs_mode) return 0;
  u32 c = 4096; while (c < want && c < 49152u) c <<= 1;
  return c > 49152u ? 49152u : c;
}
// The fill kernel's block of messages: staged in LDS - where the dependent loads of the parse are several times shorter than in L2 -
// when the call's LARGEST block (WireStats.max_block) leaves a CU several waves (up to CBH_WIRE_FILL_LDS_MAX bytes, in 1 KB steps);
// a call of larger messages parses them in place (the same code on a global pointer: cbh_wire_fill_kernel).
// CBH_WIRE_LDS=0/1: never; CBH_WIRE_FILL_LDS_MAX=bytes: the bound.
static u32 wire_fill_lds_cap(u32 max_block) {
  static const int mode = [] { const char* e = getenv("CBH_WIRE_LDS"); return e ? atoi(e) : 2; }();
  static const u32 most = [] { const char* e = getenv("CBH_WIRE_FILL_LDS_MAX"); return e ? (u32)atoi(e) : 32768u; }();
  if (mode < 2 || max_block == 0 || max_block > most) return 0;
  const u32 c = (max_block + 16u + CBH_WIRE_SLACK + 1023u) & ~1023u;
  return c > most ? 0u : c;
}
static bool is_pinned(const void* p);
// the replica's link streams (made on first use; CBH_WIRE_LINK_STREAMS=0: every batch copies on its own stream, as before)
static bool wire_link_streams(Replica* rep) {
  static const bool on = [] { const char* e = getenv("CBH_WIRE_LINK_STREAMS"); return !(e && *e == '0'); }();
  if (!on) return false;
  std::lock_guard<std::mutex> lk(rep->wstream_mu);
  if (!rep->link_streams_tried) {
    rep->link_streams_tried = true;
    if (hipStreamCreateWithFlags(&rep->up_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); rep->up_stream = nullptr; }
    if (rep->up_stream && hipStreamCreateWithFlags(&rep->down_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); rep->down_stream = nullptr; }
    if (!rep->down_stream && rep->up_stream) { (void)hipStreamDestroy(rep->up_stream); rep->up_stream = nullptr; }
  }
  return rep->up_stream != nullptr;
}
static hipEvent_t wire_event(cbh_device_batch* b, int which) {
  if (!b->w_ev[which]) {
    std::lock_guard<std::mutex> lk(b->rep->wstream_mu);
    if (!b->rep->wevents_idle.empty()) { b->w_ev[which] = b->rep->wevents_idle.back(); b->rep->wevents_idle.pop_back(); }
    else if (hipEventCreateWithFlags(&b->w_ev[which], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); b->w_ev[which] = nullptr; }
  }
  return b->w_ev[which];
}
// a place of the batch's page-locked block as a KERNEL addresses it: what hipHostGetDevicePointer says of the block, not the host pointer
// taken on trust
template <class T> static T* pin_dev(const cbh_device_batch* b, T* host_ptr) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(host_ptr) + b->w_pinned_delta);
}
// n_words of device memory -> the batch's page-locked block (cbh_wire_publish_kernel: no copy engine); read after a synchronise
static int wire_publish(cbh_device_batch* b, const void* d_src, void* pinned_dst, u32 n_words) {
  WirePublishArgs pa; pa.src = static_cast<const u32*>(d_src); pa.dst = pin_dev(b, static_cast<u32*>(pinned_dst)); pa.n_words = n_words; pa.pad = 0;
  hipLaunchKernelGGL(cbh_wire_publish_kernel, dim3(1), dim3(64), 0, b->stream, pa);
  HIPCHK(hipGetLastError());
  return 0;
}
static WireStats* wire_stats_land(cbh_device_batch* b) { return static_cast<WireStats*>(b->w_pinned) + 1; }   // (slot 1 of the batch's page-locked block)
static int wire_stats_read(cbh_device_batch* b, const WireStats* d_stats, WireStats& st) {
  WireStats* land = wire_stats_land(b);
  if (wire_publish(b, d_stats, land, (u32)(sizeof(st) / 4)) != 0) return -1;
  HIPCHK(hipStreamSynchronize(b->stream));
  st = *land;
  return 0;
}
static int wire_stats_write(cbh_device_batch* b, WireStats* d_stats, const WireStats& st) {
  WireStats* from = static_cast<WireStats*>(b->w_pinned);       // (slot 0)
  *from = st;
  HIPCHK(hipMemcpyAsync(d_stats, from, sizeof(st), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}

// (cbh_wire_check_pb) the uploads of a call's slices go over the link ONE AFTER THE OTHER, in slice order, so that slice k is being
// decided while slice k + 1 is still on its way: a slice's upload waits for the event its predecessor recorded behind its own
struct WireChain {
  hipEvent_t wait = nullptr, record = nullptr;
  std::atomic<int>* prev_recorded = nullptr; std::atomic<int>* recorded = nullptr;
  void done() { if (recorded) recorded->store(1, std::memory_order_release); }   // (also on every early return: the successor must not wait for ever)
};
// (cbh_wire_flatten_requests) `bytes` / `offsets` / `n` are CheckResourcesRequests: the messages the flattener works on are made on
// the device (cbh_wire_req.h)
struct WireRequests {
  const uint8_t* aux = nullptr; const uint64_t* aux_offsets = nullptr;   // serialized engine AuxData per request, or null
  uint32_t* first_input = nullptr;   // out [n + 1]: the inputs of request r are first_input[r] .. first_input[r + 1]
  uint8_t* flags = nullptr;          // out [n] (may be null): bit 0 = include_meta
};
static int wire_flatten_impl(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                             const char* default_version, const char* default_scope, const uint8_t* globals_pb, size_t globals_len,
                             cbh_device_batch** out, cbh_wire_info* info, WireChain* chain, const WireRequests* reqs = nullptr);
extern "C" int cbh_wire_flatten_requests(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n_requests,
                                         const uint8_t* aux_bytes, const uint64_t* aux_offsets, const char* default_version, const char* default_scope,
                                         const uint8_t* globals_pb, size_t globals_len, uint32_t* first_input, uint8_t* request_flags,
                                         cbh_device_batch** out, cbh_wire_info* info) {
  if (!first_input) return fail("null argument");
  if ((aux_bytes == nullptr) != (aux_offsets == nullptr)) return fail("cbh_wire_flatten_requests: aux_bytes and aux_offsets go together");
  WireRequests rq; rq.aux = aux_bytes; rq.aux_offsets = aux_offsets; rq.first_input = first_input; rq.flags = request_flags;
  return wire_flatten_impl(t, device_index, bytes, offsets, n_requests, default_version, default_scope, globals_pb, globals_len, out, info, nullptr, &rq);
}
extern "C" int cbh_wire_flatten(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                                const char* default_version, const char* default_scope, const uint8_t* globals_pb, size_t globals_len,
                                cbh_device_batch** out, cbh_wire_info* info) {
  return wire_flatten_impl(t, device_index, bytes, offsets, n, default_version, default_scope, globals_pb, globals_len, out, info, nullptr);
}
static int wire_flatten_impl(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                             const char* default_version, const char* default_scope, const uint8_t* globals_pb, size_t globals_len,
                             cbh_device_batch** out, cbh_wire_info* info, WireChain* chain, const WireRequests* reqs) {
  struct ChainGuard { WireChain* c; ~ChainGuard() { if (c) c->done(); } } chain_guard{chain};
  if (!t || !out || !info || (n && (!bytes || !offsets)) || (globals_len && !globals_pb)) return fail("null argument");
  std::memset(info, 0, sizeof(*info));
  info->first_bad = CBH_NONE; info->n_requests = n;
  if (device_index >= t->reps.size()) return fail("device index out of range");
  if (t->wire.why_not) { info->n_host = n; g_err = t->wire.why_not; return 1; }
  u64 total = n ? offsets[n] : 0;   // (requests: of the CheckInputs made of them, below)
  std::string dv = default_version ? default_version : "default", ds = default_scope ? default_scope : "";
  if (!ds.empty() && ds[0] == '.') ds.erase(0, 1);   // scope_value (namer.go:276-278)
  if (total + dv.size() + ds.size() + globals_len + 64 > 0xFFFFFFFFull) return fail("cbh_wire_flatten: more than 4 GB of messages in one call");
  Replica* rep = t->reps[device_index];
  HIPCHK(hipSetDevice(rep->device));
  cbh_device_batch* b = new (std::nothrow) cbh_device_batch();
  if (!b) return fail("out of memory");
  cbh_table_retain(t);
  b->table = t; b->rep = rep; b->wire = true;
  {
    std::lock_guard<std::mutex> lk(rep->wstream_mu);
    if (!rep->wstreams_idle.empty()) { b->stream = rep->wstreams_idle.back(); rep->wstreams_idle.pop_back(); b->own_wire_stream = true; }
    else if (rep->wstreams_made < Replica::MAX_WIRE_STREAMS && hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) == hipSuccess) { ++rep->wstreams_made; rep->wstreams_all.push_back(b->stream); b->own_wire_stream = true; }
  }
  {
    // (two WireStats, the call's tail strings, the messages' offsets - later the outputs' -, the outputs' flags, two route words)
    const size_t want = 2 * sizeof(WireStats) + dv.size() + ds.size() + 6 + globals_len + 64 + ((size_t)n + 1) * 8 + (size_t)n + 64 + 64;
    std::lock_guard<std::mutex> lk(rep->wstream_mu);
    for (size_t k = 0; k < rep->wpinned_idle.size(); ++k)
      if (rep->wpinned_idle[k].second >= want) { b->w_pinned = rep->wpinned_idle[k].first; b->w_pinned_cap = rep->wpinned_idle[k].second; rep->wpinned_idle[k] = rep->wpinned_idle.back(); rep->wpinned_idle.pop_back(); break; }
    if (!b->w_pinned) {
      size_t cap = 1 << 16; while (cap < want) cap <<= 1;
      if (hipHostMalloc(&b->w_pinned, cap, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); b->w_pinned = nullptr; }   // (kernels write it: wire_publish)
      b->w_pinned_cap = cap;
    }
  }
  if (!b->w_pinned) { cbh_batch_release(b); return fail("cbh_wire_flatten: hipHostMalloc failed"); }
  {
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, b->w_pinned, 0) != hipSuccess || !dev) { (void)hipGetLastError(); dev = b->w_pinned; }
    b->w_pinned_delta = static_cast<char*>(dev) - static_cast<char*>(b->w_pinned);
  }
  if (!b->own_wire_stream) b->stream = rep->rstreams[rep->next_rstream.fetch_add(1, std::memory_order_relaxed) % (uint32_t)rep->n_rstreams.load(std::memory_order_relaxed)];
  hipStream_t s = b->stream;
  auto bail = [&](int rc) { cbh_batch_release(b); return rc; };
  // (a sliced call, WireChain) the big upload of this slice goes behind its predecessor's
  auto chain_wait = [&](bool enqueue_order_only) -> bool {
    if (chain && chain->prev_recorded) {   // (the predecessor's thread has enqueued the record by now, or is about to)
      while (!chain->prev_recorded->load(std::memory_order_acquire)) std::this_thread::yield();
      if (enqueue_order_only) return true;   // (one upload stream: its order is the order of the calls)
      // The successor's upload is handed to the copy engines only when the predecessor's has LANDED (a wait on the host, not a
      // dependency on the device): uploads queued ahead of time are spread over the engines, and a slice's answers - a copy the
      // other way, asked for later - then wait behind them all (measured: downloads began when the last upload had ended, although
      // the link carries both directions at once: tools/pcie_duplex.hip).  CBH_WIRE_CHAIN_DEVICE=1: the dependency on the device.
      static const bool on_device = getenv("CBH_WIRE_CHAIN_DEVICE") != nullptr;
      if (chain->wait && (on_device ? hipStreamWaitEvent(s, chain->wait, 0) : hipEventSynchronize(chain->wait)) != hipSuccess) { fail("cbh_wire_flatten: waiting for the previous slice's upload failed"); return false; }
    }
    return true;
  };
  auto chain_record = [&]() { if (chain && chain->record) { (void)hipEventRecord(chain->record, s); chain->done(); } };
  u8* d_msg = nullptr; u64* d_moff = nullptr;
  const u32 n_in = n;
  const size_t tail_room = dv.size() + ds.size() + globals_len + 64;
  if (reqs) {
    // ---- CheckResourcesRequests -> the CheckInputs of their resource entries, on the device (cbh_wire_req.h): counts, two prefix
    // sums on the host, the split.  From here on `n` / `total` are the inputs' and their bytes'.
    const u32 nr = n_in;
    const u64 rtotal = total, atotal = (reqs->aux_offsets && nr) ? reqs->aux_offsets[nr] : 0;
    if (atotal > 0xFFFFFFFFull) return bail(fail("cbh_wire_flatten_requests: more than 4 GB of auxiliary data in one call"));
    WireReqArgs q; std::memset(&q, 0, sizeof(q));
    u8* d_req = nullptr; u64* d_roff = nullptr; u8* d_aux = nullptr; u64* d_aoff = nullptr; u32* d_first = nullptr; u64* d_fbyte = nullptr;
    int rq = 0;
    rq |= dalloc(b, d_req, (size_t)rtotal + 8); rq |= dalloc(b, d_roff, (size_t)nr + 1);
    rq |= dalloc(b, q.n_inputs, (size_t)nr + 1); rq |= dalloc(b, q.n_bytes, (size_t)nr + 1); rq |= dalloc(b, q.flags, (size_t)nr + 1);
    rq |= dalloc(b, d_first, (size_t)nr + 1); rq |= dalloc(b, d_fbyte, (size_t)nr + 1);
    if (reqs->aux_offsets) { rq |= dalloc(b, d_aux, (size_t)atotal + 8); rq |= dalloc(b, d_aoff, (size_t)nr + 1); }
    if (rq != 0) return bail(-1);
    hipEvent_t ev_rq = wire_link_streams(rep) ? wire_event(b, 0) : nullptr;   // (the replica's upload stream, as for CheckInputs below)
    hipStream_t rs = ev_rq ? rep->up_stream : s;
    if (!chain_wait(ev_rq != nullptr)) return bail(-1);
    if ((rtotal && hipMemcpyAsync(d_req, bytes, rtotal, hipMemcpyHostToDevice, rs) != hipSuccess) ||
        (nr && hipMemcpyAsync(d_roff, offsets, ((size_t)nr + 1) * 8, hipMemcpyHostToDevice, rs) != hipSuccess) ||
        (atotal && hipMemcpyAsync(d_aux, reqs->aux, atotal, hipMemcpyHostToDevice, rs) != hipSuccess) ||
        (reqs->aux_offsets && nr && hipMemcpyAsync(d_aoff, reqs->aux_offsets, ((size_t)nr + 1) * 8, hipMemcpyHostToDevice, rs) != hipSuccess))
      { fail("cbh_wire_flatten_requests: upload failed"); return bail(-1); }
    if (ev_rq) {
      const bool ok = hipEventRecord(ev_rq, rs) == hipSuccess;
      if (chain) chain->done();
      if (!ok || hipStreamWaitEvent(s, ev_rq, 0) != hipSuccess) { fail("cbh_wire_flatten_requests: upload failed"); return bail(-1); }
    } else chain_record();
    q.req = d_req; q.roff = d_roff; q.n = nr; q.end = (u32)rtotal; q.aux = d_aux; q.aoff = reqs->aux_offsets ? d_aoff : nullptr; q.aux_end = atotal;
    if (nr) hipLaunchKernelGGL(cbh_wire_req_count_kernel, dim3((nr + CBH_BLOCK - 1) / CBH_BLOCK), dim3(CBH_BLOCK), 0, s, q);
    std::vector<u32> h_inputs((size_t)nr + 1, 0), h_first((size_t)nr + 1, 0); std::vector<u64> h_bytes((size_t)nr + 1, 0), h_fbyte((size_t)nr + 1, 0);
    if (nr && (hipMemcpyAsync(h_inputs.data(), q.n_inputs, (size_t)nr * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
               hipMemcpyAsync(h_bytes.data(), q.n_bytes, (size_t)nr * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
               (reqs->flags && hipMemcpyAsync(reqs->flags, q.flags, (size_t)nr, hipMemcpyDeviceToHost, s) != hipSuccess)))
      { fail("cbh_wire_flatten_requests: download failed"); return bail(-1); }
    if (hipStreamSynchronize(s) != hipSuccess) { fail("cbh_wire_flatten_requests failed"); return bail(-1); }
    u64 n_inputs = 0, n_bytes = 0;
    for (u32 r = 0; r < nr; ++r) {
      if (h_inputs[r] == CBH_WREQ_BAD) { info->first_bad = r; fail("malformed CheckResourcesRequest at index " + std::to_string(r)); return bail(-1); }
      h_first[r] = (u32)n_inputs; h_fbyte[r] = n_bytes;
      n_inputs += h_inputs[r]; n_bytes += h_bytes[r];
      if (n_inputs > 0x7FFFFFFFull) { fail("cbh_wire_flatten_requests: too many resource entries in one call"); return bail(-1); }
    }
    h_first[nr] = (u32)n_inputs; h_fbyte[nr] = n_bytes;
    if (n_bytes + tail_room > 0xFFFFFFFFull) { fail("cbh_wire_flatten_requests: more than 4 GB of CheckInputs in one call"); return bail(-1); }
    std::memcpy(reqs->first_input, h_first.data(), ((size_t)nr + 1) * 4);
    n = (u32)n_inputs; total = n_bytes;
    info->n_requests = n;
    rq = 0;
    rq |= dalloc(b, d_msg, (size_t)total + tail_room); rq |= dalloc(b, d_moff, (size_t)n + 1);
    if (rq != 0) return bail(-1);
    q.first_input = d_first; q.first_byte = d_fbyte; q.msg = d_msg; q.moff = d_moff;
    // (pageable sources: the copies are staged before the call returns to this thread, the vectors outlive them)
    if (hipMemcpyAsync(d_first, h_first.data(), ((size_t)nr + 1) * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_fbyte, h_fbyte.data(), ((size_t)nr + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemsetAsync(d_moff, 0, 8, s) != hipSuccess)
      { fail("cbh_wire_flatten_requests: upload failed"); return bail(-1); }
    if (nr) hipLaunchKernelGGL(cbh_wire_req_split_kernel, dim3((nr + (CBH_BLOCK / 64u) - 1) / (CBH_BLOCK / 64u)), dim3(CBH_BLOCK), 0, s, q);
    if (hipStreamSynchronize(s) != hipSuccess) { fail("cbh_wire_flatten_requests failed"); return bail(-1); }   // (h_first / h_fbyte go out of scope)
  }
  // a malformed message: by index of the CheckInput, or - requests - of the request its resource entry belongs to
  auto bad_input = [&](u32 i) {
    if (!reqs) { info->first_bad = i; fail("malformed CheckInput at index " + std::to_string(i)); return; }
    u32 r = 0;
    while (r + 1u < n_in && reqs->first_input[r + 1u] <= i) ++r;
    info->first_bad = r;
    fail("malformed CheckResourcesRequest at index " + std::to_string(r) + " (resource entry " + std::to_string(i - reqs->first_input[r]) + ")");
  };
  const u32 nw = (n + 63u) / 64u, ncol = t->meta[CBH_M_NCOLUMNS];
  WireArgs a; std::memset(&a, 0, sizeof(a));
  const TableDev& td = rep->dev;
  a.t_str_off = td.str_off; a.t_str_bytes = td.str_bytes; a.K = td.K; a.t_flags = td.flags;
  a.tix = rep->w_tix; a.tix_mask = t->wire.tix_mask; a.scope_of_sid = rep->w_scope_of_sid;
  a.cols = rep->w_cols; a.col_keys = rep->w_col_keys; a.n_cols = ncol; a.sens_cols = t->meta[CBH_M_SENS_COLS];
  a.n = n;
  a.dver_off = (u32)total; a.dver_len = (u32)dv.size(); a.dscope_off = (u32)(total + dv.size()); a.dscope_len = (u32)ds.size();
  a.claims_off = (u32)(total + dv.size() + ds.size());
  a.globals_off = a.claims_off + 6u; a.globals_len = (u32)globals_len;
  WireStats* d_stats = nullptr;
  int rc = 0;
  if (!reqs) { rc |= dalloc(b, d_msg, (size_t)total + tail_room); rc |= dalloc(b, d_moff, (size_t)n + 1); }
  rc |= dalloc(b, a.cnt, (size_t)n + 1); rc |= dalloc(b, a.status, (size_t)n + 1);
  rc |= dalloc(b, a.wavesum, 4 * (size_t)nw + 4); rc |= dalloc(b, a.waveoff, 2 * (size_t)nw + 4);
  rc |= dalloc(b, d_stats, 1);
  if (rc != 0) return bail(-1);
  a.msg = d_msg; a.moff = d_moff; a.stats = d_stats;
  WireStats st; cbh_wire_stats_init(st);
  std::string tail = dv + ds + "claims";
  if (globals_len) tail.append(reinterpret_cast<const char*>(globals_pb), globals_len);
  // everything small goes through the batch's page-locked block: statistics (slots 0 / 1), the tail, the offsets
  u8* pin = static_cast<u8*>(b->w_pinned);
  WireStats* pin_st = reinterpret_cast<WireStats*>(pin);
  u8* pin_tail = pin + 2 * sizeof(WireStats);
  u64* pin_off = reinterpret_cast<u64*>(pin + ((2 * sizeof(WireStats) + tail.size() + 63) & ~(size_t)63));
  *pin_st = st;
  std::memcpy(pin_tail, tail.data(), tail.size());
  if (!reqs) { if (n) std::memcpy(pin_off, offsets, ((size_t)n + 1) * 8); else pin_off[0] = 0; }
  b->w_pin_out_at = (size_t)(reinterpret_cast<u8*>(pin_off) - pin);   // (cbh_wire_outputs: the outputs' offsets and flags land here, written by the kernels)
  // the uploads: on the replica's upload stream (one copy engine for this direction, the slices of a call in their order - the
  // chain only orders the ENQUEUEING then), the batch's own stream takes over behind an event; else on the batch's stream
  hipEvent_t ev_up = (!reqs && wire_link_streams(rep)) ? wire_event(b, 0) : nullptr;
  hipStream_t us = ev_up ? rep->up_stream : s;
  if (!reqs) {
    if (!chain_wait(ev_up != nullptr)) return bail(-1);
    if (total && hipMemcpyAsync(d_msg, bytes, total, hipMemcpyHostToDevice, us) != hipSuccess) { fail("cbh_wire_flatten: upload failed"); return bail(-1); }
    if (!ev_up) chain_record();
  }
  // (the small ones on the batch's own stream: on the upload stream every one of them would be a gap between two slices' messages)
  if (hipMemcpyAsync(d_msg + total, pin_tail, tail.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
      (!reqs && hipMemcpyAsync(d_moff, pin_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess) ||
      hipMemcpyAsync(d_stats, pin_st, sizeof(st), hipMemcpyHostToDevice, s) != hipSuccess) { fail("cbh_wire_flatten: upload failed"); return bail(-1); }
  if (ev_up) {
    const bool ok = hipEventRecord(ev_up, us) == hipSuccess;
    if (chain) chain->done();   // (the successor may enqueue its uploads now)
    if (!ok || hipStreamWaitEvent(s, ev_up, 0) != hipSuccess) { fail("cbh_wire_flatten: upload failed"); return bail(-1); }
  }
  {   // the count kernel's staging block: the call's average block and a half (the fill is sized by the largest, which the count finds)
    const u64 avg_block = n ? total / n * 64u : 0u;
    a.lds_cap = wire_fill_lds_cap((u32)std::min<u64>(avg_block + avg_block / 2u + 256u, 0xFFFFFFFFull));
  }
  if (nw) hipLaunchKernelGGL(cbh_wire_count_kernel, dim3(nw), dim3(CBH_BLOCK), a.lds_cap, s, a);
  wmark("uploads+count enqueued");
  // Group the requests by route (cbh_wire.h cbh_wire_route_kernel ...): what the host flattener's routing sort does for the
  // decision kernels' merged walk - launched behind every fill (launch_routes).  CBH_WIRE_GROUP=0: leave the batch in input order
  // (measurement aid).
  static const bool group_on = [] { const char* e = getenv("CBH_WIRE_GROUP"); return !(e && *e == '0'); }();
  WireRouteArgs ra; std::memset(&ra, 0, sizeof(ra));
  u32* pin_routes = reinterpret_cast<u32*>(static_cast<u8*>(b->w_pinned) + b->w_pinned_cap - 64);   // (the block's last words: routes in use, overflow flag)
  const bool try_group = group_on && n >= 2u * CBH_BLOCK;
  auto launch_routes = [&]() -> int {
    if (!try_group) return 0;
    if (!ra.rt_key) {
      ra.n = n; ra.n_cols = ncol; ra.req_u32 = a.req_u32; ra.roles = a.roles; ra.col_tag = a.col_tag; ra.col_val = a.col_val;
      ra.multi = &d_stats->multi_route;
      int rr = 0;
      rr |= dalloc(b, ra.rt_key, (size_t)CBH_WIRE_ROUTE_SLOTS + ((size_t)CBH_WIRE_ROUTE_SLOTS + 2 + 1) / 2);   // (the keys and, behind them, the counters: one memset)
      ra.rt_cnt = reinterpret_cast<u32*>(ra.rt_key + CBH_WIRE_ROUTE_SLOTS);
      ra.host_routes = pin_dev(b, pin_routes); ra.stats = d_stats; ra.host_stats = pin_dev(b, wire_stats_land(b));
      rr |= dalloc(b, ra.slot, (size_t)n); rr |= dalloc(b, ra.rank, (size_t)n); rr |= dalloc(b, ra.inv, (size_t)n);
      rr |= dalloc(b, ra.req_out, (size_t)CBH_RQ_NFIELDS * n); rr |= dalloc(b, ra.col_tag_out, (size_t)ncol * n); rr |= dalloc(b, ra.col_val_out, (size_t)ncol * n);
      if (rr != 0) return -1;
    }
    if (hipMemsetAsync(ra.rt_key, 0, ((size_t)CBH_WIRE_ROUTE_SLOTS + ((size_t)CBH_WIRE_ROUTE_SLOTS + 2 + 1) / 2) * 8, s) != hipSuccess) return fail("cbh_wire_flatten: memset failed");
    hipLaunchKernelGGL(cbh_wire_route_kernel, dim3(nw), dim3(CBH_BLOCK), 0, s, ra);
    hipLaunchKernelGGL(cbh_wire_route_scan_kernel, dim3(1), dim3(CBH_BLOCK), 0, s, ra);   // (leaves the route words AND the fill's statistics with the host)
    hipLaunchKernelGGL(cbh_wire_gather_kernel, dim3(nw), dim3(CBH_BLOCK), 0, s, ra);
    HIPCHK(hipGetLastError());
    return 0;
  };
  u32 slots = cbh_wire_dict_slots(n), heap_cap = cbh_wire_heap_guess(total);
  u32 n_host_count = 0; bool have_outputs = false; u32 runs = 0;
  for (;;) {
    // (re)start from the scan: the dictionary is empty, the scan interns the call's default strings first
    rc = 0;
    {   // the dictionary's words and, behind them, its flag bytes: one block, one memset
      u64* dict = nullptr;
      rc |= dalloc(b, dict, (size_t)slots + ((size_t)slots / 4 + 1 + 1) / 2);
      if (rc != 0) return bail(-1);
      a.lix = dict; a.lflags = reinterpret_cast<u32*>(dict + slots);
      a.lix_mask = slots - 1;
      if (hipMemsetAsync(dict, 0, ((size_t)slots + ((size_t)slots / 4 + 1 + 1) / 2) * 8, s) != hipSuccess) { fail("cbh_wire_flatten: memset failed"); return bail(-1); }
    }
    a.host_stats = pin_dev(b, wire_stats_land(b));   // (the scan kernel leaves the statistics there itself)
    hipLaunchKernelGGL(cbh_wire_scan_kernel, dim3(1), dim3(CBH_BLOCK), 0, s, a);
    if (hipStreamSynchronize(s) != hipSuccess) { fail("cbh_wire_flatten failed"); return bail(-1); }
    st = *wire_stats_land(b);
    wmark("counts known");
    if (!have_outputs) {
      n_host_count = st.n_host;
      if (st.first_bad != CBH_NONE) { bad_input(st.first_bad); return bail(-1); }
      if (st.n_host) {   // the count already found messages for the host flattener (more than 64 actions / 255 roles): no point in filling
        info->n_tuples = st.n_tuples; info->n_host = st.n_host;
        g_err = "cbh_wire_flatten: " + std::to_string(st.n_host) + " message(s) are the host flattener's (more than 64 actions or 255 roles)";
        return bail(1);
      }
      rc = 0;
      rc |= dalloc(b, a.req_u32, (size_t)CBH_RQ_NFIELDS * n); rc |= dalloc(b, a.roles, (size_t)st.n_roles); rc |= dalloc(b, a.tuple_action, (size_t)st.n_tuples);
      rc |= dalloc(b, a.col_tag, (size_t)ncol * n); rc |= dalloc(b, a.col_val, (size_t)ncol * n);
      rc |= dalloc(b, a.in_span, (size_t)n * 2 * CBH_WSPAN_N); rc |= dalloc(b, a.act_span, (size_t)st.n_tuples * 2);
      if (rc != 0) return bail(-1);
      have_outputs = true;
    }
    bool again = false;
    for (;;) {   // the fill, once more with the heap it asked for if the guess was short
      rc = 0;
      rc |= dalloc(b, a.heap_tag, (size_t)heap_cap); rc |= dalloc(b, a.heap_val, (size_t)heap_cap);
      if (rc != 0) return bail(-1);
      a.heap_cap = heap_cap;
      // dynamic LDS: room for a wave's 64 messages (a quarter above the call's average; a wave whose block is larger parses in place)
      a.lds_cap = wire_fill_lds_cap(st.max_block);
      if (nw && a.lds_cap) hipLaunchKernelGGL(cbh_wire_fill_lds_kernel, dim3(nw), dim3(CBH_BLOCK), cbh_wire_fill_cur_bytes(ncol) + a.lds_cap, s, a);
      else if (nw) hipLaunchKernelGGL(cbh_wire_fill_kernel, dim3(nw), dim3(CBH_BLOCK), cbh_wire_fill_cur_bytes(ncol), s, a);
      ++runs;
      // what the fill wanted and the routing of what it wrote (for nothing, the rare time the fill is run again) - ONE wait for both
      if (!try_group && wire_publish(b, d_stats, wire_stats_land(b), (u32)(sizeof(st) / 4)) != 0) return bail(-1);
      if (launch_routes() != 0) return bail(-1);
      wmark("fill+routes enqueued");
      if (hipStreamSynchronize(s) != hipSuccess) { fail("cbh_wire_flatten failed"); return bail(-1); }
      wmark("filled");
      st = *wire_stats_land(b);
      if (st.flags & CBH_WF_DICT_FULL) { again = true; break; }
      if (st.heap_used <= heap_cap) break;
      heap_cap = st.heap_used;
      WireStats reset = st; reset.heap_used = 0; reset.n_host = n_host_count; reset.flags = 0; reset.route_lo = reset.route_hi = reset.multi_route = 0;
      if (wire_stats_write(b, d_stats, reset) != 0) return bail(-1);
    }
    if (!again) break;
    if (slots >= (1u << 30)) { fail("cbh_wire_flatten: the batch-local dictionary cannot grow further"); return bail(-1); }
    slots *= 4;
    WireStats reset = st; reset.heap_used = 0; reset.n_host = n_host_count; reset.flags = 0; reset.route_lo = reset.route_hi = reset.multi_route = 0;
    if (wire_stats_write(b, d_stats, reset) != 0) return bail(-1);
  }
  { const hipError_t le = hipGetLastError(); if (le != hipSuccess) { fail(std::string("cbh_wire_flatten: ") + hipGetErrorString(le)); return bail(-1); } }
  if (st.heap_used >= (1u << 30)) { fail("cbh_wire_flatten: batch too large: nested attribute values exceed the heap's 30-bit offsets"); return bail(-1); }   // (as cbi_flatten_pb)
  info->n_tuples = st.n_tuples; info->n_host = st.n_host; info->dict_slots = slots; info->heap_len = st.heap_used; info->fill_runs = runs;
  if (st.first_bad != CBH_NONE) { bad_input(st.first_bad); return bail(-1); }
  if (st.n_host) { g_err = "cbh_wire_flatten: " + std::to_string(st.n_host) + " message(s) are the host flattener's (more than 64 actions, a resource kind to rewrite that no policy names, containers nested too deep)"; return bail(1); }
  BatchDev& d = b->dev;
  d.n_requests = n; d.n_tuples = st.n_tuples; d.n_roles = st.n_roles; d.n_columns = ncol; d.n_strings = slots; d.heap_len = st.heap_used;
  d.req_lo = 0; d.req_hi = n;
  d.req_u32 = a.req_u32; d.roles = a.roles; d.tuple_req = nullptr; d.tuple_action = a.tuple_action; d.col_tag = a.col_tag; d.col_val = a.col_val;
  b->w_req_input = a.req_u32;
  d.heap_tag = a.heap_tag; d.heap_val = a.heap_val; d.str_off = nullptr; d.str_bytes = d_msg; d.str_flags = (const u8*)a.lflags; d.str_keys = a.lix;
  b->w_in_span = a.in_span; b->w_act_span = a.act_span; b->w_moff = d_moff; b->w_dver_off = a.dver_off; b->w_dver_len = a.dver_len;
  static const bool force_any = getenv("CBH_FLAT_ANY") != nullptr;
  b->max_actions = st.max_actions; b->max_roles = st.max_roles; b->plain_tags = !force_any && !(st.flags & CBH_WF_CONTAINER_IN_SENS);
  b->wide_lo = st.wide_hi ? st.wide_lo : 0; b->wide_hi = st.wide_hi;
  rc = 0;
  const bool globs = nfa_maxw(rep->dev) != 0;
  rc |= dalloc(b, d.gbits, globs ? (size_t)3 * slots : (size_t)1);
  d.n_gwords = (rep->dev.flags & CBH_MF_WALK2) ? w2_gwords(rep->dev.gslots_generic, rep->dev.gslots_all, b->plain_tags) : 0;
  d.n_gslots = 0;
  if (d.n_gwords) rc |= dalloc(b, d.gres, (size_t)d.n_gwords * n); else d.gres = nullptr;
  rc |= dalloc(b, b->out.effect, (size_t)st.n_tuples); rc |= dalloc(b, b->out.policy, (size_t)st.n_tuples);
  rc |= dalloc(b, b->out.scope, (size_t)st.n_tuples); rc |= dalloc(b, b->out.status, (size_t)st.n_tuples);
  rc |= dalloc(b, b->out.edr, (size_t)n); rc |= dalloc(b, b->d_args, 1);
  if (rc != 0) return bail(-1);
  if (globs && hipMemsetAsync(d.gbits, 0, (size_t)3 * slots * sizeof(u64), s) != hipSuccess) { fail("cbh_wire_flatten: memset failed"); return bail(-1); }
  { const hipError_t le = hipGetLastError(); if (le != hipSuccess) { fail(std::string("cbh_wire_flatten: ") + hipGetErrorString(le)); return bail(-1); } }
  if (try_group && pin_routes[1] == 0u && pin_routes[0] > 1u) {   // grouped (not: a full route table, or one route - nothing to group)
    d.req_u32 = ra.req_out; d.col_tag = ra.col_tag_out; d.col_val = ra.col_val_out;
    b->w_inv = ra.inv;
    if (b->wide_hi) { b->wide_lo = 0; b->wide_hi = n; }   // the wider requests lie anywhere now: their launch skips the others lane by lane
    info->n_routes = pin_routes[0];
  }
  *out = b;
  return 0;
}

// Where the strings a CheckOutput repeats sit in each message (what cbi_assemble_wire_pb reads instead of walking the messages
// again): in_span [n][6] (offset, length) pairs relative to the message - request id, principal id / version, resource kind /
// version / id; act_span [n_tuples] (offset, length) of each action; act_off [n + 1] first tuple of each input.
extern "C" int cbh_wire_spans_download(cbh_table* t, cbh_device_batch* b, uint32_t* in_span, uint32_t* act_span, uint32_t* act_off) {
  if (!t || !b || !in_span || !act_span || !act_off) return fail("null argument");
  if (!b->wire) return fail("cbh_wire_spans_download: not a batch of cbh_wire_flatten");
  Replica* rep = b->rep;
  HIPCHK(hipSetDevice(rep->device));
  const BatchDev& d = b->dev;
  const size_t n = d.n_requests;
  if (n) HIPCHK(hipMemcpyAsync(in_span, b->w_in_span, n * 2 * CBH_WSPAN_N * 4, hipMemcpyDeviceToHost, b->stream));
  if (d.n_tuples) HIPCHK(hipMemcpyAsync(act_span, b->w_act_span, (size_t)d.n_tuples * 2 * 4, hipMemcpyDeviceToHost, b->stream));
  if (n) HIPCHK(hipMemcpyAsync(act_off, b->w_req_input + (size_t)CBH_RQ_ACT_OFF * n, n * 4, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  act_off[n] = d.n_tuples;
  return 0;
}


// The serialized CheckOutputs of a batch the device flattened, written by the device (cbh_wire.h cbh_wire_out_*): after
// cbh_check_resident on `b`, three launches on its stream - sizes, scan, bytes - and one copy back.
extern "C" int cbh_wire_outputs(cbh_table* t, cbh_device_batch* b, uint8_t* bytes, size_t cap, uint64_t* offsets, uint8_t* flags, size_t* need) {
  if (!t || !b || !offsets || !need || (cap && !bytes)) return fail("null argument");
  if (!b->wire) return fail("cbh_wire_outputs: not a batch of cbh_wire_flatten");
  Replica* rep = b->rep;
  HIPCHK(hipSetDevice(rep->device));
  hipStream_t s = b->stream;
  const BatchDev& d = b->dev;
  const u32 n = d.n_requests, nw = (n + 63u) / 64u;
  *need = 0;
  const bool fresh_ostats = !b->w_sizes;
  if (!b->w_sizes) {
    int rc = 0;
    rc |= dalloc(b, b->w_sizes, (size_t)n + 1); rc |= dalloc(b, b->w_wavesum, (size_t)nw + 1); rc |= dalloc(b, b->w_waveoff, (size_t)nw + 1);
    rc |= dalloc(b, b->w_ostats, 1); rc |= dalloc(b, b->w_out_off, (size_t)n + 1); rc |= dalloc(b, b->w_out_flags, (size_t)n + 1);
    if (rc != 0) return -1;
  }
  WireOutArgs a; std::memset(&a, 0, sizeof(a));
  const TableDev& td = rep->dev;
  a.t_str_off = td.str_off; a.t_str_bytes = td.str_bytes;
  a.scope_sid = reinterpret_cast<const u32*>(static_cast<const uint8_t*>(rep->image) + t->wire.scope_sid_offset); a.n_scopes = t->wire.n_scopes;
  a.n_policies = t->wire.n_policies; a.name_off = rep->w_name_off; a.name_bytes = rep->w_name_bytes; a.n_dr = t->wire.n_dr; a.n = n;
  a.msg = d.str_bytes; a.moff = b->w_moff; a.dver_off = b->w_dver_off; a.dver_len = b->w_dver_len;
  a.req_u32 = b->w_req_input; a.tuple_action = d.tuple_action; a.in_span = b->w_in_span; a.act_span = b->w_act_span; a.inv = b->w_inv;
  a.effect = b->out.effect; a.policy = b->out.policy; a.scope = b->out.scope; a.status = b->out.status; a.edr = b->out.edr;
  a.sizes = b->w_sizes; a.wavesum = b->w_wavesum; a.waveoff = b->w_waveoff; a.stats = b->w_ostats; a.out_off = b->w_out_off; a.out_flags = b->w_out_flags;
  WireOutStats st; std::memset(&st, 0, sizeof(st));
  static_assert(sizeof(WireOutStats) <= sizeof(WireStats), "the batch's page-locked block has two WireStats slots");
  // The outputs' offsets and flags are written by the kernels straight into the batch's page-locked block (where the messages'
  // offsets went up from: long since on the device) when it has the room - two copies less on the link per call, and none that
  // waits behind another slice's bulk copy; the caller's arrays are filled from there.
  u8* pin = static_cast<u8*>(b->w_pinned);
  const size_t off_bytes = ((size_t)n + 1) * 8, flags_at = b->w_pin_out_at + ((off_bytes + 63) & ~(size_t)63);
  const bool direct = pin && b->w_pin_out_at && flags_at + (size_t)n + 64 + 64 <= b->w_pinned_cap;
  u64* pin_off = direct ? reinterpret_cast<u64*>(pin + b->w_pin_out_at) : nullptr;
  u8* pin_flags = direct ? pin + flags_at : nullptr;
  if (direct) { a.out_off = pin_dev(b, pin_off); a.out_flags = pin_dev(b, pin_flags); }
  if (b->w_total_known) { st.total = b->w_total; st.errors = b->w_out_errors; }   // sizes and offsets of these results are on the device already
  else {
    WireOutStats* pin_st = static_cast<WireOutStats*>(b->w_pinned);   // (page-locked slot 0; the scan kernel writes it and clears the error bits behind itself)
    if (fresh_ostats) HIPCHK(hipMemsetAsync(b->w_ostats, 0, sizeof(st), s));
    a.host_stats = pin_dev(b, pin_st);
    if (nw) hipLaunchKernelGGL(cbh_wire_out_size_kernel, dim3(nw), dim3(CBH_BLOCK), 0, s, a);
    hipLaunchKernelGGL(cbh_wire_out_scan_kernel, dim3(1), dim3(CBH_BLOCK), 0, s, a);
    HIPCHK(hipStreamSynchronize(s));
    st = *pin_st;
    b->w_total_known = true; b->w_total = st.total; b->w_out_errors = st.errors;
  }
  if (st.errors & 1u) return fail("cbh_wire_outputs: a policy or scope id of the results is out of the table's range");
  if (st.errors & 2u) return fail("cbh_wire_outputs: a CheckOutput exceeds 16 MB");
  *need = (size_t)st.total;
  if (st.total > cap) { g_err = "cbh_wire_outputs: the output buffer is too small"; return 2; }
  // Where the bytes are written: into device memory and one copy back - or, with CBH_WIRE_OUT_DIRECT=1, into the CALLER's buffer when
  // that is page-locked memory the device can reach (the kernel's 16-byte stores cross the link themselves).  Measured on the sliced
  // road (C2, 250 000 messages per call): the direct stores run at 43 GB/s and slow the copy engine's uploads and the other slices'
  // kernels beside them - 2.17 ms a call against 1.94 ms with the copy (tools/pcie_duplex.hip: engine upload + kernel download
  // 1.16 ms, both by the engines 0.92 ms) - so the copy is the default.
  static const bool direct_on = [] { const char* e = getenv("CBH_WIRE_OUT_DIRECT"); return e && *e == '1'; }();
  u8* d_out = nullptr; u8* host_out = nullptr;
  if (direct_on && st.total && is_pinned(bytes) && hipHostGetDevicePointer((void**)&host_out, bytes, 0) != hipSuccess) { (void)hipGetLastError(); host_out = nullptr; }
  if (host_out) {
    a.out_bias = (u32)(reinterpret_cast<uintptr_t>(host_out) & 15u);
    a.out = host_out - a.out_bias;
  } else {
    if (dalloc(b, d_out, (size_t)st.total + 1) != 0) return -1;
    a.out = d_out; a.out_bias = 0;
  }
  a.lds_cap = wire_lds_cap(n ? (size_t)(st.total / n) * 80u + 256u : 0u, 1);
  if (nw) hipLaunchKernelGGL(cbh_wire_out_write_kernel, dim3(nw), dim3(CBH_BLOCK), a.lds_cap, s, a);
  // the bytes' way back: on the replica's download stream (the copy engine of that direction), behind an event of the kernel
  hipEvent_t ev_w = (st.total && d_out && wire_link_streams(rep)) ? wire_event(b, 0) : nullptr, ev_d = ev_w ? wire_event(b, 1) : nullptr;
  if (ev_w && ev_d && hipEventRecord(ev_w, s) == hipSuccess && hipStreamWaitEvent(rep->down_stream, ev_w, 0) == hipSuccess) {
    HIPCHK(hipMemcpyAsync(bytes, d_out, (size_t)st.total, hipMemcpyDeviceToHost, rep->down_stream));
    if (hipEventRecord(ev_d, rep->down_stream) != hipSuccess) {   // the copy flies with nothing to wait on but its stream
      (void)hipGetLastError(); (void)hipStreamSynchronize(rep->down_stream);
      return fail("cbh_wire_outputs: hipEventRecord failed behind the download");
    }
  } else {
    ev_d = nullptr;
    if (st.total && d_out) HIPCHK(hipMemcpyAsync(bytes, d_out, (size_t)st.total, hipMemcpyDeviceToHost, s));
  }
  if (!direct) {
    HIPCHK(hipMemcpyAsync(offsets, b->w_out_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, s));
    if (flags && n) HIPCHK(hipMemcpyAsync(flags, b->w_out_flags, (size_t)n, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  if (ev_d) HIPCHK(hipEventSynchronize(ev_d));
  HIPCHK(hipGetLastError());
  if (direct) {
    std::memcpy(offsets, pin_off, off_bytes);
    if (flags && n) std::memcpy(flags, pin_flags, (size_t)n);
  }
  if (d_out) {   // the output block goes back to the pool now: a batch that is asked again allocates again
    std::lock_guard<std::mutex> lk(rep->pool_mu);
    for (size_t i = b->allocs.size(); i-- > 0;) if (b->allocs[i].first == d_out) { rep->pool_free.push_back(b->allocs[i]); b->allocs.erase(b->allocs.begin() + (long)i); break; }
  }
  return 0;
}

// Bytes in, bytes out in ONE call: serialized CheckInputs -> serialized CheckOutputs by the device road (cbh_wire_flatten,
// cbh_check_resident, cbh_wire_outputs), the call cut into up to four slices of contiguous messages that go down the road side
// by side, each on a thread and a stream of its own - one slice's copies run under another's kernels, which a single caller
// thread making the three calls in a row never gets (its H2D, kernels and D2H queue behind each other).  The slices' outputs
// land back to back in `out_bytes`: every slice first learns its size (the size / scan launches), the bases follow, then each
// writes and copies into its own range.  Returns 0; 1 = some message is the host flattener's (info->n_host; nothing was
// written); 2 = `out_cap` is too small, *need holds the size; < 0 error.
//
// The same for what the SERVER receives (`rm`): the units are serialized CheckResourcesRequests, a slice is a range of requests, its
// CheckInputs are made on the device (cbh_wire_req.h); the outputs of request r are out_offsets[first_input[r]] ..
// out_offsets[first_input[r + 1]]; with `rm->effective_policies` every request also gets its audit trail (one group per request:
// the one decision-log entry svc.CheckResources writes for the call).  2 also when out_offsets / out_flags hold fewer inputs than the
// requests have (info->n_requests = the inputs).
struct WireReqMode {
  const uint8_t* aux = nullptr; const uint64_t* aux_offsets = nullptr;
  uint32_t* first_input = nullptr; uint8_t* request_flags = nullptr; size_t out_inputs_cap = 0;
  uint32_t* effective_policies = nullptr;
};
static int wire_check_sliced(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                             const char* default_version, const char* default_scope, const uint8_t* globals_pb, size_t globals_len,
                             const cbh_params* p, uint8_t* out_bytes, size_t out_cap, uint64_t* out_offsets, uint8_t* out_flags,
                             size_t* need, cbh_wire_info* info, const WireReqMode* rm) {
  TableRef ref(t);
  std::memset(info, 0, sizeof(*info));
  info->first_bad = CBH_NONE; info->n_requests = rm ? 0u : n;
  *need = 0;
  static const u32 max_slices = [] { const char* e = getenv("CBH_WIRE_SLICES"); const long v = e ? atol(e) : 4; return (u32)std::min<long>(std::max<long>(v, 1), 8); }();
  // (requests: by their bytes - a request holds any number of resource entries -, about 4 MB to a slice.  CBH_WIRE_SLICE_MIN /
  // CBH_WIRE_SLICE_MIN_BYTES: the smallest slice, for tests and measurements)
  static const u32 slice_min = [] { const char* e = getenv("CBH_WIRE_SLICE_MIN"); const long v = e ? atol(e) : 16384; return (u32)std::max<long>(v, 1); }();
  static const u64 slice_min_bytes = [] { const char* e = getenv("CBH_WIRE_SLICE_MIN_BYTES"); const long long v = e ? atoll(e) : (1ll << 22); return (u64)std::max<long long>(v, 1); }();
  // Calls of the road that are in flight at once (several caller threads, or one with cbh_wire_check_pb_submit) share the link and
  // the copy engines: more than about four slices side by side lose (a lone call cut into eight: 392 M against 489 M decisions/s,
  // profiles/r04_wire_onecall.txt; two calls of four: 420 M against 545 M one at a time, round 6) - so a call takes its share of four.
  static std::atomic<int> in_flight{0};
  struct InFlight { std::atomic<int>& c; int k; explicit InFlight(std::atomic<int>& c_) : c(c_), k(c_.fetch_add(1) + 1) {} ~InFlight() { c.fetch_sub(1); } } mine(in_flight);
  const u32 share = std::max<u32>(1u, max_slices / (u32)std::max(1, mine.k));
  const u32 S = std::max<u32>(1u, std::min<u32>(share, rm ? (u32)std::min<u64>(n, (n ? offsets[n] : 0) / slice_min_bytes) : n / slice_min));
  const u32 words = (t->wire.n_policies + 31u) / 32u;
  struct Slice {
    u32 lo = 0, hi = 0; cbh_device_batch* b = nullptr; cbh_wire_info wi{}; size_t total = 0, base = 0; int rc = 0; std::string err;
    std::vector<uint64_t> off, ooff, aoff; std::vector<uint32_t> first; u32 n_in = 0, in_base = 0;
  };
  std::vector<Slice> sl(S);
  // Even slices.  (A smaller LAST slice - what the call waits for at the end is that slice's road after the last message has gone up -
  // was measured and lost: 1.98 ms a call against 1.79 ms, the larger slices in front delay everything behind them.
  // CBH_WIRE_LAST_SLICE=percent of an even share for the last one.)
  static const double last_share = [] { const char* e = getenv("CBH_WIRE_LAST_SLICE"); const double v = e ? atof(e) / 100.0 : 1.0; return v < 0.1 ? 0.1 : v > 1.0 ? 1.0 : v; }();
  {
    const double unit = (double)n / ((double)(S - 1) + (S > 1 ? last_share : 1.0));
    u32 at = 0;
    for (u32 k = 0; k < S; ++k) {
      sl[k].lo = at;
      at = (k + 1 == S) ? n : std::min<u32>(n, (u32)(unit * (double)(k + 1) + 0.5));
      if (at < sl[k].lo) at = sl[k].lo;
      sl[k].hi = at;
    }
  }
  // the slices' uploads in slice order (WireChain)
  if (device_index >= t->reps.size()) return fail("device index out of range");
  HIPCHK(hipSetDevice(t->reps[device_index]->device));
  std::vector<hipEvent_t> evs(S, nullptr);
  std::vector<std::atomic<int>> recorded(S);
  std::vector<WireChain> chains(S);
  struct EvGuard { std::vector<hipEvent_t>& e; ~EvGuard() { for (auto x : e) if (x) (void)hipEventDestroy(x); } } ev_guard{evs};
  for (u32 k = 0; k < S; ++k) {
    recorded[k].store(0);
    if (S > 1 && hipEventCreateWithFlags(&evs[k], hipEventDisableTiming) != hipSuccess) return fail("cbh_wire_check_pb: hipEventCreate failed");
    chains[k].record = evs[k]; chains[k].recorded = &recorded[k];
    if (k) { chains[k].wait = evs[k - 1]; chains[k].prev_recorded = &recorded[k - 1]; }
  }
  // the trail of a slice of requests (cbh_check_batch_trail with one group per REQUEST): the inputs' groups follow from the split's
  // first_input; a batch the flattener grouped by route keeps its results by position, so the groups move with the inputs
  auto trail_on = [&](Slice& x, u32*& d_ep) -> int {
    cbh_device_batch* b = x.b;
    hipStream_t s = b->stream;
    const u32 cnt = x.hi - x.lo;
    const size_t ep_n = (size_t)(cnt ? cnt : 1u) * (words ? words : 1u);
    if (dalloc(b, d_ep, ep_n) != 0) return -1;
    HIPCHK(hipMemsetAsync(d_ep, 0, ep_n * 4, s));
    u32* d_grp = nullptr;
    if (x.n_in) {
      std::vector<u32> grp(x.n_in);
      for (u32 r = 0; r < cnt; ++r) for (u32 i = x.first[r]; i < x.first[r + 1]; ++i) grp[i] = r;
      u32* d_by_input = nullptr;
      if (dalloc(b, d_by_input, (size_t)x.n_in) != 0) return -1;
      HIPCHK(hipMemcpyAsync(d_by_input, grp.data(), (size_t)x.n_in * 4, hipMemcpyHostToDevice, s));
      d_grp = d_by_input;
      if (b->w_inv) {
        if (dalloc(b, d_grp, (size_t)x.n_in) != 0) return -1;
        WireScatterArgs sa; sa.by_input = d_by_input; sa.inv = b->w_inv; sa.by_position = d_grp; sa.n = x.n_in; sa.pad = 0;
        hipLaunchKernelGGL(cbh_wire_scatter_u32_kernel, dim3((x.n_in + 255u) / 256u), dim3(256), 0, s, sa);
        HIPCHK(hipGetLastError());
      }
      HIPCHK(hipStreamSynchronize(s));   // (grp is a pageable source going out of scope)
    }
    b->out.eff_pol = d_ep; b->out.ep_words = words; b->dev.ep_group = d_grp;
    return 0;
  };
  // stage 1 (per slice): flatten, decide, sizes of the outputs
  auto stage1 = [&](u32 k) {
    Slice& x = sl[k];
    const u32 cnt = x.hi - x.lo;
    x.off.resize((size_t)cnt + 1);
    const uint64_t o0 = n ? offsets[x.lo] : 0;
    for (u32 i = 0; i <= cnt; ++i) x.off[i] = (n ? offsets[x.lo + i] : 0) - o0;
    WireRequests rq;
    if (rm) {
      x.first.assign((size_t)cnt + 1, 0u);
      rq.first_input = x.first.data(); rq.flags = rm->request_flags ? rm->request_flags + x.lo : nullptr;
      if (rm->aux_offsets) {
        const uint64_t a0 = n ? rm->aux_offsets[x.lo] : 0;
        x.aoff.resize((size_t)cnt + 1);
        for (u32 i = 0; i <= cnt; ++i) x.aoff[i] = (n ? rm->aux_offsets[x.lo + i] : 0) - a0;   // (out of order: the device refuses the request)
        rq.aux = rm->aux ? rm->aux + a0 : nullptr; rq.aux_offsets = x.aoff.data();
      }
    }
    x.rc = wire_flatten_impl(t, device_index, bytes ? bytes + o0 : nullptr, x.off.data(), cnt, default_version, default_scope, globals_pb, globals_len, &x.b, &x.wi,
                             S > 1 ? &chains[k] : nullptr, rm ? &rq : nullptr);
    if (x.rc != 0) { x.err = g_err; x.b = nullptr; return; }
    x.n_in = x.wi.n_requests;   // the slice's messages (requests: the inputs made of them)
    cbh_params q = *p;
    q.flags &= ~(u32)CBH_F_WANT_EFFECTIVE_POLICIES;
    u32* d_ep = nullptr;
    if (rm && rm->effective_policies) {
      if (trail_on(x, d_ep) != 0) { x.rc = -1; x.err = g_err; return; }
      q.flags |= CBH_F_WANT_EFFECTIVE_POLICIES;
    }
    wmark("flattened");
    x.rc = cbh_check_resident(t, x.b, &q);
    if (x.rc != 0) { x.err = g_err; return; }
    wmark("decision enqueued");
    if (d_ep && words && cnt) {
      if (hipMemcpyAsync(rm->effective_policies + (size_t)x.lo * words, d_ep, (size_t)cnt * words * 4, hipMemcpyDeviceToHost, x.b->stream) != hipSuccess ||
          hipStreamSynchronize(x.b->stream) != hipSuccess) { x.rc = fail("cbh_wire_check_requests_trail_pb: download failed"); x.err = g_err; return; }
    }
    x.ooff.resize((size_t)x.n_in + 1);
    size_t nd = 0;
    const int r = cbh_wire_outputs(t, x.b, nullptr, 0, x.ooff.data(), nullptr, &nd);   // cap 0: sizes only (2 = "too small" unless the slice has no output bytes)
    if (r != 0 && r != 2) { x.rc = r; x.err = g_err; return; }
    x.total = nd;
    wmark("sizes known");
  };
  auto stage2 = [&](u32 k) {
    Slice& x = sl[k];
    size_t nd = 0;
    x.rc = cbh_wire_outputs(t, x.b, out_bytes + x.base, x.total, x.ooff.data(), out_flags ? out_flags + x.in_base : nullptr, &nd);
    if (x.rc != 0) { x.err = g_err; return; }
    for (u32 i = 0; i <= x.n_in; ++i) out_offsets[x.in_base + i] = x.ooff[i] + x.base;
    wmark("written + copied back");
  };
  // A slice writes as soon as the slices before it know their sizes (its base is their sum): no barrier between the stages, so
  // the first slice's answers are on their way back while the last slice's messages are still going up.  A slice that failed,
  // or met a message for the host flattener, publishes "no size": nobody writes after that.
  std::vector<std::atomic<int>> sized(S);   // 0 not yet, 1 size known, 2 failed
  for (auto& q : sized) q.store(0);
  std::atomic<int> overflow{0};
  const size_t inputs_cap = rm ? (out_offsets ? rm->out_inputs_cap : 0) : (size_t)n;
  const auto call_t0 = std::chrono::steady_clock::now();
  std::vector<WireMarks> marks(trace_on() ? S : 0);
  auto work = [&](u32 k) {
    if (!marks.empty()) { marks[k].t0 = call_t0; tl_marks = &marks[k]; }
    struct Untrace { ~Untrace() { tl_marks = nullptr; } } untrace;
    wmark("thread runs");
    stage1(k);
    Slice& x = sl[k];
    sized[k].store(x.rc == 0 ? 1 : 2, std::memory_order_release);
    if (x.rc != 0) return;
    size_t base = 0; u32 in_base = 0;
    for (u32 j = 0; j < k; ++j) {
      int st;
      while ((st = sized[j].load(std::memory_order_acquire)) == 0) std::this_thread::yield();
      if (st == 2) return;
      base += sl[j].total; in_base += sl[j].n_in;
    }
    x.base = base; x.in_base = in_base;
    wmark("predecessors sized");
    if (base + x.total > out_cap || (size_t)in_base + x.n_in > inputs_cap) { overflow.store(1); return; }
    stage2(k);
  };
  {
    std::vector<std::thread> th;
    for (u32 k = 1; k < S; ++k) th.emplace_back(work, k);
    work(0);
    for (auto& q : th) q.join();
  }
  if (!marks.empty()) {
    const double end = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - call_t0).count();
    for (u32 k = 0; k < S; ++k) {
      std::string line = "[cbh] wire slice " + std::to_string(k) + ":";
      for (auto& m : marks[k].v) { char buf[96]; std::snprintf(buf, sizeof(buf), "  %s %.0f", m.first, m.second); line += buf; }
      std::fprintf(stderr, "%s  | joined %.0f us\n", line.c_str(), end);
    }
  }
  auto release = [&] { for (auto& x : sl) if (x.b) { cbh_batch_release(x.b); x.b = nullptr; } };
  int rc = 0; std::string err;
  size_t total = 0; u64 inputs = 0;
  for (auto& x : sl) {
    info->n_tuples += x.wi.n_tuples; info->n_host += x.wi.n_host; info->heap_len += x.wi.heap_len; info->dict_slots += x.wi.dict_slots;
    info->fill_runs = std::max(info->fill_runs, x.wi.fill_runs); info->n_routes = std::max(info->n_routes, x.wi.n_routes);
    if (x.wi.first_bad != CBH_NONE && info->first_bad == CBH_NONE) info->first_bad = x.lo + x.wi.first_bad;
    if (x.rc < 0 && rc >= 0) { rc = x.rc; err = x.err; }
    else if (x.rc == 1 && rc == 0) { rc = 1; err = x.err; }
    total += x.total;
    if (rm && x.rc == 0) { for (u32 r = 0; r <= x.hi - x.lo; ++r) rm->first_input[x.lo + r] = (u32)inputs + x.first[r]; }
    inputs += x.n_in;
  }
  release();
  if (rm) info->n_requests = (u32)inputs;
  if (rc != 0) { g_err = err; return rc; }
  *need = total;
  if (overflow.load() || total > out_cap || inputs > inputs_cap) {
    g_err = rm ? "cbh_wire_check_requests_pb: the output buffer (or out_offsets / out_flags) is too small" : "cbh_wire_check_pb: the output buffer is too small";
    return 2;
  }
  if (inputs == 0 && out_offsets) out_offsets[0] = 0;
  return 0;
}
extern "C" int cbh_wire_check_pb(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                                 const char* default_version, const char* default_scope, const uint8_t* globals_pb, size_t globals_len,
                                 const cbh_params* p, uint8_t* out_bytes, size_t out_cap, uint64_t* out_offsets, uint8_t* out_flags,
                                 size_t* need, cbh_wire_info* info) {
  if (!t || !p || !out_offsets || !need || !info || (n && (!bytes || !offsets)) || (out_cap && !out_bytes)) return fail("null argument");
  return wire_check_sliced(t, device_index, bytes, offsets, n, default_version, default_scope, globals_pb, globals_len, p, out_bytes, out_cap, out_offsets, out_flags,
                           need, info, nullptr);
}
// ---- the same call without blocking the caller (include/cerbos_hip.h cbh_wire_check_pb_submit / _collect).  The ticket owns a
// worker thread that makes the synchronous call; the strings are copied, the buffers are the caller's and stay untouched until
// collect.  A caller that keeps two tickets in flight has the second call's uploads under the first's downloads - the fill and
// drain of one call's slices are what a lone synchronous caller pays on top of the link's own time.
struct cbh_wire_ticket {
  cbh_table* table = nullptr;   // (the reference submit took: released by collect, whatever table the caller names there)
  std::thread worker;
  std::string ver, scope, err;
  int rc = -1;
  size_t need = 0;
  cbh_wire_info info{};
};
extern "C" int cbh_wire_check_pb_submit(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                                        const char* default_version, const char* default_scope, const uint8_t* globals_pb, size_t globals_len,
                                        const cbh_params* p, uint8_t* out_bytes, size_t out_cap, uint64_t* out_offsets, uint8_t* out_flags,
                                        cbh_wire_ticket** ticket) {
  if (!ticket) return fail("null argument");
  *ticket = nullptr;
  if (!t || !p || !out_offsets || (n && (!bytes || !offsets)) || (out_cap && !out_bytes)) return fail("null argument");
  cbh_wire_ticket* k = new (std::nothrow) cbh_wire_ticket();
  if (!k) return fail("out of memory");
  k->ver = default_version ? default_version : ""; k->scope = default_scope ? default_scope : "";
  const bool has_ver = default_version != nullptr, has_scope = default_scope != nullptr;
  const cbh_params params = *p;
  cbh_table_retain(t);   // the table outlives the call whatever the caller does with its own reference meanwhile
  k->table = t;
  try {
    k->worker = std::thread([=]() {
      k->rc = wire_check_sliced(t, device_index, bytes, offsets, n, has_ver ? k->ver.c_str() : nullptr, has_scope ? k->scope.c_str() : nullptr, globals_pb, globals_len,
                                &params, out_bytes, out_cap, out_offsets, out_flags, &k->need, &k->info, nullptr);
      if (k->rc != 0) k->err = g_err;   // (the worker's own thread-local message: handed to the collecting thread)
    });
  } catch (...) { cbh_table_release(t); delete k; return fail("cbh_wire_check_pb_submit: cannot start a worker thread"); }
  *ticket = k;
  return 0;
}
extern "C" int cbh_wire_check_pb_collect(cbh_table* t, cbh_wire_ticket* ticket, size_t* need, cbh_wire_info* info) {
  if (!ticket) return fail("null argument");
  if (t && t != ticket->table) return fail("cbh_wire_check_pb_collect: the ticket was issued for another table");   // (the ticket stays valid)
  if (ticket->worker.joinable()) ticket->worker.join();
  const int rc = ticket->rc;
  if (need) *need = ticket->need;
  if (info) *info = ticket->info;
  if (rc != 0) g_err = ticket->err;
  cbh_table* held = ticket->table;
  delete ticket;
  cbh_table_release(held);   // submit's reference
  return rc;
}
static int wire_check_requests_impl(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n_requests,
                                    const uint8_t* aux_bytes, const uint64_t* aux_offsets, const char* default_version, const char* default_scope,
                                    const uint8_t* globals_pb, size_t globals_len, const cbh_params* p, uint32_t* first_input, uint8_t* request_flags,
                                    uint8_t* out_bytes, size_t out_cap, uint64_t* out_offsets, uint8_t* out_flags, size_t out_inputs_cap, size_t* need,
                                    cbh_wire_info* info, uint32_t* effective_policies) {
  if (!t || !p || !need || !info || !first_input || (n_requests && (!bytes || !offsets)) || (out_cap && !out_bytes)) return fail("null argument");
  if ((aux_bytes == nullptr) != (aux_offsets == nullptr)) return fail("cbh_wire_check_requests_pb: aux_bytes and aux_offsets go together");
  WireReqMode rm;
  rm.aux = aux_bytes; rm.aux_offsets = aux_offsets; rm.first_input = first_input; rm.request_flags = request_flags; rm.out_inputs_cap = out_inputs_cap;
  rm.effective_policies = effective_policies;
  return wire_check_sliced(t, device_index, bytes, offsets, n_requests, default_version, default_scope, globals_pb, globals_len, p, out_bytes, out_cap,
                           out_offsets, out_flags, need, info, &rm);
}
extern "C" int cbh_wire_check_requests_pb(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n_requests,
                                          const uint8_t* aux_bytes, const uint64_t* aux_offsets, const char* default_version, const char* default_scope,
                                          const uint8_t* globals_pb, size_t globals_len, const cbh_params* p, uint32_t* first_input, uint8_t* request_flags,
                                          uint8_t* out_bytes, size_t out_cap, uint64_t* out_offsets, uint8_t* out_flags, size_t out_inputs_cap, size_t* need,
                                          cbh_wire_info* info) {
  return wire_check_requests_impl(t, device_index, bytes, offsets, n_requests, aux_bytes, aux_offsets, default_version, default_scope, globals_pb, globals_len, p,
                                  first_input, request_flags, out_bytes, out_cap, out_offsets, out_flags, out_inputs_cap, need, info, nullptr);
}
// ... and with the audit trail of every request: effective_policies[r * words + w] (words = (cbh_table_num_policies + 31) / 32) has bit
// k set when policy k was among those the engine went through for ANY resource entry of request r - AuditTrail.EffectivePolicies of
// the one decision-log entry the server writes for the call (check.go:302-304, svc CheckResources: one entry per request).
extern "C" int cbh_wire_check_requests_trail_pb(cbh_table* t, uint32_t device_index, const uint8_t* bytes, const uint64_t* offsets, uint32_t n_requests,
                                                const uint8_t* aux_bytes, const uint64_t* aux_offsets, const char* default_version, const char* default_scope,
                                                const uint8_t* globals_pb, size_t globals_len, const cbh_params* p, uint32_t* first_input, uint8_t* request_flags,
                                                uint8_t* out_bytes, size_t out_cap, uint64_t* out_offsets, uint8_t* out_flags, size_t out_inputs_cap, size_t* need,
                                                cbh_wire_info* info, uint32_t* effective_policies) {
  if (!effective_policies) return fail("null argument");
  return wire_check_requests_impl(t, device_index, bytes, offsets, n_requests, aux_bytes, aux_offsets, default_version, default_scope, globals_pb, globals_len, p,
                                  first_input, request_flags, out_bytes, out_cap, out_offsets, out_flags, out_inputs_cap, need, info, effective_policies);
}

// ---- one-shot path: CheckResources round trip for a host batch ------------------------------------------
// All arrays of the batch go into ONE device block per device used, laid out for the whole batch; a device
// that decides the request range [lo, hi) receives only the slices of that range (the kernels address the
// whole-batch layout through BatchDev.req_lo / req_hi).
static OneShot* ctx_acquire(Replica* r) {
  std::unique_lock<std::mutex> lk(r->ctx_mu);
  for (;;) {
    if (!r->ctx_idle.empty()) { auto* c = r->ctx_idle.back(); r->ctx_idle.pop_back(); return c; }
    if (r->ctx_count < Replica::MAX_ONESHOT) {
      ++r->ctx_count;
      lk.unlock();
      auto* c = new (std::nothrow) OneShot();
      bool ok = c != nullptr;
      for (int i = 0; ok && i < N_STREAMS; ++i) ok = hipStreamCreateWithFlags(&c->s[i], hipStreamNonBlocking) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&c->ev_setup, hipEventDisableTiming) == hipSuccess;
      if (c) for (auto& e : c->ev_piece) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
      if (!ok) {
        if (c) { for (auto& s : c->s) if (s) (void)hipStreamDestroy(s); if (c->ev_setup) (void)hipEventDestroy(c->ev_setup); for (auto& e : c->ev_piece) if (e) (void)hipEventDestroy(e); delete c; c = nullptr; }
        lk.lock(); --r->ctx_count; r->ctx_cv.notify_one();
      }
      return c;
    }
    r->ctx_cv.wait(lk);
  }
}
struct CtxLease {
  Replica* r; OneShot* c;
  int used = N_STREAMS;   // streams the call has touched
  ~CtxLease() {
    if (!c) return;
    (void)hipSetDevice(r->device);
    for (int i = 0; i < used; ++i) (void)hipStreamSynchronize(c->s[i]);   // an error return must not leave copies from caller memory in flight
    { std::lock_guard<std::mutex> lk(r->ctx_mu); r->ctx_idle.push_back(c); }
    r->ctx_cv.notify_one();
  }
};
static int ctx_reserve(OneShot* c, size_t hbytes, size_t dbytes) {
  if (hbytes > c->h_cap) {
    if (c->h) { (void)hipHostFree(c->h); c->h = nullptr; c->h_cap = 0; }
    size_t cap = 1 << 16; while (cap < hbytes) cap <<= 1;
    HIPCHK(hipHostMalloc((void**)&c->h, cap, hipHostMallocDefault));
    c->h_cap = cap;
  }
  if (dbytes && dbytes > c->d_cap) {
    if (c->d) { (void)hipFree(c->d); c->d = nullptr; c->d_cap = 0; }
    size_t cap = 1 << 16; while (cap < dbytes) cap <<= 1;
    HIPCHK(hipMalloc((void**)&c->d, cap));
    c->d_cap = cap;
  }
  return 0;
}

static bool is_pinned(const void* p) {
  if (!p) return true;   // an absent array does not decide
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

struct Seg { size_t off, bytes; const void* src; };
// The canonical order of a batch's arrays.  It is the order of the device block AND of a caller's "slab" (one
// page-locked block holding all arrays, cbh_batch_bind_slab): a slab crosses PCIe in ONE copy because the device
// block mirrors it byte for byte.  What a table may not need comes last - the raw request strings (req_u32 rows
// CBH_RQ_NCORE..), then the batch-local string pool - so that the one copy simply stops earlier.
struct InOffsets { size_t roles, act, ctag, cval, htag, hval, req, soff, sbytes, sflags, end; };
static InOffsets in_offsets(const cbh_batch* in) {
  InOffsets o; size_t cur = 0;
  const size_t NR = in->n_requests, NT = in->n_tuples, NS = in->n_strings;
  auto seg = [&](size_t bytes) { const size_t at = cur; cur += (bytes + 255) & ~(size_t)255; return at; };
  o.roles = seg((size_t)in->n_roles * 4); o.act = seg(NT * 4);
  o.ctag = seg((size_t)in->n_columns * NR + 4);   // + 4: a lane reads the aligned dword around its tag byte
  o.cval = seg((size_t)in->n_columns * NR * 8);
  o.htag = seg(in->heap_len); o.hval = seg((size_t)in->heap_len * 8);
  o.req = seg((size_t)CBH_RQ_NFIELDS * NR * 4);
  o.soff = seg(NS ? (NS + 1) * 4 : 0); o.sbytes = seg(in->str_bytes_len); o.sflags = seg(NS);
  o.end = cur;
  return o;
}
struct OutOffsets { size_t eff, status, pol, scope, edr, end; };
static OutOffsets out_offsets(size_t NT, size_t NR) {
  OutOffsets o; size_t cur = 0;
  auto seg = [&](size_t bytes) { const size_t at = cur; cur += (bytes + 255) & ~(size_t)255; return at; };
  o.eff = seg(NT); o.status = seg(NT); o.pol = seg(NT * 4); o.scope = seg(NT * 4); o.edr = seg(NR * 8);
  o.end = cur;
  return o;
}
extern "C" size_t cbh_batch_slab_bytes(const cbh_batch* counts) { return counts ? in_offsets(counts).end : 0; }
extern "C" void cbh_batch_bind_slab(cbh_batch* b, void* slab) {
  if (!b || !slab) return;
  const InOffsets o = in_offsets(b);
  uint8_t* p = static_cast<uint8_t*>(slab);
  b->roles = (const uint32_t*)(p + o.roles); b->tuple_req = nullptr; b->tuple_action = (const uint32_t*)(p + o.act);
  b->col_tag = p + o.ctag; b->col_val = (const uint64_t*)(p + o.cval); b->heap_tag = p + o.htag; b->heap_val = (const uint64_t*)(p + o.hval);
  b->req_u32 = (const uint32_t*)(p + o.req); b->str_off = (const uint32_t*)(p + o.soff); b->str_bytes = p + o.sbytes; b->str_flags = p + o.sflags;
}
extern "C" size_t cbh_result_slab_bytes(uint32_t n_tuples, uint32_t n_requests) { return out_offsets(n_tuples, n_requests).end; }
extern "C" void cbh_result_bind_slab(cbh_result* r, void* slab, uint32_t n_tuples, uint32_t n_requests) {
  if (!r || !slab) return;
  const OutOffsets o = out_offsets(n_tuples, n_requests);
  uint8_t* p = static_cast<uint8_t*>(slab);
  r->effect = p + o.eff; r->status = p + o.status; r->policy = (uint32_t*)(p + o.pol); r->scope = (uint32_t*)(p + o.scope); r->edr_mask = (uint64_t*)(p + o.edr);
}

struct Layout {
  Seg args, req, roles, act, ctag, cval, htag, hval, soff, sbytes, sflags, gbits, gres, eff, pol, scope, status, edr;
  size_t in_begin, in_end, out_begin, total;
};
static Layout make_layout(const cbh_batch* in, const cbh_table* t) {
  Layout L;
  const size_t NR = in->n_requests, NT = in->n_tuples, NS = in->n_strings;
  const InOffsets io = in_offsets(in);
  const size_t A = (sizeof(KernelArgs) + 255) & ~(size_t)255;
  L.args = Seg{0, sizeof(KernelArgs), nullptr};
  L.in_begin = A;
  L.roles = Seg{A + io.roles, (size_t)in->n_roles * 4, in->roles}; L.act = Seg{A + io.act, NT * 4, in->tuple_action};
  L.ctag = Seg{A + io.ctag, (size_t)in->n_columns * NR, in->col_tag}; L.cval = Seg{A + io.cval, (size_t)in->n_columns * NR * 8, in->col_val};
  L.htag = Seg{A + io.htag, in->heap_len, in->heap_tag}; L.hval = Seg{A + io.hval, (size_t)in->heap_len * 8, in->heap_val};
  L.req = Seg{A + io.req, (size_t)CBH_RQ_NFIELDS * NR * 4, in->req_u32};
  L.soff = Seg{A + io.soff, NS ? (NS + 1) * 4 : 0, in->str_off}; L.sbytes = Seg{A + io.sbytes, in->str_bytes_len, in->str_bytes};
  L.sflags = Seg{A + io.sflags, NS, in->str_flags};
  L.in_end = A + io.end;
  size_t cur = L.in_end;
  L.gbits = Seg{cur, 3 * NS * 8, nullptr}; cur += (L.gbits.bytes + 255) & ~(size_t)255;
  // results of the evaluation sites (cbh_walk2_pre_kernel -> cbh_walk2_kernel), sized for a batch that needs all of them
  const size_t gw = (t->meta[CBH_M_FLAGS] & CBH_MF_WALK2) ? w2_gwords(t->meta[CBH_M_GSLOTS_GENERIC], t->meta[CBH_M_GSLOTS_ALL], false) : 0;
  L.gres = Seg{cur, gw * NR * 8, nullptr}; cur += (L.gres.bytes + 255) & ~(size_t)255;
  L.out_begin = cur;
  const OutOffsets oo = out_offsets(NT, NR);
  L.eff = Seg{cur + oo.eff, NT, nullptr}; L.status = Seg{cur + oo.status, NT, nullptr}; L.pol = Seg{cur + oo.pol, NT * 4, nullptr};
  L.scope = Seg{cur + oo.scope, NT * 4, nullptr}; L.edr = Seg{cur + oo.edr, NR * 8, nullptr};
  L.total = cur + oo.end;
  return L;
}
// are the batch's arrays one slab in canonical order?  -> its base address, else nullptr
static const uint8_t* slab_base(const Layout& L) {
  const uint8_t* base = nullptr;
  for (const Seg* g : {&L.roles, &L.act, &L.ctag, &L.cval, &L.htag, &L.hval, &L.req, &L.soff, &L.sbytes, &L.sflags}) {
    if (!g->bytes) continue;
    const uint8_t* b = static_cast<const uint8_t*>(g->src) - (g->off - L.in_begin);
    if (!base) base = b; else if (b != base) return nullptr;
  }
  return base;
}
static void bind_args(KernelArgs& ka, const TableDev& tdev, const cbh_batch* in, const cbh_params* p, const Layout& L, uint8_t* base) {
  std::memset(&ka, 0, sizeof(ka));
  ka.t = tdev; ka.now_ns = p->now_ns; ka.flags = p->flags & ~(u32)CBH_FI_MASK;
  BatchDev& d = ka.b;
  d.n_requests = in->n_requests; d.n_tuples = in->n_tuples; d.n_roles = in->n_roles;
  d.n_columns = in->n_columns; d.n_strings = in->n_strings; d.heap_len = in->heap_len;
  d.req_lo = 0; d.req_hi = in->n_requests;
  d.req_u32 = (const u32*)(base + L.req.off); d.roles = (const u32*)(base + L.roles.off); d.tuple_req = nullptr;
  d.tuple_action = (const u32*)(base + L.act.off); d.col_tag = base + L.ctag.off; d.col_val = (const u64*)(base + L.cval.off);
  d.heap_tag = base + L.htag.off; d.heap_val = (const u64*)(base + L.hval.off); d.str_off = (const u32*)(base + L.soff.off);
  d.str_bytes = base + L.sbytes.off; d.str_flags = base + L.sflags.off; d.gbits = (u64*)(base + L.gbits.off);
  d.gres = L.gres.bytes ? (u64*)(base + L.gres.off) : nullptr; d.n_gwords = 0; d.n_gslots = 0;   // n_gwords: per launch (launch_plan)
  ka.o.effect = base + L.eff.off; ka.o.policy = (u32*)(base + L.pol.off); ka.o.scope = (u32*)(base + L.scope.off);
  ka.o.status = base + L.status.off; ka.o.edr = (u64*)(base + L.edr.off);
}
static void launch_resolve(const Replica* rep, const KernelArgs& ka, const Layout& L, hipStream_t s, int& rc) {
  const u32 maxw = nfa_maxw(rep->dev);
  if (!ka.b.n_strings) return;
  if (maxw) {
    const u32 grid = (ka.b.n_strings + CBH_BLOCK - 1) / CBH_BLOCK;
    hipLaunchKernelGGL(cbh_resolve_globs_kernel, dim3(grid), dim3(CBH_BLOCK), (size_t)(2 + 512) * maxw * sizeof(u64), s, rep->dev, ka.b);
  } else if (hipMemsetAsync(ka.b.gbits, 0, L.gbits.bytes, s) != hipSuccess) rc = -1;   // no automata: no string matches a glob
}
static void launch_check(const Replica* rep, KernelArgs ka, const KernelArgs* d_args, u32 lo, u32 hi, const BatchShape& sh, hipStream_t s) {
  if (hi <= lo) return;
  launch_plan(plan_for(rep->dev, sh.max_actions, sh.max_roles, sh.plain_tags(), ka.flags), rep->dev, ka, d_args, lo, hi, sh.wide_lo, sh.wide_hi, 0, s);
}

// a small batch on one device: everything packed into the pinned staging block.  Two ways across PCIe:
//   copy      one H2D of the inputs, kernels on device memory, one D2H of the results (three queue operations);
//   zero-copy the kernels read the inputs from, and write the results to, the page-locked block itself (it is
//             mapped into the device's address space): a few KB per request wave over PCIe, ONE queue operation.
// Zero-copy wins while a batch is a handful of waves (the latency case); the choice is by input size.
static int run_small(cbh_table* t, Replica* rep, const cbh_batch* in, const cbh_params* p, cbh_result* out, const BatchShape& sh, const Layout& L) {
  HIPCHK(hipSetDevice(rep->device));
  CtxLease lease{rep, ctx_acquire(rep)};
  OneShot* c = lease.c;
  if (!c) return fail("could not create a launch context");
  static const size_t zc_limit = [] { const char* e = getenv("CBH_ZEROCOPY_BYTES"); return e ? (size_t)atol(e) : (size_t)(64 << 10); }();
  const bool zero_copy = L.in_end <= zc_limit;
  if (ctx_reserve(c, L.total, zero_copy ? 0 : L.total) != 0) return -1;
  hipStream_t s = c->s[0];
  lease.used = 1;
  const double t_0 = trace_on() ? now_us() : 0;
  uint8_t* base = c->d;
  if (zero_copy) HIPCHK(hipHostGetDevicePointer((void**)&base, c->h, 0));
  KernelArgs ka;
  bind_args(ka, rep->dev, in, p, L, base);
  std::memcpy(c->h + L.args.off, &ka, sizeof(ka));
  for (const Seg* g : {&L.req, &L.roles, &L.act, &L.ctag, &L.cval, &L.htag, &L.hval, &L.soff, &L.sbytes, &L.sflags})
    if (g->bytes) std::memcpy(c->h + g->off, g->src, g->bytes);
  int rc = 0;
  if (zero_copy) {
    if (in->n_strings && !nfa_maxw(rep->dev)) std::memset(c->h + L.gbits.off, 0, L.gbits.bytes);   // no automata: no string matches a glob
    else launch_resolve(rep, ka, L, s, rc);
  } else {
    HIPCHK(hipMemcpyAsync(c->d, c->h, L.in_end, hipMemcpyHostToDevice, s));
    launch_resolve(rep, ka, L, s, rc);
  }
  launch_check(rep, ka, (const KernelArgs*)(base + L.args.off), 0, in->n_requests, sh, s);
  HIPCHK(hipGetLastError());
  if (rc != 0) return fail("hipMemsetAsync failed");
  if (!zero_copy && L.total > L.out_begin) HIPCHK(hipMemcpyAsync(c->h + L.out_begin, c->d + L.out_begin, L.total - L.out_begin, hipMemcpyDeviceToHost, s));
  const double t_1 = trace_on() ? now_us() : 0;
  HIPCHK(stream_wait(s));
  if (trace_on()) std::fprintf(stderr, "[cbh] small zero_copy=%d in=%zu B enqueue=%.1f us wait=%.1f us\n", (int)zero_copy, L.in_end, t_1 - t_0, now_us() - t_1);
  struct Dst { const Seg* g; void* dst; };
  const Dst outs[5] = {{&L.eff, out->effect}, {&L.pol, out->policy}, {&L.scope, out->scope}, {&L.status, out->status}, {&L.edr, out->edr_mask}};
  for (const Dst& o : outs) if (o.dst && o.g->bytes) std::memcpy(o.dst, c->h + o.g->off, o.g->bytes);
  (void)t;
  return 0;
}

// the request range [lo, hi) of a large batch on one device
// the bytes of a slab that go up: it stops before the raw request strings / the string pool when the table reads neither
static size_t slab_upload_end(const Replica* rep, const Layout& L, size_t NR) {
  const bool reads_strings = (rep->dev.flags & CBH_MF_READS_REQUEST_STRINGS) != 0, need_bytes = (rep->dev.flags & CBH_MF_NEEDS_STRING_BYTES) != 0;
  return need_bytes ? L.in_end : L.req.off + (size_t)(reads_strings ? CBH_RQ_NFIELDS : CBH_RQ_NCORE) * NR * 4;
}
// One DMA engine does not fill the host link (a 29 MB slab went up at ~39 GB/s where the link gives 56): a large slab goes up
// in pieces on the context's streams - each stream's copies run on an engine of their own - and stream 0 waits for all.
static int slab_upload(OneShot* c, const Replica* rep, const Layout& L, const uint8_t* slab, size_t NR) {
  static const u32 slab_split = [] { const char* e = getenv("CBH_SLAB_SPLIT"); const long v = e ? atol(e) : 2; return (u32)std::min<long>(std::max<long>(v, 1), N_STREAMS); }();
  uint8_t* base = c->d;
  const size_t up = slab_upload_end(rep, L, NR) - L.in_begin;
  const u32 pieces = up >= ((size_t)8 << 20) ? slab_split : 1u;
  if (pieces <= 1) { HIPCHK(hipMemcpyAsync(base + L.in_begin, slab, up, hipMemcpyHostToDevice, c->s[0])); return 0; }
  const size_t step = ((up / pieces) + 4095) & ~(size_t)4095;
  for (u32 i = 0; i < pieces; ++i) {
    const size_t o = (size_t)i * step, n = o >= up ? 0 : std::min(step, up - o);
    if (!n) break;
    HIPCHK(hipMemcpyAsync(base + L.in_begin + o, slab + o, n, hipMemcpyHostToDevice, c->s[i]));
    if (i) { HIPCHK(hipEventRecord(c->ev_piece[i], c->s[i])); HIPCHK(hipStreamWaitEvent(c->s[0], c->ev_piece[i], 0)); }
  }
  return 0;
}
// `pre`: a context the caller holds whose slab upload is already in flight (cbh_check_batch starts it before it validates)
static int run_range(cbh_table* t, Replica* rep, const cbh_batch* in, const cbh_params* p, cbh_result* out, const BatchShape& sh,
                     const Layout& L, u32 lo, u32 hi, bool pinned, u32 chunk_requests, OneShot* pre = nullptr) {
  HIPCHK(hipSetDevice(rep->device));
  CtxLease lease{rep, pre ? nullptr : ctx_acquire(rep)};
  OneShot* c = pre ? pre : lease.c;
  if (!c) return fail("could not create a launch context");
  const double t_0 = trace_on() ? now_us() : 0;
  if (!pre && ctx_reserve(c, 4096, L.total) != 0) return -1;
  const size_t NR = in->n_requests;
  const bool whole = lo == 0 && hi == NR;
  KernelArgs ka;
  uint8_t* base = c->d;
  bind_args(ka, rep->dev, in, p, L, base);
  std::memcpy(c->h, &ka, sizeof(ka));
  const KernelArgs* d_args = (const KernelArgs*)(base + L.args.off);
  const u32* act_off = in->req_u32 + (size_t)CBH_RQ_ACT_OFF * NR; const u32* act_cnt = in->req_u32 + (size_t)CBH_RQ_ACT_CNT * NR;
  // tuples of the requests [a, b), a < b (ACT_OFF ascends whenever a batch is split; a batch in any other
  // order is only ever handled whole)
  auto tuples_of = [&](u32 a, u32 b, size_t& tb, size_t& te) {
    if (!sh.ascending) { tb = 0; te = in->n_tuples; return; }
    tb = act_off[a]; te = (size_t)act_off[b - 1] + act_cnt[b - 1];
  };
  const bool reads_strings = (rep->dev.flags & CBH_MF_READS_REQUEST_STRINGS) != 0;

  const bool need_bytes = (rep->dev.flags & CBH_MF_NEEDS_STRING_BYTES) != 0;
  hipStream_t s0 = c->s[0];
  int rc = 0;

  // ---- a slab (cbh_batch_bind_slab) in page-locked memory, decided whole on this device: ONE copy up - it stops
  // before the raw request strings / the string pool when the table reads neither -, the kernels, and the
  // results down in as few copies as the caller's result arrays are contiguous (one for a result slab)
  const uint8_t* slab = (whole && pinned) ? slab_base(L) : nullptr;
  if (slab) {
    const size_t end = slab_upload_end(rep, L, NR);
    HIPCHK(hipMemcpyAsync(base, c->h, sizeof(ka), hipMemcpyHostToDevice, s0));
    if (!pre && slab_upload(c, rep, L, slab, NR) != 0) return -1;
    launch_resolve(rep, ka, L, s0, rc);
    if (rc != 0) return fail("hipMemsetAsync failed");
    launch_check(rep, ka, d_args, 0, (u32)NR, sh, s0);
    HIPCHK(hipGetLastError());
    struct Run { size_t off, bytes; uint8_t* dst; };
    Run run{0, 0, nullptr};
    const Seg* segs[5] = {&L.eff, &L.status, &L.pol, &L.scope, &L.edr};
    void* dsts[5] = {out->effect, out->status, out->policy, out->scope, out->edr_mask};
    for (int i = 0; i < 5; ++i) {
      if (!dsts[i] || !segs[i]->bytes) continue;
      uint8_t* d = static_cast<uint8_t*>(dsts[i]);
      if (run.dst && d == run.dst + (segs[i]->off - run.off)) { run.bytes = segs[i]->off + segs[i]->bytes - run.off; continue; }   // contiguous with the run: extend it
      if (run.dst) HIPCHK(hipMemcpyAsync(run.dst, base + run.off, run.bytes, hipMemcpyDeviceToHost, s0));
      run = Run{segs[i]->off, segs[i]->bytes, d};
    }
    if (run.dst) HIPCHK(hipMemcpyAsync(run.dst, base + run.off, run.bytes, hipMemcpyDeviceToHost, s0));
    const double t_1 = trace_on() ? now_us() : 0;
    HIPCHK(stream_wait(s0));
    if (trace_on()) std::fprintf(stderr, "[cbh] slab dev=%d up=%zu B enqueue=%.1f us wait=%.1f us\n", rep->device, end - L.in_begin, t_1 - t_0, now_us() - t_1);
    return 0;
  }

  // ---- setup on stream 0: launch arguments + the arrays that are not per request (roles, heap, strings)
  HIPCHK(hipMemcpyAsync(base, c->h, sizeof(ka), hipMemcpyHostToDevice, s0));
  for (const Seg* g : {&L.roles, &L.htag, &L.hval, &L.soff, &L.sbytes, &L.sflags}) {
    if (!need_bytes && (g == &L.soff || g == &L.sbytes || g == &L.sflags)) continue;   // no program looks inside a string
    if (g->bytes) HIPCHK(hipMemcpyAsync(base + g->off, g->src, g->bytes, hipMemcpyHostToDevice, s0));
  }
  launch_resolve(rep, ka, L, s0, rc);
  if (rc != 0) return fail("hipMemsetAsync failed");
  HIPCHK(hipEventRecord(c->ev_setup, s0));

  // rows [r0, r1) of a field-major [rows][NR] array of `esz`-byte elements, requests [a, b): one 2-D copy
  static const int copy_mode = [] { const char* e = getenv("CBH_COPY_MODE"); return e ? atoi(e) : 0; }();   // 1: a row at a time instead of 2-D copies
  auto up2d = [&](const Seg& g, size_t esz, u32 r0, u32 r1, u32 a, u32 b, hipStream_t s) -> hipError_t {
    if (r1 <= r0 || b <= a) return hipSuccess;
    const size_t pitch = NR * esz, o = (size_t)r0 * pitch + (size_t)a * esz;
    if (a == 0 && b == NR) return hipMemcpyAsync(base + g.off + o, (const uint8_t*)g.src + o, (size_t)(r1 - r0) * pitch, hipMemcpyHostToDevice, s);
    if (copy_mode == 1) {
      for (u32 r = r0; r < r1; ++r) {
        const size_t oo = (size_t)r * pitch + (size_t)a * esz;
        const hipError_t e = hipMemcpyAsync(base + g.off + oo, (const uint8_t*)g.src + oo, (size_t)(b - a) * esz, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    }
    return hipMemcpy2DAsync(base + g.off + o, pitch, (const uint8_t*)g.src + o, pitch, (size_t)(b - a) * esz, r1 - r0, hipMemcpyHostToDevice, s);
  };
  if (!pinned) {
    // pageable arrays: the driver stages every copy itself and the calling thread waits for it - chunking buys
    // nothing, so the range goes up array by array, is decided by one launch and comes down array by array
    hipStream_t s = s0;
    HIPCHK(up2d(L.req, 4, 0, reads_strings ? CBH_RQ_NFIELDS : CBH_RQ_NCORE, lo, hi, s));
    HIPCHK(up2d(L.ctag, 1, 0, in->n_columns, lo, hi, s));
    HIPCHK(up2d(L.cval, 8, 0, in->n_columns, lo, hi, s));
    size_t tb = 0, te = 0;
    if (hi > lo) tuples_of(lo, hi, tb, te);
    if (te > tb) HIPCHK(hipMemcpyAsync(base + L.act.off + tb * 4, in->tuple_action + tb, (te - tb) * 4, hipMemcpyHostToDevice, s));
    launch_check(rep, ka, d_args, lo, hi, sh, s);
    HIPCHK(hipGetLastError());
    if (te > tb) {
      HIPCHK(hipMemcpyAsync(out->effect + tb, base + L.eff.off + tb, te - tb, hipMemcpyDeviceToHost, s));
      if (out->policy) HIPCHK(hipMemcpyAsync(out->policy + tb, base + L.pol.off + tb * 4, (te - tb) * 4, hipMemcpyDeviceToHost, s));
      if (out->scope) HIPCHK(hipMemcpyAsync(out->scope + tb, base + L.scope.off + tb * 4, (te - tb) * 4, hipMemcpyDeviceToHost, s));
      if (out->status) HIPCHK(hipMemcpyAsync(out->status + tb, base + L.status.off + tb, te - tb, hipMemcpyDeviceToHost, s));
    }
    if (out->edr_mask && hi > lo) HIPCHK(hipMemcpyAsync(out->edr_mask + lo, base + L.edr.off + (size_t)lo * 8, (size_t)(hi - lo) * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (trace_on()) std::fprintf(stderr, "[cbh] range [%u,%u) dev=%d pageable total=%.1f us\n", lo, hi, rep->device, now_us() - t_0);
    (void)t;
    return 0;
  }
  // ---- page-locked arrays: chunks of the range round-robin over the streams; on each stream a chunk is
  // uploaded, decided and downloaded in order, and the three streams overlap each other's phases
  u32 k = 0;
  for (u32 a = lo; a < hi; a += chunk_requests, ++k) {
    const u32 b = std::min<u64>((u64)a + chunk_requests, hi);
    hipStream_t s = c->s[k % N_STREAMS];
    if (k < (u32)N_STREAMS && s != s0) HIPCHK(hipStreamWaitEvent(s, c->ev_setup, 0));
    if (reads_strings) HIPCHK(up2d(L.req, 4, 0, CBH_RQ_NFIELDS, a, b, s));
    else HIPCHK(up2d(L.req, 4, 0, CBH_RQ_NCORE, a, b, s));
    HIPCHK(up2d(L.ctag, 1, 0, in->n_columns, a, b, s));
    HIPCHK(up2d(L.cval, 8, 0, in->n_columns, a, b, s));
    size_t tb = 0, te = 0;
    tuples_of(a, b, tb, te);
    if (te > tb) HIPCHK(hipMemcpyAsync(base + L.act.off + tb * 4, in->tuple_action + tb, (te - tb) * 4, hipMemcpyHostToDevice, s));
    launch_check(rep, ka, d_args, a, b, sh, s);
    if (te > tb) {
      HIPCHK(hipMemcpyAsync(out->effect + tb, base + L.eff.off + tb, te - tb, hipMemcpyDeviceToHost, s));
      if (out->policy) HIPCHK(hipMemcpyAsync(out->policy + tb, base + L.pol.off + tb * 4, (te - tb) * 4, hipMemcpyDeviceToHost, s));
      if (out->scope) HIPCHK(hipMemcpyAsync(out->scope + tb, base + L.scope.off + tb * 4, (te - tb) * 4, hipMemcpyDeviceToHost, s));
      if (out->status) HIPCHK(hipMemcpyAsync(out->status + tb, base + L.status.off + tb, te - tb, hipMemcpyDeviceToHost, s));
    }
    if (out->edr_mask) HIPCHK(hipMemcpyAsync(out->edr_mask + a, base + L.edr.off + (size_t)a * 8, (size_t)(b - a) * 8, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipGetLastError());
  const double t_1 = trace_on() ? now_us() : 0;
  for (auto& s : c->s) HIPCHK(stream_wait(s));
  if (trace_on()) std::fprintf(stderr, "[cbh] range [%u,%u) dev=%d pinned chunks=%u enqueue=%.1f us wait=%.1f us\n", lo, hi, rep->device, k, t_1 - t_0, now_us() - t_1);
  return 0;
}

extern "C" int cbh_check_batch(cbh_table* t, const cbh_batch* in, const cbh_params* p, cbh_result* out) {
  if (!t || !in || !p || !out) return fail("null argument");
  if (in->n_tuples && !out->effect) return fail("cbh_result.effect is required");
  TableRef ref(t);
  BatchShape sh;
  if (validate_header(t, in) != 0) return -1;
  const Layout L = make_layout(in, t);
  const u32 NR = in->n_requests;
  if (L.in_end <= SMALL_BATCH_BYTES || NR == 0) {
    if (validate_batch(t, in, sh) != 0) return -1;
    return run_small(t, t->reps[0], in, p, out, sh, L);
  }

  // chunks of the three-stream pipeline carry at least ~32 MB of input each: a copy costs a fixed ~20 us on top of
  // its bytes, so smaller chunks lose more to that than the overlap wins (measured, profiles/r02_oneshot_probe.txt)
  const u32 chunk_env = [&] {
    const char* e = getenv("CBH_CHUNK_REQUESTS"); long v = e ? atol(e) : 0;
    if (v > 0) return (u32)((v + 63) & ~63l);
    const size_t per_request = NR ? std::max<size_t>(1, (L.in_end - L.in_begin) / NR) : 1;
    return (u32)std::min<size_t>(0xFFFFFFC0u, ((((size_t)32 << 20) / per_request) + 63) & ~(size_t)63);
  }();
  bool pinned = true;
  for (const void* q : {(const void*)in->req_u32, (const void*)in->tuple_action, (const void*)in->col_tag, (const void*)in->col_val,
                        (const void*)out->effect, (const void*)out->policy, (const void*)out->scope, (const void*)out->status, (const void*)out->edr_mask})
    pinned = pinned && is_pinned(q);
  // One device and a page-locked slab: the upload starts NOW and the O(n_requests) validation below runs while the DMA does
  // (a batch that fails it never reaches a kernel: the lease waits for the copies and hands the context back).
  CtxLease early{t->reps[0], nullptr};
  if (pinned && t->reps.size() == 1) {
    if (const uint8_t* slab = slab_base(L)) {
      HIPCHK(hipSetDevice(t->reps[0]->device));
      early.c = ctx_acquire(t->reps[0]);
      if (!early.c) return fail("could not create a launch context");
      if (ctx_reserve(early.c, 4096, L.total) != 0 || slab_upload(early.c, t->reps[0], L, slab, NR) != 0) return -1;
    }
  }
  if (validate_batch(t, in, sh) != 0) return -1;
  // contiguous request ranges over the devices (engine.go:309-338 deals inputs to workers; here a worker is a GPU)
  u32 n_dev = 1;
  if (t->reps.size() > 1 && sh.ascending) n_dev = (u32)std::min<size_t>(t->reps.size(), std::max<u32>(1, NR / SHARD_MIN_REQUESTS));
  if (n_dev == 1) return run_range(t, t->reps[0], in, p, out, sh, L, 0, NR, pinned, sh.ascending ? chunk_env : NR, early.c);
  std::vector<int> rcs(n_dev, 0);
  std::vector<std::string> errs(n_dev);
  auto work = [&](u32 i) {
    const u32 lo = (u32)(((u64)NR * i / n_dev) & ~63ull), hi = i + 1 == n_dev ? NR : (u32)(((u64)NR * (i + 1) / n_dev) & ~63ull);
    rcs[i] = run_range(t, t->reps[i], in, p, out, sh, L, lo, hi, pinned, chunk_env);
    if (rcs[i] != 0) errs[i] = g_err;
  };
  std::vector<std::thread> th;
  for (u32 i = 1; i < n_dev; ++i) th.emplace_back(work, i);
  work(0);
  for (auto& x : th) x.join();
  for (u32 i = 0; i < n_dev; ++i) if (rcs[i] != 0) return fail("device " + std::to_string(t->reps[i]->device) + ": " + errs[i]);
  return 0;
}
// The trace pass (cerbos_hip.h): the batch packed into the staging block, one copy up, the tracing kernel, the
// results and the log down.  Not a fast path - it serves the (few) inputs whose evaluation errors / outputs are wanted.
extern "C" int cbh_trace_batch(cbh_table* t, const cbh_batch* in, const cbh_params* p, cbh_result* out, cbh_trace* trace) {
  if (!t || !in || !p || !out || !trace) return fail("null argument");
  if (in->n_tuples && !out->effect) return fail("cbh_result.effect is required");
  if (trace->capacity && !trace->records) return fail("cbh_trace.records is required");
  TableRef ref(t);
  Replica* rep = t->reps[0];
  if (!rep->dev.trace_pool) return fail("the table was lowered without the trace sections");
  BatchShape sh;
  if (validate_batch(t, in, sh) != 0) return -1;
  trace->count = 0;
  if (in->n_requests == 0) return 0;
  const Layout L = make_layout(in, t);
  const size_t log_off = (L.total + 255) & ~(size_t)255;                       // {count, pad ...} then the records
  const size_t rec_off = log_off + 256, rec_bytes = (size_t)trace->capacity * CBH_TRACE_RECORD_WORDS * 4;
  const size_t total = rec_off + rec_bytes;
  HIPCHK(hipSetDevice(rep->device));
  CtxLease lease{rep, ctx_acquire(rep)};
  OneShot* c = lease.c;
  if (!c) return fail("could not create a launch context");
  if (ctx_reserve(c, total, total) != 0) return -1;
  hipStream_t s = c->s[0];
  lease.used = 1;
  uint8_t* base = c->d;
  KernelArgs ka;
  bind_args(ka, rep->dev, in, p, L, base);
  ka.o.trace_rec = (u32*)(base + rec_off); ka.o.trace_cnt = (u32*)(base + log_off); ka.o.trace_cap = trace->capacity;
  std::memcpy(c->h + L.args.off, &ka, sizeof(ka));
  for (const Seg* g : {&L.req, &L.roles, &L.act, &L.ctag, &L.cval, &L.htag, &L.hval, &L.soff, &L.sbytes, &L.sflags})
    if (g->bytes) std::memcpy(c->h + g->off, g->src, g->bytes);
  int rc = 0;
  HIPCHK(hipMemcpyAsync(c->d, c->h, L.in_end, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(base + log_off, 0, 256, s));
  launch_resolve(rep, ka, L, s, rc);
  if (rc != 0) return fail("hipMemsetAsync failed");
  const u32 grid = (in->n_requests + CBH_BLOCK - 1) / CBH_BLOCK;
  hipLaunchKernelGGL(cbh_trace_kernel, dim3(grid), dim3(CBH_BLOCK), check_lds_bytes(ka.b, rep->dev.flags), s, ka, (const KernelArgs*)(base + L.args.off));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h + L.out_begin, c->d + L.out_begin, log_off + 256 - L.out_begin, hipMemcpyDeviceToHost, s));
  HIPCHK(stream_wait(s));
  struct Dst { const Seg* g; void* dst; };
  const Dst outs[5] = {{&L.eff, out->effect}, {&L.pol, out->policy}, {&L.scope, out->scope}, {&L.status, out->status}, {&L.edr, out->edr_mask}};
  for (const Dst& o : outs) if (o.dst && o.g->bytes) std::memcpy(o.dst, c->h + o.g->off, o.g->bytes);
  std::memcpy(&trace->count, c->h + log_off, 4);
  const size_t kept = std::min<size_t>(trace->count, trace->capacity);
  if (kept) HIPCHK(hipMemcpy(trace->records, base + rec_off, kept * CBH_TRACE_RECORD_WORDS * 4, hipMemcpyDeviceToHost));
  return 0;
}
#endif  // !__HIP_DEVICE_COMPILE__
