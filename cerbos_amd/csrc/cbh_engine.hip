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
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "cbh_kernels.h"
#include "cbh_image.h"

// ======================================================================== host code
// (skipped in the device compilation pass, where the device structs carry address-space qualifiers)
#if !defined(__HIP_DEVICE_COMPILE__)

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return -1; }
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(_e)); } while (0)

struct cbh_table {
  int device = 0;
  void* image = nullptr; size_t image_len = 0; bool owns_image = true;
  std::vector<uint32_t> meta;
  TableDev dev{};
  hipStream_t stream = nullptr;
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
  // One-shot calls (cbh_check_batch) run on their own small set of contexts - stream, pinned staging
  // block, device block - so that calls from different threads overlap on the device.
  struct OneShot { hipStream_t stream = nullptr; uint8_t* h = nullptr; size_t h_cap = 0; uint8_t* d = nullptr; size_t d_cap = 0; };
  static constexpr int MAX_ONESHOT = 8;
  std::mutex ctx_mu;
  std::condition_variable ctx_cv;
  std::vector<OneShot*> ctx_idle;
  int ctx_count = 0;
};

struct cbh_device_batch {
  cbh_table* table = nullptr;
  BatchDev dev{};
  OutDev out{};
  KernelArgs* d_args = nullptr;   // device copy of the launch arguments
  KernelArgs last_args;           // what d_args currently holds
  bool have_args = false;
  u32 max_actions = 0;            // largest CBH_RQ_ACT_CNT of the batch: selects the action-mask width
  std::vector<std::pair<void*, size_t>> allocs;   // (block, capacity) taken from the table's pool
};

static int g_device = 0;
static bool g_inited = false;

extern "C" const char* cbh_last_error(void) { return g_err.c_str(); }
extern "C" uint32_t cbh_abi_version(void) { return CBH_ABI_VERSION; }

extern "C" int cbh_init(const cbh_config* cfg) {
  if (cfg && cfg->abi_version != CBH_ABI_VERSION) return fail("ABI version mismatch");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return fail("no HIP device available: the decision engine requires an MI355X (no CPU fallback)");
  g_device = cfg ? cfg->device : 0;
  if (g_device < 0 || g_device >= n) return fail("invalid device ordinal");
  HIPCHK(hipSetDevice(g_device));
  g_inited = true;
  return 0;
}
extern "C" void cbh_shutdown(void) { g_inited = false; }

static int parse_image(cbh_table* t, const uint8_t* host_copy, size_t len) {
  const char* e = cbh_parse_image(t->dev, t->meta, static_cast<const uint8_t*>(t->image), host_copy, len);
  return e ? fail(e) : 0;
}

static int table_finish(cbh_table* t) {
  HIPCHK(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
  for (auto& sl : t->ring) for (auto& e : sl.ev) HIPCHK(hipEventCreate(&e));
  return 0;
}

extern "C" int cbh_table_load(const void* blob, size_t len, cbh_table** out) {
  if (!g_inited) return fail("cbh_init has not been called");
  if (!blob || !out) return fail("null argument");
  HIPCHK(hipSetDevice(g_device));
  cbh_table* t = new (std::nothrow) cbh_table();
  if (!t) return fail("out of memory");
  t->device = g_device; t->image_len = len;
  if (hipMalloc(&t->image, len) != hipSuccess) { delete t; return fail("hipMalloc(table image) failed"); }
  if (hipMemcpy(t->image, blob, len, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(t->image); delete t; return fail("hipMemcpy(table image) failed"); }
  if (parse_image(t, static_cast<const uint8_t*>(blob), len) != 0 || table_finish(t) != 0) { (void)hipFree(t->image); delete t; return -1; }
  *out = t;
  return 0;
}

extern "C" int cbh_table_adopt_device_image(void* device_image, size_t len, cbh_table** out) {
  if (!g_inited) return fail("cbh_init has not been called");
  if (!device_image || !out) return fail("null argument");
  HIPCHK(hipSetDevice(g_device));
  std::vector<uint8_t> host(len);
  HIPCHK(hipMemcpy(host.data(), device_image, len, hipMemcpyDeviceToHost));
  cbh_table* t = new (std::nothrow) cbh_table();
  if (!t) return fail("out of memory");
  t->device = g_device; t->image = device_image; t->image_len = len; t->owns_image = false;
  if (parse_image(t, host.data(), len) != 0 || table_finish(t) != 0) { delete t; return -1; }
  *out = t;
  return 0;
}

extern "C" void cbh_table_release(cbh_table* t) {
  if (!t) return;
  (void)hipSetDevice(t->device);
  if (t->stream) { (void)hipStreamSynchronize(t->stream); (void)hipStreamDestroy(t->stream); }
  for (auto& sl : t->ring) for (auto& e : sl.ev) if (e) (void)hipEventDestroy(e);
  if (t->image && t->owns_image) (void)hipFree(t->image);
  for (auto& a : t->pool_free) (void)hipFree(a.first);
  for (auto* c : t->ctx_idle) {
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->h) (void)hipHostFree(c->h);
    if (c->d) (void)hipFree(c->d);
    delete c;
  }
  delete t;
}
extern "C" uint32_t cbh_table_num_strings(const cbh_table* t) { return t ? t->meta[CBH_M_NSTRINGS] : 0; }
extern "C" uint32_t cbh_table_num_columns(const cbh_table* t) { return t ? t->meta[CBH_M_NCOLUMNS] : 0; }
extern "C" uint64_t cbh_table_device_bytes(const cbh_table* t) { return t ? t->image_len : 0; }
extern "C" void* cbh_table_device_ptr(const cbh_table* t) { return t ? t->image : nullptr; }

// Device buffers of batches come from a per-table pool of power-of-two blocks: a small synchronous
// CheckResources round trip must not pay ~17 hipMalloc / hipFree pairs (each hipFree also
// synchronises the device).  Blocks go back to the pool on cbh_batch_release and to the driver on
// cbh_table_release.
static int pool_alloc(cbh_device_batch* b, size_t bytes, void** out) {
  size_t cap = 256;
  while (cap < bytes) cap <<= 1;
  cbh_table* t = b->table;
  {
    std::lock_guard<std::mutex> lk(t->pool_mu);
    for (size_t i = 0; i < t->pool_free.size(); ++i)
      if (t->pool_free[i].second == cap) {
        *out = t->pool_free[i].first;
        t->pool_free[i] = t->pool_free.back(); t->pool_free.pop_back();
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
  if (b->table) {
    (void)hipSetDevice(b->table->device); (void)hipStreamSynchronize(b->table->stream);
    std::lock_guard<std::mutex> lk(b->table->pool_mu);
    for (auto& a : b->allocs) b->table->pool_free.push_back(a);
  } else {
    for (auto& a : b->allocs) (void)hipFree(a.first);
  }
  delete b;
}

template <typename T>
static int up(cbh_device_batch* b, const T*& dst, const T* src, size_t n, hipStream_t s) {
  dst = nullptr;
  size_t bytes = (n ? n : 1) * sizeof(T);
  void* p = nullptr;
  if (pool_alloc(b, bytes, &p) != 0) return -1;
  if (n) {
    if (!src) return fail("cbh_batch: a required array is NULL");
    HIPCHK(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
  }
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

extern "C" int cbh_batch_upload(cbh_table* t, const cbh_batch* in, cbh_device_batch** out) {
  if (!t || !in || !out) return fail("null argument");
  if (in->n_columns != t->meta[CBH_M_NCOLUMNS]) return fail("cbh_batch.n_columns does not match the table's column schema");
  HIPCHK(hipSetDevice(t->device));
  cbh_device_batch* b = new (std::nothrow) cbh_device_batch();
  if (!b) return fail("out of memory");
  b->table = t;
  BatchDev& d = b->dev;
  d.n_requests = in->n_requests; d.n_tuples = in->n_tuples; d.n_roles = in->n_roles;
  d.n_columns = in->n_columns; d.n_strings = in->n_strings; d.heap_len = in->heap_len;
  hipStream_t s = t->stream;
  const size_t NR = in->n_requests;
  int rc = 0;
  rc |= up(b, d.req_u32, in->req_u32, (size_t)CBH_RQ_NFIELDS * NR, s);
  rc |= up(b, d.roles, in->roles, in->n_roles, s);
  rc |= up(b, d.tuple_req, in->tuple_req, in->n_tuples, s);
  rc |= up(b, d.tuple_action, in->tuple_action, in->n_tuples, s);
  rc |= up(b, d.col_tag, in->col_tag, (size_t)in->n_columns * NR, s);
  rc |= up(b, d.col_val, in->col_val, (size_t)in->n_columns * NR, s);
  rc |= up(b, d.heap_tag, in->heap_tag, in->heap_len, s);
  rc |= up(b, d.heap_val, in->heap_val, in->heap_len, s);
  rc |= up(b, d.str_off, in->str_off, (size_t)in->n_strings + 1, s);
  rc |= up(b, d.str_bytes, in->str_bytes, in->str_bytes_len, s);
  rc |= up(b, d.str_flags, in->str_flags, in->n_strings, s);
  rc |= dalloc(b, d.gbits, (size_t)3 * in->n_strings);
  rc |= dalloc(b, b->out.effect, in->n_tuples);
  rc |= dalloc(b, b->out.policy, in->n_tuples);
  rc |= dalloc(b, b->out.scope, in->n_tuples);
  rc |= dalloc(b, b->out.status, in->n_tuples);
  rc |= dalloc(b, b->out.edr, NR);
  rc |= dalloc(b, b->d_args, 1);
  for (size_t r = 0; r < NR; ++r) {
    const u32 n = in->req_u32[(size_t)CBH_RQ_ACT_CNT * NR + r];
    if (n > CBH_MAX_ACTIONS_PER_REQUEST) { cbh_batch_release(b); return fail("cbh_batch: a request carries more than CBH_MAX_ACTIONS_PER_REQUEST actions"); }
    if (n > b->max_actions) b->max_actions = n;
  }
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

static void collect_slot(cbh_table* t, cbh_table::Slot& sl) {   // the slot's last event has completed
  if (!sl.pending) return;
  float a = 0, c = 0;
  if (sl.resolved && hipEventElapsedTime(&a, sl.ev[0], sl.ev[1]) != hipSuccess) a = 0;
  if (hipEventElapsedTime(&c, sl.ev[2], sl.ev[3]) == hipSuccess) {
    t->resolve_ms_sum += a; t->check_ms_sum += c; t->timed += 1;
  }
  sl.pending = false;
}
static void collect_times(cbh_table* t) {   // after the stream has been synchronised
  for (auto& sl : t->ring) collect_slot(t, sl);
}

extern "C" int cbh_check_resident(cbh_table* t, cbh_device_batch* b, const cbh_params* p) {
  if (!t || !b || !p) return fail("null argument");
  if (b->table != t) return fail("batch was uploaded for a different table");
  std::lock_guard<std::mutex> lk(t->mu);
  HIPCHK(hipSetDevice(t->device));
  hipStream_t s = t->stream;
  // Kernel durations come from the dispatches' own begin / end timestamps (hipExtLaunchKernelGGL
  // with start / stop events: what rocprofv3's kernel trace reads too), not from event-record
  // packets placed around them, which would sit between back-to-back launches and add their own
  // latency to the figure.
  // Every fourth launch is timed (and the first few, so that a short run has a figure): a
  // timestamped dispatch costs the queue a little more than a plain one.
  const uint64_t launch_no = t->launches++;
  const bool timed = launch_no < 4 || (launch_no & 3) == 0;
  cbh_table::Slot scratch_slot;
  cbh_table::Slot& sl = timed ? t->ring[t->next_slot++ % cbh_table::RING] : scratch_slot;
  if (timed && sl.pending) { HIPCHK(hipEventSynchronize(sl.ev[3])); collect_slot(t, sl); }
  const BatchDev& d = b->dev;
  {
    // launch arguments live in device memory; re-sent only when they change (the kernel itself
    // writes every output word of every request, so nothing needs clearing between launches)
    KernelArgs ka;
    std::memset(&ka, 0, sizeof(ka));
    ka.t = t->dev; ka.b = d; ka.o = b->out; ka.now_ns = p->now_ns; ka.flags = p->flags;
    if (!b->have_args || std::memcmp(&ka, &b->last_args, sizeof(ka)) != 0) {
      b->last_args = ka; b->have_args = true;
      HIPCHK(hipMemcpyAsync(b->d_args, &b->last_args, sizeof(ka), hipMemcpyHostToDevice, s));
    }
  }
  // batch-local strings against the table's glob automata; a table without globs has nothing to
  // resolve (the bits were zeroed once at upload)
  const u32 maxw = std::max(std::max(t->dev.nfa_words[0], t->dev.nfa_words[1]), t->dev.nfa_words[2]);
  sl.resolved = d.n_strings && maxw;
  if (sl.resolved) {
    const u32 grid = (d.n_strings + CBH_BLOCK - 1) / CBH_BLOCK;
    const size_t lds = (size_t)(2 + 512) * maxw * sizeof(u64);
    if (timed) hipExtLaunchKernelGGL(cbh_resolve_globs_kernel, dim3(grid), dim3(CBH_BLOCK), lds, s, sl.ev[0], sl.ev[1], 0, t->dev, d);
    else hipLaunchKernelGGL(cbh_resolve_globs_kernel, dim3(grid), dim3(CBH_BLOCK), lds, s, t->dev, d);
  }
  sl.pending = false;
  if (d.n_requests) {
    const u32 grid = (d.n_requests + CBH_BLOCK - 1) / CBH_BLOCK;   // one lane per request
    const u32 ncc = d.n_columns < CBH_CACHE_COLS ? d.n_columns : CBH_CACHE_COLS;
    const size_t dyn_lds = (size_t)ncc * CBH_BLOCK * 12;   // column cache: value low / high / tag dword per lane
    const cbh_check_kernel_fn kernel = cbh_pick_check_kernel(t->dev.flags, t->dev.n_dr, maxw != 0 || (t->dev.flags & CBH_MF_HAS_ANY_PATTERN), b->max_actions);
    if (timed) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(CBH_BLOCK), dyn_lds, s, sl.ev[2], sl.ev[3], 0, b->last_args, (const KernelArgs*)b->d_args);
    else hipLaunchKernelGGL(kernel, dim3(grid), dim3(CBH_BLOCK), dyn_lds, s, b->last_args, (const KernelArgs*)b->d_args);
    sl.pending = timed;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int cbh_synchronize(cbh_table* t) {
  if (!t) return fail("null argument");
  std::lock_guard<std::mutex> lk(t->mu);
  HIPCHK(hipSetDevice(t->device));
  HIPCHK(hipStreamSynchronize(t->stream));
  collect_times(t);
  return 0;
}

extern "C" int cbh_kernel_time_ms(cbh_table* t, float* check_ms, float* resolve_ms) {
  if (!t) return fail("null argument");
  std::lock_guard<std::mutex> lk(t->mu);
  if (t->timed == 0) return fail("no timed launches yet");
  if (check_ms) *check_ms = (float)(t->check_ms_sum / (double)t->timed);
  if (resolve_ms) *resolve_ms = (float)(t->resolve_ms_sum / (double)t->timed);
  t->check_ms_sum = t->resolve_ms_sum = 0; t->timed = 0;
  return 0;
}

extern "C" int cbh_result_download(cbh_table* t, cbh_device_batch* b, cbh_result* out) {
  if (!t || !b || !out) return fail("null argument");
  if (b->dev.n_tuples && !out->effect) return fail("cbh_result.effect is required");
  std::lock_guard<std::mutex> lk(t->mu);
  HIPCHK(hipSetDevice(t->device));
  hipStream_t s = t->stream;
  const BatchDev& d = b->dev;
  if (d.n_tuples) HIPCHK(hipMemcpyAsync(out->effect, b->out.effect, d.n_tuples, hipMemcpyDeviceToHost, s));
  if (out->policy && d.n_tuples) HIPCHK(hipMemcpyAsync(out->policy, b->out.policy, (size_t)d.n_tuples * 4, hipMemcpyDeviceToHost, s));
  if (out->scope && d.n_tuples) HIPCHK(hipMemcpyAsync(out->scope, b->out.scope, (size_t)d.n_tuples * 4, hipMemcpyDeviceToHost, s));
  if (out->status && d.n_tuples) HIPCHK(hipMemcpyAsync(out->status, b->out.status, d.n_tuples, hipMemcpyDeviceToHost, s));
  if (out->edr_mask && d.n_requests) HIPCHK(hipMemcpyAsync(out->edr_mask, b->out.edr, (size_t)d.n_requests * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  collect_times(t);
  return 0;
}

// ---- one-shot path: CheckResources round trip for a host batch ------------------------------------------
// All arrays of the batch go into ONE device block.  A small batch (the latency case) is packed into
// a pinned staging block and crosses PCIe in one copy each way; a large one copies array by array
// straight from / to the caller's memory (the driver pins those pages on the fly, which beats a host
// memcpy into staging).  Each call owns a context (stream + blocks), so concurrent callers overlap.
static cbh_table::OneShot* ctx_acquire(cbh_table* t) {
  std::unique_lock<std::mutex> lk(t->ctx_mu);
  for (;;) {
    if (!t->ctx_idle.empty()) { auto* c = t->ctx_idle.back(); t->ctx_idle.pop_back(); return c; }
    if (t->ctx_count < cbh_table::MAX_ONESHOT) {
      ++t->ctx_count;
      lk.unlock();
      auto* c = new (std::nothrow) cbh_table::OneShot();
      if (c && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; c = nullptr; }
      if (!c) { lk.lock(); --t->ctx_count; t->ctx_cv.notify_one(); }
      return c;
    }
    t->ctx_cv.wait(lk);
  }
}
struct CtxLease {
  cbh_table* t; cbh_table::OneShot* c;
  ~CtxLease() {
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);   // an error return must not leave copies from caller memory in flight
    { std::lock_guard<std::mutex> lk(t->ctx_mu); t->ctx_idle.push_back(c); }
    t->ctx_cv.notify_one();
  }
};
static int ctx_reserve(cbh_table::OneShot* c, size_t hbytes, size_t dbytes) {
  if (hbytes > c->h_cap) {
    if (c->h) { (void)hipHostFree(c->h); c->h = nullptr; c->h_cap = 0; }
    size_t cap = 1 << 16; while (cap < hbytes) cap <<= 1;
    HIPCHK(hipHostMalloc((void**)&c->h, cap, hipHostMallocDefault));
    c->h_cap = cap;
  }
  if (dbytes > c->d_cap) {
    if (c->d) { (void)hipFree(c->d); c->d = nullptr; c->d_cap = 0; }
    size_t cap = 1 << 16; while (cap < dbytes) cap <<= 1;
    HIPCHK(hipMalloc((void**)&c->d, cap));
    c->d_cap = cap;
  }
  return 0;
}

extern "C" int cbh_check_batch(cbh_table* t, const cbh_batch* in, const cbh_params* p, cbh_result* out) {
  if (!t || !in || !p || !out) return fail("null argument");
  if (in->n_columns != t->meta[CBH_M_NCOLUMNS]) return fail("cbh_batch.n_columns does not match the table's column schema");
  if (in->n_tuples && !out->effect) return fail("cbh_result.effect is required");
  const size_t NR = in->n_requests, NT = in->n_tuples, NS = in->n_strings;
  u32 max_actions = 0;
  if (NR && !in->req_u32) return fail("cbh_batch: a required array is NULL");
  for (size_t r = 0; r < NR; ++r) {
    const u32 n = in->req_u32[(size_t)CBH_RQ_ACT_CNT * NR + r];
    if (n > CBH_MAX_ACTIONS_PER_REQUEST) return fail("cbh_batch: a request carries more than CBH_MAX_ACTIONS_PER_REQUEST actions");
    if (n > max_actions) max_actions = n;
  }
  // layout of the device block: [launch arguments | inputs ... | glob bits | outputs ...], 256-byte aligned pieces
  struct Seg { size_t off, bytes; const void* src; };
  size_t cur = 0;
  auto seg = [&](const void* src, size_t bytes) { Seg g{cur, bytes, src}; cur += (bytes + 255) & ~(size_t)255; return g; };
  const Seg s_args = seg(nullptr, sizeof(KernelArgs));
  const Seg ins[11] = {
    seg(in->req_u32, (size_t)CBH_RQ_NFIELDS * NR * 4), seg(in->roles, (size_t)in->n_roles * 4), seg(in->tuple_req, NT * 4),
    seg(in->tuple_action, NT * 4), seg(in->col_tag, (size_t)in->n_columns * NR), seg(in->col_val, (size_t)in->n_columns * NR * 8),
    seg(in->heap_tag, in->heap_len), seg(in->heap_val, (size_t)in->heap_len * 8), seg(in->str_off, (NS + 1) * 4),
    seg(in->str_bytes, in->str_bytes_len), seg(in->str_flags, NS)};
  for (const Seg& g : ins) if (g.bytes && !g.src) return fail("cbh_batch: a required array is NULL");
  const size_t in_end = cur;
  const Seg s_gbits = seg(nullptr, 3 * NS * 8);
  const size_t out_begin = cur;
  const Seg s_eff = seg(nullptr, NT), s_pol = seg(nullptr, NT * 4), s_scope = seg(nullptr, NT * 4), s_status = seg(nullptr, NT),
            s_edr = seg(nullptr, NR * 8);
  const size_t total = cur;
  const bool staged = in_end <= ((size_t)4 << 20);

  HIPCHK(hipSetDevice(t->device));
  CtxLease lease{t, ctx_acquire(t)};
  cbh_table::OneShot* c = lease.c;
  if (!c) return fail("could not create a launch context");
  if (ctx_reserve(c, staged ? total : 256, total) != 0) return -1;
  hipStream_t s = c->stream;

  KernelArgs ka;
  std::memset(&ka, 0, sizeof(ka));
  ka.t = t->dev; ka.now_ns = p->now_ns; ka.flags = p->flags;
  BatchDev& d = ka.b;
  d.n_requests = in->n_requests; d.n_tuples = in->n_tuples; d.n_roles = in->n_roles;
  d.n_columns = in->n_columns; d.n_strings = in->n_strings; d.heap_len = in->heap_len;
  uint8_t* base = c->d;
  d.req_u32 = (const u32*)(base + ins[0].off); d.roles = (const u32*)(base + ins[1].off); d.tuple_req = (const u32*)(base + ins[2].off);
  d.tuple_action = (const u32*)(base + ins[3].off); d.col_tag = base + ins[4].off; d.col_val = (const u64*)(base + ins[5].off);
  d.heap_tag = base + ins[6].off; d.heap_val = (const u64*)(base + ins[7].off); d.str_off = (const u32*)(base + ins[8].off);
  d.str_bytes = base + ins[9].off; d.str_flags = base + ins[10].off; d.gbits = (u64*)(base + s_gbits.off);
  ka.o.effect = base + s_eff.off; ka.o.policy = (u32*)(base + s_pol.off); ka.o.scope = (u32*)(base + s_scope.off);
  ka.o.status = base + s_status.off; ka.o.edr = (u64*)(base + s_edr.off);

  std::memcpy(c->h + s_args.off, &ka, sizeof(ka));
  if (staged) {
    for (const Seg& g : ins) if (g.bytes) std::memcpy(c->h + g.off, g.src, g.bytes);
    HIPCHK(hipMemcpyAsync(base, c->h, in_end, hipMemcpyHostToDevice, s));
  } else {
    HIPCHK(hipMemcpyAsync(base, c->h, sizeof(ka), hipMemcpyHostToDevice, s));
    for (const Seg& g : ins) if (g.bytes) HIPCHK(hipMemcpyAsync(base + g.off, g.src, g.bytes, hipMemcpyHostToDevice, s));
  }
  const u32 maxw = std::max(std::max(t->dev.nfa_words[0], t->dev.nfa_words[1]), t->dev.nfa_words[2]);
  if (NS) {
    if (maxw) {
      const u32 grid = (d.n_strings + CBH_BLOCK - 1) / CBH_BLOCK;
      const size_t lds = (size_t)(2 + 512) * maxw * sizeof(u64);
      hipLaunchKernelGGL(cbh_resolve_globs_kernel, dim3(grid), dim3(CBH_BLOCK), lds, s, t->dev, d);
    } else {
      HIPCHK(hipMemsetAsync(d.gbits, 0, s_gbits.bytes, s));   // no automata: no string matches a glob
    }
  }
  if (NR) {
    const u32 grid = (d.n_requests + CBH_BLOCK - 1) / CBH_BLOCK;   // one lane per request
    const u32 ncc = d.n_columns < CBH_CACHE_COLS ? d.n_columns : CBH_CACHE_COLS;
    const size_t dyn_lds = (size_t)ncc * CBH_BLOCK * 12;
    const cbh_check_kernel_fn kernel = cbh_pick_check_kernel(t->dev.flags, t->dev.n_dr, maxw != 0 || (t->dev.flags & CBH_MF_HAS_ANY_PATTERN), max_actions);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(CBH_BLOCK), dyn_lds, s, ka, (const KernelArgs*)(base + s_args.off));
  }
  HIPCHK(hipGetLastError());
  struct Dst { const Seg* g; void* dst; };
  const Dst outs[5] = {{&s_eff, out->effect}, {&s_pol, out->policy}, {&s_scope, out->scope}, {&s_status, out->status}, {&s_edr, out->edr_mask}};
  if (staged) {
    if (total > out_begin) HIPCHK(hipMemcpyAsync(c->h + out_begin, base + out_begin, total - out_begin, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (const Dst& o : outs) if (o.dst && o.g->bytes) std::memcpy(o.dst, c->h + o.g->off, o.g->bytes);
  } else {
    for (const Dst& o : outs) if (o.dst && o.g->bytes) HIPCHK(hipMemcpyAsync(o.dst, base + o.g->off, o.g->bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  return 0;
}
#endif  // !__HIP_DEVICE_COMPILE__
