// TEST INFRASTRUCTURE ONLY - never part of the product, never on the product's include path.
//
// A stand-in for <hip/hip_runtime.h> that lets cerbos_amd/csrc/cbh_engine.hip - the library's HOST side: pools, streams, slices,
// the launches of every entry point - be compiled as plain C++ and run without a GPU (tests/sim_engine.py build() ->
// tests/hostsim/_build/libcerbos_hip_sim.so).  Device memory is host memory (filled with a pattern: nothing may rely on zeroes), copies
// are memcpy, streams and events are tokens, and a kernel launch runs the kernel's source on the fiber scheduler of
// tests/hostsim/hostsim.cpp (one workgroup = its lanes as ucontext fibers, wave primitives and __syncthreads() as rendezvous),
// one launch at a time.  The CPU tier loads it INSTEAD of libcerbos_hip.so only where a test says so (tests/sim_engine.py); it proves
// nothing about speed and is not a fallback - cerbos_amd.capi never looks for it.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#define CBH_HOSTSIM 1
#define CBH_HOSTSIM_ENGINE 1
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
struct uint4 { uint32_t x, y, z, w; };
struct dim3 { uint32_t x, y, z; dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {} };

namespace hs {
static const int MAX_BLOCK = 256, WAVE = 64;
struct Fiber {
  ucontext_t ctx;
  std::vector<char> stack;
  dim3 tid, bid;
  bool done = false;
  int waiting = 0;      // 0 running, 1 wave rendezvous, 2 block barrier
  uint64_t xchg = 0;
  uint32_t arg = 0;
  int op = 0;
};
inline Fiber* fibers() { static Fiber f[MAX_BLOCK]; return f; }
inline int& cur() { static int c = 0; return c; }
inline ucontext_t& sched() { static ucontext_t s; return s; }
inline dim3& block_dim() { static dim3 d; return d; }
inline std::function<void()>& body() { static std::function<void()> b; return b; }
inline std::recursive_mutex& mu() { static std::recursive_mutex m; return m; }
inline uint64_t& launches() { static uint64_t n = 0; return n; }
enum { OP_BALLOT = 1, OP_READLANE = 2 };
inline void yield(int why) { Fiber& f = fibers()[cur()]; f.waiting = why; swapcontext(&f.ctx, &sched()); }
inline void fiber_main() {
  body()();
  Fiber& f = fibers()[cur()];
  f.done = true; f.waiting = 0;
  swapcontext(&f.ctx, &sched());
}
inline bool resolve_wave(int w, int block) {
  Fiber* g = fibers();
  const int base = w * WAVE, top = std::min(block, base + WAVE);
  int first = -1, op = 0;
  for (int l = base; l < top; ++l) {
    Fiber& f = g[l];
    if (f.done) continue;
    if (f.waiting != 1) return false;
    if (first < 0) { first = l; op = f.op; }
    else if (f.op != op) { std::fprintf(stderr, "engine sim: lanes of a wave diverged across different cross-lane ops\n"); std::abort(); }
  }
  if (first < 0) return false;
  if (op == OP_BALLOT) {
    uint64_t m = 0;
    for (int l = base; l < top; ++l) if (!g[l].done && g[l].xchg) m |= 1ull << (l - base);
    for (int l = base; l < top; ++l) if (!g[l].done) { g[l].xchg = m; g[l].waiting = 0; }
  } else {
    const uint32_t lane = g[first].arg;
    for (int l = base; l < top; ++l)
      if (!g[l].done && g[l].arg != lane) { std::fprintf(stderr, "engine sim: readlane index is not wave-uniform\n"); std::abort(); }
    if (g[base + lane].done) { std::fprintf(stderr, "engine sim: readlane from an exited lane\n"); std::abort(); }
    const uint64_t v = g[base + lane].xchg;
    for (int l = base; l < top; ++l) if (!g[l].done) { g[l].xchg = v; g[l].waiting = 0; }
  }
  return true;
}
inline void run_block(uint32_t blk, int block) {
  Fiber* g = fibers();
  for (int i = 0; i < block; ++i) {
    Fiber& f = g[i];
    if (f.stack.empty()) f.stack.resize(256 * 1024);
    f.done = false; f.waiting = 0; f.tid = dim3(i, 0, 0); f.bid = dim3(blk, 0, 0);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.data();
    f.ctx.uc_stack.ss_size = f.stack.size();
    f.ctx.uc_link = &sched();
    makecontext(&f.ctx, fiber_main, 0);
  }
  for (;;) {
    bool progressed = false, all_done = true;
    for (int i = 0; i < block; ++i) {
      Fiber& f = g[i];
      if (f.done) continue;
      all_done = false;
      if (f.waiting == 0) { cur() = i; swapcontext(&sched(), &f.ctx); progressed = true; }
    }
    if (all_done) return;
    for (int w = 0; w < (block + WAVE - 1) / WAVE; ++w) progressed |= resolve_wave(w, block);
    bool all_bar = true; int n = 0;
    for (int i = 0; i < block; ++i) if (!g[i].done) { ++n; all_bar &= g[i].waiting == 2; }
    if (n && all_bar) { for (int i = 0; i < block; ++i) g[i].waiting = 0; progressed = true; }
    if (!progressed) { std::fprintf(stderr, "engine sim: deadlock (lanes wait at different sync points)\n"); std::abort(); }
  }
}
// one launch at a time, whichever host thread asks (the sliced calls launch from several)
template <class F>
inline void launch(dim3 grid, dim3 block, F&& f) {
  std::lock_guard<std::recursive_mutex> lk(mu());
  if (block.x > (uint32_t)MAX_BLOCK) { std::fprintf(stderr, "engine sim: workgroup of %u lanes\n", block.x); std::abort(); }
  body() = std::function<void()>(f);
  block_dim() = block;
  ++launches();
  for (uint32_t b = 0; b < grid.x; ++b) run_block(b, (int)block.x);
  body() = nullptr;
}
}  // namespace hs

#define threadIdx (hs::fibers()[hs::cur()].tid)
#define blockIdx (hs::fibers()[hs::cur()].bid)
#define blockDim (hs::block_dim())

static inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; std::memcpy(&v, &d, 8); return v; }
static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p |= v; return o; }
using std::trunc;
static inline uint64_t wave_ballot(bool p) {
  hs::Fiber& f = hs::fibers()[hs::cur()];
  f.op = hs::OP_BALLOT; f.xchg = p ? 1 : 0;
  hs::yield(1);
  return hs::fibers()[hs::cur()].xchg;
}
static inline uint32_t wave_readlane(uint32_t v, uint32_t lane) {
  hs::Fiber& f = hs::fibers()[hs::cur()];
  f.op = hs::OP_READLANE; f.xchg = v; f.arg = lane;
  hs::yield(1);
  return (uint32_t)hs::fibers()[hs::cur()].xchg;
}
static inline void __syncthreads() { hs::yield(2); }

// ---- the runtime's host API, as far as cbh_engine.hip uses it
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600, hipErrorInvalidValue = 1 };
struct hs_stream { int id; };
struct hs_event { int id; };
typedef hs_stream* hipStream_t;
typedef hs_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2 };
enum { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { int type; int device; void* devicePointer; void* hostPointer; };

static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "engine sim"; }
// CBH_SIM_DEVICES=<n>: the simulated node has n "devices" (they share the one memory; what is exercised is the library's
// bookkeeping per replica - broadcast of the image by peer copies, request ranges per device, streams and pools per replica)
static inline int hs_devices() { static const int n = [] { const char* e = getenv("CBH_SIM_DEVICES"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }(); return n; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = hs_devices(); return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d >= 0 && d < hs_devices() ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = hs_devices() > 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
// fault injection (tests/test_sim_engine.py): after `n` more successful allocations every further one fails, until the budget is reset
// (-1 = never).  What is under test is the library's way out: an error code and its message, nothing leaked into the next call, no
// thread left waiting for a slice that gave up.
inline long& hs_alloc_budget() { static long b = -1; return b; }
extern "C" __attribute__((visibility("default"))) void cbh_sim_set_alloc_budget(long n) { std::lock_guard<std::recursive_mutex> lk(hs::mu()); hs_alloc_budget() = n; }
static inline hipError_t hs_alloc(void** p, size_t n) {
  {
    std::lock_guard<std::recursive_mutex> lk(hs::mu());
    long& b = hs_alloc_budget();
    if (b == 0) { *p = nullptr; return hipErrorInvalidValue; }
    if (b > 0) --b;
  }
  const size_t cap = (n + 255) & ~(size_t)255;
  void* q = nullptr;
  if (posix_memalign(&q, 256, cap ? cap : 256) != 0) return hipErrorInvalidValue;
  static const int fill = [] { const char* e = getenv("CBH_SIM_FILL"); return e ? (int)strtol(e, nullptr, 0) & 0xFF : 0xA5; }();
  std::memset(q, fill, cap ? cap : 256);   // device memory is not zeroed (CBH_SIM_FILL=0x00 / 0xFF: other garbage, same answers expected)
  *p = q;
  return hipSuccess;
}
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hs_alloc(reinterpret_cast<void**>(p), n); }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
// page-locked blocks are remembered: hipPointerGetAttributes tells them from pageable memory, as the library's one-shot path asks
inline std::vector<std::pair<const char*, size_t>>& hs_pinned() { static std::vector<std::pair<const char*, size_t>> v; return v; }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) {
  const hipError_t e = hs_alloc(reinterpret_cast<void**>(p), n);
  if (e == hipSuccess) { std::lock_guard<std::recursive_mutex> lk(hs::mu()); hs_pinned().emplace_back(reinterpret_cast<const char*>(*p), n ? n : 1); }
  return e;
}
static inline hipError_t hipHostFree(void* p) {
  { std::lock_guard<std::recursive_mutex> lk(hs::mu()); auto& v = hs_pinned(); for (size_t i = 0; i < v.size(); ++i) if (v[i].first == p) { v[i] = v.back(); v.pop_back(); break; } }
  std::free(p);
  return hipSuccess;
}
template <class T> static inline hipError_t hipHostGetDevicePointer(T** d, void* h, unsigned) { *d = static_cast<T*>(h); return hipSuccess; }
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  std::lock_guard<std::recursive_mutex> lk(hs::mu());
  for (auto& b : hs_pinned()) if (static_cast<const char*>(p) >= b.first && static_cast<const char*>(p) < b.first + b.second) { a->type = hipMemoryTypeHost; return hipSuccess; }
  a->type = 0;
  return hipErrorInvalidValue;   // pageable memory: the runtime does not know the pointer
}
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
// ... and the same for asynchronous copies (a failed copy moves nothing)
inline long& hs_copy_budget() { static long b = -1; return b; }
extern "C" __attribute__((visibility("default"))) void cbh_sim_set_copy_budget(long n) { std::lock_guard<std::recursive_mutex> lk(hs::mu()); hs_copy_budget() = n; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) {
  std::lock_guard<std::recursive_mutex> lk(hs::mu());   // (kernels run under the same lock: a copy never lands in the middle of one)
  long& b = hs_copy_budget();
  if (b == 0) return hipErrorInvalidValue;
  if (b > 0) --b;
  if (n) std::memmove(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpyPeer(void* d, int, const void* s, int, size_t n) { if (n) std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t = nullptr) {
  std::lock_guard<std::recursive_mutex> lk(hs::mu());
  for (size_t r = 0; r < height; ++r) std::memmove(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) {
  std::lock_guard<std::recursive_mutex> lk(hs::mu());
  if (n) std::memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new hs_stream{0}; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = new hs_stream{0}; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hs_event{0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hs_event{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
// workgroups of a kernel a CU holds at once: by LDS alone here (160 KB a CU, at most eight) - enough to exercise both forms of the
// column cache's tags (cbh_engine.hip packed_tags_pay)
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t lds) {
  *n = (int)std::min<size_t>(8, lds ? (160u * 1024u) / lds : 8);
  return hipSuccess;
}

// The kernels' dynamic LDS is one static array here (cbh_check_wave.h cbh_dyn_lds), larger than any launch asks for; the bytes BEHIND
// what a launch asked for are painted before it and looked at after it: a kernel that writes beyond the size the host computed - on
// the GPU such a write is dropped, or lands in a neighbour's LDS - aborts the test.
namespace hs {
template <class F>
inline void launch_checked(dim3 grid, dim3 block, size_t lds, unsigned char* dyn, size_t dyn_bytes, const char* name, F&& f) {
  std::lock_guard<std::recursive_mutex> lk(mu());
  if (lds > 160u * 1024u) { std::fprintf(stderr, "engine sim: %s asks for %zu bytes of dynamic LDS: more than a CU has\n", name, lds); std::abort(); }
  if (getenv("CBH_SIM_LDS_REPORT")) {   // the largest request per kernel symbol, printed when the process ends
    static std::vector<std::pair<std::string, size_t>> seen;
    static const int once = std::atexit([] { for (auto& e : seen) std::fprintf(stderr, "engine sim: dynamic LDS of %s: up to %zu bytes\n", e.first.c_str(), e.second); });
    (void)once;
    bool found = false;
    for (auto& e : seen) if (e.first == name) { e.second = std::max(e.second, lds); found = true; }
    if (!found) seen.emplace_back(name, lds);
  }
  if (lds > dyn_bytes) { std::fprintf(stderr, "engine sim: %s asks for %zu bytes of dynamic LDS, the simulation holds %zu\n", name, lds, dyn_bytes); std::abort(); }
  std::memset(dyn + lds, 0xC3, dyn_bytes - lds);
  launch(grid, block, f);
  for (size_t i = lds; i < dyn_bytes; ++i)
    if (dyn[i] != 0xC3) { std::fprintf(stderr, "engine sim: %s wrote dynamic LDS at byte %zu, beyond the %zu its launch asked for\n", name, i, lds); std::abort(); }
}
}  // namespace hs
#define hipLaunchKernelGGL(fn, grid, block, lds, stream, ...) hs::launch_checked((grid), (block), (size_t)(lds), cbh_dyn_lds, sizeof(cbh_dyn_lds), #fn, [=]() { fn(__VA_ARGS__); })
#define hipExtLaunchKernelGGL(fn, grid, block, lds, stream, ev0, ev1, flags, ...) hs::launch_checked((grid), (block), (size_t)(lds), cbh_dyn_lds, sizeof(cbh_dyn_lds), #fn, [=]() { fn(__VA_ARGS__); })
