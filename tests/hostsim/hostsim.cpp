// TEST INFRASTRUCTURE ONLY - never loaded by the product (cerbos_amd.capi loads
// libcerbos_hip.so and nothing else).
//
// Compiles the *device* source of the decision kernel (cerbos_amd/csrc/cbh_kernels.h) as
// plain host C++ by shimming the HIP keywords.  One workgroup = 256 fibers (ucontext) on one
// OS thread; the wave-level primitives the kernel uses (ballot / readlane) and
// __syncthreads() are rendezvous points between the fibers of a wave / block, so a
// wave-cooperative kernel runs with exactly the cross-lane semantics it has on the GPU as
// long as every lane of a wave reaches every cross-lane call (the kernel's own discipline).
// This lets the CPU-only test tier (-m "not gpu") exercise lowering + bytecode + kernel logic
// against the oracle and the reference's golden cases.  It proves nothing about performance
// and is not a fallback: the GPU tier re-runs the same cases through the real library.
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CBH_HOSTSIM 1
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
struct uint4 { uint32_t x, y, z, w; };
struct Dim3 { uint32_t x = 0, y = 0, z = 0; };

static const int HS_BLOCK = 64, HS_WAVE = 64;   // = CBH_BLOCK (checked below)
struct Fiber {
  ucontext_t ctx;
  std::vector<char> stack;
  Dim3 tid, bid;
  bool done = false;
  int waiting = 0;      // 0 running, 1 wave rendezvous, 2 block barrier
  uint64_t xchg = 0;    // value contributed to / received from a rendezvous
  uint32_t arg = 0;
  int op = 0;
};
static Fiber g_fibers[HS_BLOCK];
static int g_cur = 0;
static ucontext_t g_sched;
#define threadIdx (g_fibers[g_cur].tid)
#define blockIdx (g_fibers[g_cur].bid)

static inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; std::memcpy(&v, &d, 8); return v; }
static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p |= v; return o; }
using std::trunc;

static void hs_yield(int why) {
  g_fibers[g_cur].waiting = why;
  swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}
enum { HS_OP_BALLOT = 1, HS_OP_READLANE = 2 };
// wave primitives used by the kernel (see cbh_wave.h for the device versions)
static inline uint64_t wave_ballot(bool p) {
  Fiber& f = g_fibers[g_cur];
  f.op = HS_OP_BALLOT; f.xchg = p ? 1 : 0;
  hs_yield(1);
  return g_fibers[g_cur].xchg;
}
static inline uint32_t wave_readlane(uint32_t v, uint32_t lane) {
  Fiber& f = g_fibers[g_cur];
  f.op = HS_OP_READLANE; f.xchg = v; f.arg = lane;
  hs_yield(1);
  return (uint32_t)g_fibers[g_cur].xchg;
}
static inline void __syncthreads() { hs_yield(2); }

#include "../../cerbos_amd/csrc/cbh_kernels.h"
#include "../../cerbos_amd/csrc/cbh_image.h"

static_assert(HS_BLOCK == CBH_BLOCK, "hostsim block size must match the kernels'");
static thread_local std::string g_err;
extern "C" const char* hostsim_last_error() { return g_err.c_str(); }

static KernelArgs* g_args;

static uint32_t g_max_actions, g_max_roles; static bool g_plain;   // same kernel selection as cbh_check_resident (cbh_engine.hip)
static cbh_check_kernel_fn g_kernel;   // the kernel the fibers run

static int g_last_kind = -1;   // which kernel family decided the last batch (hostsim_last_kind: tests assert the one they mean to exercise)
static bool g_trace;   // hostsim_trace: the trace pass's kernel (cbh_trace_batch)

static void fiber_main() {
  if (g_trace) cbh_trace_kernel(*g_args, g_args);
  else g_kernel(*g_args, g_args);
  g_fibers[g_cur].done = true;
  g_fibers[g_cur].waiting = 0;
  swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}

static bool resolve_wave(int w) {
  // all unfinished lanes of wave w are waiting at a wave rendezvous -> complete it
  int base = w * HS_WAVE, first = -1, op = 0;
  for (int l = 0; l < HS_WAVE; ++l) {
    Fiber& f = g_fibers[base + l];
    if (f.done) continue;
    if (f.waiting != 1) return false;
    if (first < 0) { first = l; op = f.op; }
    else if (f.op != op) { std::fprintf(stderr, "hostsim: lanes of a wave diverged across different cross-lane ops\n"); std::abort(); }
  }
  if (first < 0) return false;
  if (op == HS_OP_BALLOT) {
    uint64_t m = 0;
    for (int l = 0; l < HS_WAVE; ++l) if (!g_fibers[base + l].done && g_fibers[base + l].xchg) m |= 1ull << l;
    for (int l = 0; l < HS_WAVE; ++l) if (!g_fibers[base + l].done) { g_fibers[base + l].xchg = m; g_fibers[base + l].waiting = 0; }
  } else {
    uint32_t lane = g_fibers[base + first].arg;
    for (int l = 0; l < HS_WAVE; ++l)
      if (!g_fibers[base + l].done && g_fibers[base + l].arg != lane) { std::fprintf(stderr, "hostsim: readlane index is not wave-uniform\n"); std::abort(); }
    if (g_fibers[base + lane].done) { std::fprintf(stderr, "hostsim: readlane from an exited lane\n"); std::abort(); }
    uint64_t v = g_fibers[base + lane].xchg;
    for (int l = 0; l < HS_WAVE; ++l) if (!g_fibers[base + l].done) { g_fibers[base + l].xchg = v; g_fibers[base + l].waiting = 0; }
  }
  return true;
}

static void run_block(uint32_t blk) {
  for (int i = 0; i < HS_BLOCK; ++i) {
    Fiber& f = g_fibers[i];
    if (f.stack.empty()) f.stack.resize(256 * 1024);
    f.done = false; f.waiting = 0; f.tid.x = i; f.bid.x = blk;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.data();
    f.ctx.uc_stack.ss_size = f.stack.size();
    f.ctx.uc_link = &g_sched;
    makecontext(&f.ctx, fiber_main, 0);
  }
  for (;;) {
    bool progressed = false, all_done = true;
    for (int i = 0; i < HS_BLOCK; ++i) {
      Fiber& f = g_fibers[i];
      if (f.done) continue;
      all_done = false;
      if (f.waiting == 0) { g_cur = i; swapcontext(&g_sched, &f.ctx); progressed = true; }
    }
    if (all_done) return;
    for (int w = 0; w < HS_BLOCK / HS_WAVE; ++w) progressed |= resolve_wave(w);
    // block barrier: every unfinished fiber waits at __syncthreads
    bool all_bar = true; int n = 0;
    for (int i = 0; i < HS_BLOCK; ++i) if (!g_fibers[i].done) { ++n; all_bar &= g_fibers[i].waiting == 2; }
    if (n && all_bar) { for (int i = 0; i < HS_BLOCK; ++i) g_fibers[i].waiting = 0; progressed = true; }
    if (!progressed) { std::fprintf(stderr, "hostsim: deadlock (lanes wait at different sync points)\n"); std::abort(); }
  }
}

// gbits: [3][n_strings] glob match bits of the batch-local strings (computed by the caller with the
// Python simulation of the same automaton).
static int run_sim(const void* blob, size_t len, const cbh_batch* in, const cbh_params* p, cbh_result* out, uint64_t* gbits, cbh_trace* trace) {
  KernelArgs a{};
  std::vector<uint32_t> meta;
  const uint8_t* base = static_cast<const uint8_t*>(blob);
  if (const char* e = cbh_parse_image(a.t, meta, base, base, len)) { g_err = e; return -1; }
  if (in->n_columns != meta[CBH_M_NCOLUMNS]) { g_err = "n_columns mismatch"; return -1; }
  BatchDev& b = a.b;
  b.n_requests = in->n_requests; b.n_tuples = in->n_tuples; b.n_roles = in->n_roles;
  b.n_columns = in->n_columns; b.n_strings = in->n_strings; b.heap_len = in->heap_len;
  b.req_lo = 0; b.req_hi = in->n_requests;
  static const uint32_t none[4] = {0, 0, 0, 0};   // the kernels read element 0 of these unconditionally (masked afterwards)
  b.req_u32 = in->req_u32; b.roles = in->roles; b.tuple_req = in->tuple_req; b.tuple_action = in->tuple_action;
  b.col_tag = in->col_tag; b.col_val = in->col_val; b.heap_tag = in->heap_tag; b.heap_val = in->heap_val;
  b.str_off = in->str_off; b.str_bytes = in->str_bytes; b.str_flags = in->str_flags; b.gbits = gbits;
  if (!b.roles) b.roles = none;
  if (!b.tuple_action) b.tuple_action = none;
  a.o = OutDev{out->effect, out->policy, out->scope, out->status, out->edr_mask, nullptr, nullptr, 0, 0};
  g_trace = trace != nullptr;
  if (trace) {
    if (!a.t.trace_pool) { g_err = "the table was lowered without the trace sections"; return -1; }
    trace->count = 0;
    a.o.trace_rec = trace->records; a.o.trace_cnt = &trace->count; a.o.trace_cap = trace->capacity;
  }
  a.now_ns = p->now_ns; a.flags = p->flags;
  if (a.o.edr) std::memset(a.o.edr, 0, sizeof(uint64_t) * in->n_requests);
  g_args = &a;
  uint32_t max_actions = 0;
  for (uint32_t r = 0; r < in->n_requests; ++r)
    max_actions = std::max(max_actions, in->req_u32[(size_t)CBH_RQ_ACT_CNT * in->n_requests + r]);
  g_max_actions = max_actions;
  uint32_t max_roles = 0;
  for (uint32_t r = 0; r < in->n_requests; ++r)
    max_roles = std::max(max_roles, in->req_u32[(size_t)CBH_RQ_ROLE_CNT * in->n_requests + r]);
  g_max_roles = max_roles;
  g_plain = getenv("CBH_FLAT_ANY") == nullptr;   // cbh_engine.hip validate_batch: the columns a classified leaf can leave the inline code on
  for (uint32_t col = 0; col < in->n_columns && col < 32; ++col) {
    if (!((meta[CBH_M_SENS_COLS] >> col) & 1u)) continue;
    for (size_t i = (size_t)col * in->n_requests; i < (size_t)(col + 1) * in->n_requests; ++i) {
      const uint32_t x = in->col_tag[i];
      if ((x - CBH_T_INT) < 2u || (x - CBH_T_LIST) < 2u) g_plain = false;
    }
  }
  // the same choice of kernels as the library makes (cbh_engine.hip plan_for); CBH_NO_FLAT / CBH_NO_WALK2 as there
  const bool has_globs = (a.t.nfa_words[0] | a.t.nfa_words[1] | a.t.nfa_words[2] | (a.t.flags & CBH_MF_HAS_ANY_PATTERN)) != 0;
  const CbhPlan pl = cbh_plan(a.t.flags, a.t.n_dr, has_globs, a.t.gslots_generic, a.t.gslots_all, g_max_actions, g_max_roles, g_plain, a.flags,
                              getenv("CBH_NO_FLAT") != nullptr, getenv("CBH_NO_WALK2") != nullptr, getenv("CBH_FORCE_STAGED") ? 0xFFFFFFFFu : a.t.max_bucket);
  std::vector<uint64_t> gres((size_t)pl.n_gwords * in->n_requests + 1, 0xDDDDDDDDDDDDDDDDull);
  b.gres = pl.n_gwords ? gres.data() : nullptr; b.n_gwords = pl.n_gwords; b.n_gslots = pl.n_gslots;
  if (const char* e = getenv("CBH_HOSTSIM_REPORT")) { if (*e == '1') std::fprintf(stderr, "hostsim: kernel kind %d, %u result words\n", pl.kind, pl.n_gwords); }
  g_last_kind = trace ? -1 : pl.kind;
  // two launches over an arbitrary (unaligned) split of the batch: the chunk window [req_lo, req_hi) that the
  // one-shot path pipelines with (cbh_engine.hip) is exercised by every test of the CPU tier
  const uint32_t n = in->n_requests, mid = n > 3 ? (n / 2) - (n / 2) % 3 + 1 : n;
  const uint32_t cuts[3] = {0, mid, n};
  const uint32_t user_flags = a.flags & ~(uint32_t)CBH_FI_MASK;
  for (int c = 0; c < 2; ++c) {
    b.req_lo = cuts[c]; b.req_hi = cuts[c + 1];
    const uint32_t nblocks = (b.req_hi - b.req_lo + CBH_BLOCK - 1) / CBH_BLOCK;   // one lane per request
    a.flags = user_flags;
    if (!trace && pl.kind == 2) {   // as cbh_engine.hip launch_plan
      if (pl.wide_kernel) { a.flags = user_flags | CBH_FI_ONLY_WIDE; g_kernel = pl.wide_kernel; for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk); a.flags = user_flags | CBH_FI_SKIP_WIDE; }
      if (pl.n_gwords) { g_kernel = cbh_walk2_pre_kernel; for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk); }
    }
    g_kernel = pl.kernel;
    for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk);
  }
  a.flags = user_flags;
  return 0;
}
extern "C" int hostsim_check(const void* blob, size_t len, const cbh_batch* in, const cbh_params* p,
                             cbh_result* out, uint64_t* gbits) {
  return run_sim(blob, len, in, p, out, gbits, nullptr);
}
extern "C" int hostsim_last_kind() { return g_last_kind; }
extern "C" int hostsim_trace(const void* blob, size_t len, const cbh_batch* in, const cbh_params* p,
                             cbh_result* out, uint64_t* gbits, cbh_trace* trace) {
  return run_sim(blob, len, in, p, out, gbits, trace);
}
