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
#include "../../cerbos_amd/csrc/cbh_wire_host.h"

static_assert(HS_BLOCK == CBH_BLOCK, "hostsim block size must match the kernels'");
static thread_local std::string g_err;
extern "C" const char* hostsim_last_error() { return g_err.c_str(); }

static KernelArgs* g_args;
// (the simulator reads the switch on every call: tests flip it between runs)
static bool cbh_flat_use_masks_now(const void* segs, uint32_t max_bucket) {
  const char* e = getenv("CBH_FLAT_MASKS");
  if ((e && *e == '0') || getenv("CBH_FORCE_STAGED")) return false;
  return segs != nullptr && ((e && *e == '1') || max_bucket > CBH_FLAT_STAGE_MIN);
}

static uint32_t g_max_actions, g_max_roles; static bool g_plain;   // same kernel selection as cbh_check_resident (cbh_engine.hip)
static cbh_check_kernel_fn g_kernel;   // the kernel the fibers run

static bool g_used_walk_awide = false;
static bool g_used_walk_wide = false;   // did the last batch launch cbh_walk2_wide_kernel? (hostsim_last_walk_wide)
static bool g_last_masks = false;
static bool g_used_pre_split = false;   // did the last batch run the pre-pass as collector + interpreter? (hostsim_last_pre_split)
static int g_last_kind = -1;   // which kernel family decided the last batch (hostsim_last_kind: tests assert the one they mean to exercise)
static bool g_trace;   // hostsim_trace: the trace pass's kernel (cbh_trace_batch)

static WireArgs* g_wargs;   // the device flattener's kernels (cbh_wire.h): 1 count, 2 scan, 3 fill
static WireRouteArgs* g_wrargs;   // ... and the routing kernels': 7 routes, 8 scan, 9 gather
static WireOutArgs* g_woargs;   // ... and the device assembler's: 4 sizes, 5 scan, 6 bytes
static WireReqArgs* g_wqargs;   // ... and the request splitter's (cbh_wire_req.h): 10 count, 11 split
static int g_wire_kind = 0;

static void fiber_main() {
  if (g_wire_kind >= 10) { if (g_wire_kind == 10) cbh_wire_req_count_kernel(*g_wqargs); else cbh_wire_req_split_kernel(*g_wqargs); }
  else if (g_wire_kind >= 7) { if (g_wire_kind == 7) cbh_wire_route_kernel(*g_wrargs); else if (g_wire_kind == 8) cbh_wire_route_scan_kernel(*g_wrargs); else cbh_wire_gather_kernel(*g_wrargs); }
  else if (g_wire_kind >= 4) { if (g_wire_kind == 4) cbh_wire_out_size_kernel(*g_woargs); else if (g_wire_kind == 5) cbh_wire_out_scan_kernel(*g_woargs); else cbh_wire_out_write_kernel(*g_woargs); }
  else if (g_wire_kind == 1) cbh_wire_count_kernel(*g_wargs);
  else if (g_wire_kind == 2) cbh_wire_scan_kernel(*g_wargs);
  else if (g_wire_kind == 3) { if (g_wargs->lds_cap) cbh_wire_fill_lds_kernel(*g_wargs); else cbh_wire_fill_kernel(*g_wargs); }
  else if (g_trace) cbh_trace_kernel(*g_args, g_args);
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

static uint32_t* g_ep = nullptr; static uint32_t g_ep_words = 0; static const uint32_t* g_ep_group = nullptr;   // hostsim_check_trail
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
  a.o.eff_pol = g_ep; a.o.ep_words = g_ep_words; b.ep_group = g_ep_group;   // hostsim_check_trail (else null)
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
                              getenv("CBH_NO_FLAT") != nullptr, getenv("CBH_NO_WALK2") != nullptr, getenv("CBH_FORCE_STAGED") ? 0xFFFFFFFFu : a.t.max_bucket, getenv("CBH_NO_WALK2_WIDE") != nullptr,
                              cbh_flat_use_masks_now(a.t.segs, a.t.max_bucket));
  std::vector<uint64_t> gres((size_t)pl.n_gwords * in->n_requests + 1, 0xDDDDDDDDDDDDDDDDull);
  b.gres = pl.n_gwords ? gres.data() : nullptr; b.n_gwords = pl.n_gwords; b.n_gslots = pl.n_gslots;
  // CBH_PRE_SPLIT=1: the pre-pass as cbh_walk2_collect_kernel + cbh_walk2_interp_kernel (as cbh_engine.hip launch_plan)
  const bool pre_split = getenv("CBH_PRE_SPLIT") && atoi(getenv("CBH_PRE_SPLIT")) != 0 && pl.n_gwords && !(a.t.flags & CBH_MF_USES_RUNTIME_EDR);
  std::vector<uint32_t> site_cnt(pl.n_gslots + 1, 0);
  std::vector<uint64_t> site_list(pre_split ? (size_t)pl.n_gslots * in->n_requests + 1 : 1, 0xEEEEEEEEEEEEEEEEull);
  if (pre_split) { b.site_cnt = site_cnt.data(); b.site_list = site_list.data(); b.site_cap = in->n_requests; }
  g_used_pre_split = false;
  if (const char* e = getenv("CBH_HOSTSIM_REPORT")) { if (*e == '1') std::fprintf(stderr, "hostsim: kernel kind %d, %u result words\n", pl.kind, pl.n_gwords); }
  g_last_kind = trace ? -1 : pl.kind;
  g_last_masks = !trace && pl.kind == 1 && cbh_is_mask_kernel(pl.kernel);
  g_used_walk_wide = false; g_used_walk_awide = false;
  // two launches over an arbitrary (unaligned) split of the batch: the chunk window [req_lo, req_hi) that the
  // one-shot path pipelines with (cbh_engine.hip) is exercised by every test of the CPU tier
  const uint32_t n = in->n_requests, mid = n > 3 ? (n / 2) - (n / 2) % 3 + 1 : n;
  const uint32_t cuts[3] = {0, mid, n};
  const uint32_t user_flags0 = a.flags & ~(uint32_t)CBH_FI_MASK;
  const char* force_packed = getenv("CBH_PACKED_TAGS");   // (as cbh_engine.hip packed_tags_pay; unset: the first launch wide, the second packed)
  for (int c = 0; c < 2; ++c) {
    // the two forms of the column cache's tags (cbh_vm.h CBH_CC_DWORDS): every test of the CPU tier runs both
    const uint32_t user_flags = user_flags0 | ((force_packed ? atoi(force_packed) != 0 : c == 1) ? CBH_FI_PACKED_TAGS : 0u);
    b.req_lo = cuts[c]; b.req_hi = cuts[c + 1];
    const uint32_t nblocks = (b.req_hi - b.req_lo + CBH_BLOCK - 1) / CBH_BLOCK;   // one lane per request
    a.flags = user_flags;
    if (!trace && pl.kind == 2) {   // as cbh_engine.hip launch_plan
      if (pl.wide_kernel) { a.flags = user_flags | ((pl.walk_wide || pl.walk_awide) ? CBH_FI_ONLY_WIDER : CBH_FI_ONLY_WIDE); g_kernel = pl.wide_kernel; for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk); }
      if (pl.wide_kernel || pl.walk_wide || pl.walk_awide) a.flags = user_flags | CBH_FI_SKIP_WIDE;
      if (pl.walk_wide) {
        if (pl.n_gwords) { g_kernel = cbh_walk2_pre_wide_kernel; for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk); }
        g_kernel = pl.trail ? cbh_walk2_wide_trail_kernel : cbh_walk2_wide_kernel; for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk);
        g_used_walk_wide = true;
      }
      if (pl.walk_awide) {
        if (pl.n_gwords) { g_kernel = cbh_walk2_pre_awide_kernel; for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk); }
        g_kernel = pl.trail ? cbh_walk2_awide_trail_kernel : cbh_walk2_awide_kernel; for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk);
        g_used_walk_awide = true;
      }
      if (pl.n_gwords && pre_split) {
        std::fill(site_cnt.begin(), site_cnt.end(), 0u);
        g_kernel = cbh_walk2_collect_kernel; for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk);
        g_kernel = cbh_walk2_interp_kernel; for (uint32_t blk = 0; blk < pl.n_gslots * nblocks; ++blk) run_block(blk);
        g_used_pre_split = true;
      } else if (pl.n_gwords) { g_kernel = cbh_walk2_pre_kernel; for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk); }
    }
    g_kernel = pl.kernel;
    for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk);
  }
  a.flags = user_flags0;
  return 0;
}
extern "C" int hostsim_check(const void* blob, size_t len, const cbh_batch* in, const cbh_params* p,
                             cbh_result* out, uint64_t* gbits) {
  return run_sim(blob, len, in, p, out, gbits, nullptr);
}
// cbh_check_batch_trail on the simulator: effective_policies [n_groups][(policies + 31) / 32], zeroed here (cerbos_hip.h)
extern "C" int hostsim_check_trail(const void* blob, size_t len, const cbh_batch* in, const cbh_params* p, cbh_result* out, uint64_t* gbits,
                                   const uint32_t* group_of_request, uint32_t n_groups, uint32_t n_policies, uint32_t* effective_policies) {
  const uint32_t words = (n_policies + 31u) / 32u;
  std::memset(effective_policies, 0, (size_t)(n_groups ? n_groups : 1u) * words * 4u);
  g_ep = effective_policies; g_ep_words = words; g_ep_group = group_of_request;
  cbh_params q = *p; q.flags |= CBH_F_WANT_EFFECTIVE_POLICIES;
  const int rc = run_sim(blob, len, in, &q, out, gbits, nullptr);
  g_ep = nullptr; g_ep_words = 0; g_ep_group = nullptr;
  return rc;
}
extern "C" int hostsim_last_kind() { return g_last_kind; }
extern "C" int hostsim_last_pre_split() { return g_used_pre_split ? 1 : 0; }
extern "C" int hostsim_last_masks() { return g_last_masks ? 1 : 0; }   // did the last batch take the flat kernel's mask walk?
extern "C" int hostsim_last_walk_wide() { return (g_used_walk_wide ? 1 : 0) | (g_used_walk_awide ? 2 : 0); }
extern "C" int hostsim_trace(const void* blob, size_t len, const cbh_batch* in, const cbh_params* p,
                             cbh_result* out, uint64_t* gbits, cbh_trace* trace) {
  return run_sim(blob, len, in, p, out, gbits, trace);
}

// ---- the device flattener (cbh_wire.h) on the simulator: the same three launches, retries and sizing as cbh_wire_flatten ----
struct HsWire {   // valid until the next call
  uint32_t n, n_tuples, n_roles, n_columns, dict_slots, heap_len, K, fill_runs;
  const uint32_t* req_u32; const uint32_t* roles; const uint32_t* tuple_action; const uint8_t* col_tag; const uint64_t* col_val;
  const uint8_t* heap_tag; const uint64_t* heap_val; const uint64_t* dict; const uint32_t* dict_flags;
  const uint32_t* in_span; const uint32_t* act_span; const uint8_t* msg; const uint8_t* status;
  WireStats stats;
  // grouped by route (cbh_wire_route_kernel ...): the per-request arrays in grouped order and input -> position; null when not grouped
  const uint32_t* req_grouped; const uint8_t* col_tag_grouped; const uint64_t* col_val_grouped; const uint32_t* inv; uint32_t n_routes, pad;
};
static struct {
  std::vector<uint8_t> msg, status, col_tag, heap_tag; std::vector<uint64_t> moff, col_val, heap_val, dict;
  std::vector<uint32_t> cnt, wavesum, waveoff, dict_flags, req, roles, tuple_action, in_span, act_span;
  uint32_t dver_off = 0, dver_len = 0, n = 0;
  std::vector<uint64_t> rt_key, col_val_g; std::vector<uint32_t> rt_cnt, slot, rank, inv, req_g; std::vector<uint8_t> col_tag_g; bool grouped = false;
} g_w;
static void wire_launch(int kind, uint32_t nblocks) {
  g_wire_kind = kind;
  for (uint32_t blk = 0; blk < nblocks; ++blk) run_block(blk);
  g_wire_kind = 0;
}
extern "C" int hostsim_wire_flatten(const void* blob, size_t len, const uint8_t* bytes, const uint64_t* offsets, uint32_t n, const char* dver,
                                    const char* dscope, const uint8_t* globals_pb, size_t globals_len, uint32_t dict_slots_hint, uint32_t heap_hint, HsWire* out) {
  TableDev t{}; std::vector<uint32_t> meta;
  const uint8_t* base = static_cast<const uint8_t*>(blob);
  if (const char* e = cbh_parse_image(t, meta, base, base, len)) { g_err = e; return -1; }
  WireIndexHost wi;
  if (const char* e = cbh_wire_index_build(wi, base, len, meta)) { g_err = e; return -1; }
  if (wi.why_not) { g_err = wi.why_not; return 1; }
  const uint64_t total = n ? offsets[n] : 0;
  std::string dv = dver ? dver : "default", ds = dscope ? dscope : "";
  if (!ds.empty() && ds[0] == '.') ds.erase(0, 1);   // scope_value
  g_w.msg.assign(bytes, bytes + total);
  g_w.msg.insert(g_w.msg.end(), dv.begin(), dv.end()); g_w.msg.insert(g_w.msg.end(), ds.begin(), ds.end());
  { const char* cl = "claims"; g_w.msg.insert(g_w.msg.end(), cl, cl + 6); }
  if (globals_len) g_w.msg.insert(g_w.msg.end(), globals_pb, globals_pb + globals_len);
  g_w.msg.resize(g_w.msg.size() + 16, 0);
  g_w.moff.assign(offsets, offsets + n + 1);
  if (n == 0) g_w.moff.assign(1, 0);
  const uint32_t nw = (n + 63) / 64;
  WireArgs a{};
  a.t_str_off = t.str_off; a.t_str_bytes = t.str_bytes; a.K = t.K; a.t_flags = t.flags;
  a.tix = wi.tix.data(); a.tix_mask = wi.tix_mask; a.scope_of_sid = wi.scope_of_sid.data();
  a.cols = wi.cols.data(); a.col_keys = wi.col_keys.data(); a.n_cols = meta[CBH_M_NCOLUMNS]; a.sens_cols = meta[CBH_M_SENS_COLS];
  a.msg = g_w.msg.data(); a.moff = g_w.moff.data(); a.n = n;
  // the fill kernel's staging block (behind its per-column cursors): CBH_WIRE_LDS_CAP bytes at most; by default calls alternate
  // between "up to 16 KB" and "never" so that the suite parses every kind of message through both kernels
  static uint32_t flip = 0;
  uint32_t lds_most;
  { const char* e = getenv("CBH_WIRE_LDS_CAP"); lds_most = e ? (uint32_t)atoi(e) : ((flip++ & 1u) ? 0u : 16384u);
    const uint32_t room = (uint32_t)sizeof(cbh_dyn_lds) - cbh_wire_fill_cur_bytes(a.n_cols); if (lds_most > room) lds_most = room; }
  a.lds_cap = 0;
  a.dver_off = (uint32_t)total; a.dver_len = (uint32_t)dv.size(); a.dscope_off = (uint32_t)(total + dv.size()); a.dscope_len = (uint32_t)ds.size();
  a.claims_off = (uint32_t)(total + dv.size() + ds.size());
  a.globals_off = a.claims_off + 6; a.globals_len = (uint32_t)globals_len;
  g_w.cnt.assign(n + 1, 0); g_w.status.assign(n + 1, 0); g_w.wavesum.assign(4 * (size_t)nw + 4, 0); g_w.waveoff.assign(2 * (size_t)nw + 4, 0);
  a.cnt = g_w.cnt.data(); a.status = g_w.status.data(); a.wavesum = g_w.wavesum.data(); a.waveoff = g_w.waveoff.data();
  WireStats st; cbh_wire_stats_init(st);
  a.stats = &st;
  g_wargs = &a;
  a.lds_cap = lds_most;   // (the count kernel stages a wave's block when it fits, as the fill does)
  wire_launch(1, nw);
  uint32_t slots = dict_slots_hint ? dict_slots_hint : cbh_wire_dict_slots(n);
  uint32_t heap_cap = heap_hint ? heap_hint : cbh_wire_heap_guess(total);
  uint32_t runs = 0;
  for (;;) {
    g_w.dict.assign(slots, 0); g_w.dict_flags.assign(slots / 4 + 1, 0);
    a.lix = g_w.dict.data(); a.lix_mask = slots - 1; a.lflags = g_w.dict_flags.data();
    st.flags = 0; st.heap_used = 0; st.route_lo = st.route_hi = st.multi_route = 0;
    const uint32_t n_host0 = st.n_host;
    wire_launch(2, 1);
    g_w.req.assign((size_t)CBH_RQ_NFIELDS * n + 1, 0xDDDDDDDDu); g_w.roles.assign((size_t)st.n_roles + 1, 0xDDDDDDDDu);
    g_w.tuple_action.assign((size_t)st.n_tuples + 1, 0xDDDDDDDDu);
    g_w.col_tag.assign((size_t)a.n_cols * n + 1, 0xDD); g_w.col_val.assign((size_t)a.n_cols * n + 1, 0xDDDDDDDDDDDDDDDDull);
    g_w.heap_tag.assign((size_t)heap_cap + 1, 0xDD); g_w.heap_val.assign((size_t)heap_cap + 1, 0);
    g_w.in_span.assign((size_t)n * 2 * CBH_WSPAN_N + 1, 0); g_w.act_span.assign((size_t)st.n_tuples * 2 + 1, 0);
    a.req_u32 = g_w.req.data(); a.roles = g_w.roles.data(); a.tuple_action = g_w.tuple_action.data(); a.col_tag = g_w.col_tag.data(); a.col_val = g_w.col_val.data();
    a.heap_tag = g_w.heap_tag.data(); a.heap_val = g_w.heap_val.data(); a.heap_cap = heap_cap; a.in_span = g_w.in_span.data(); a.act_span = g_w.act_span.data();
    a.lds_cap = (st.max_block && (uint64_t)st.max_block + 16u + CBH_WIRE_SLACK <= lds_most) ? lds_most : 0u;   // (cbh_engine.hip wire_fill_lds_cap)
    wire_launch(3, nw);
    ++runs;
    if (st.flags & CBH_WF_DICT_FULL) { if (slots >= (1u << 30)) { g_err = "dictionary cannot grow"; return -1; } slots *= 4; st.n_host = n_host0; continue; }
    if (st.heap_used > heap_cap) { heap_cap = st.heap_used; st.n_host = n_host0; continue; }
    break;
  }
  out->n = n; out->n_tuples = st.n_tuples; out->n_roles = st.n_roles; out->n_columns = a.n_cols; out->dict_slots = slots; out->heap_len = st.heap_used;
  out->K = t.K; out->fill_runs = runs;
  out->req_u32 = g_w.req.data(); out->roles = g_w.roles.data(); out->tuple_action = g_w.tuple_action.data(); out->col_tag = g_w.col_tag.data();
  out->col_val = g_w.col_val.data(); out->heap_tag = g_w.heap_tag.data(); out->heap_val = g_w.heap_val.data(); out->dict = g_w.dict.data();
  out->dict_flags = g_w.dict_flags.data(); out->in_span = g_w.in_span.data(); out->act_span = g_w.act_span.data(); out->msg = g_w.msg.data();
  out->status = g_w.status.data(); out->stats = st;
  g_w.dver_off = a.dver_off; g_w.dver_len = a.dver_len; g_w.n = n;
  // grouping by route, as cbh_wire_flatten does after a fill that left nothing to the host
  out->req_grouped = nullptr; out->col_tag_grouped = nullptr; out->col_val_grouped = nullptr; out->inv = nullptr; out->n_routes = 0; g_w.grouped = false;
  const char* ge = getenv("CBH_WIRE_GROUP");   // (as the library: on unless CBH_WIRE_GROUP=0; CBH_WIRE_GROUP=1 also groups batches smaller than two waves)
  if (n && st.n_host == 0 && st.first_bad == CBH_NONE && !(ge && *ge == '0') && (n >= 2u * CBH_BLOCK || (ge && *ge == '1'))) {
    WireRouteArgs r{};
    r.n = n; r.n_cols = a.n_cols; r.req_u32 = g_w.req.data(); r.roles = g_w.roles.data(); r.col_tag = g_w.col_tag.data(); r.col_val = g_w.col_val.data();
    g_w.rt_key.assign(CBH_WIRE_ROUTE_SLOTS, 0); g_w.rt_cnt.assign(CBH_WIRE_ROUTE_SLOTS + 2, 0); g_w.slot.assign(n, 0); g_w.rank.assign(n, 0); g_w.inv.assign(n, 0xDDDDDDDDu);
    g_w.req_g.assign((size_t)CBH_RQ_NFIELDS * n, 0xDDDDDDDDu); g_w.col_tag_g.assign((size_t)a.n_cols * n + 1, 0xDD); g_w.col_val_g.assign((size_t)a.n_cols * n + 1, 0);
    r.multi = &st.multi_route;
    r.rt_key = g_w.rt_key.data(); r.rt_cnt = g_w.rt_cnt.data(); r.slot = g_w.slot.data(); r.rank = g_w.rank.data(); r.inv = g_w.inv.data();
    r.req_out = g_w.req_g.data(); r.col_tag_out = g_w.col_tag_g.data(); r.col_val_out = g_w.col_val_g.data();
    g_wrargs = &r;
    wire_launch(7, nw);
    if (!g_w.rt_cnt[CBH_WIRE_ROUTE_SLOTS + 1]) {
      wire_launch(8, 1);
      wire_launch(9, nw);
      out->req_grouped = g_w.req_g.data(); out->col_tag_grouped = g_w.col_tag_g.data(); out->col_val_grouped = g_w.col_val_g.data(); out->inv = g_w.inv.data();
      out->n_routes = g_w.rt_cnt[CBH_WIRE_ROUTE_SLOTS]; g_w.grouped = true;
    }
  }
  return 0;
}

// The request splitter (cbh_wire_req.h) on the simulator: n serialized CheckResourcesRequests (+ per-request engine AuxData, or
// null) -> the CheckInputs of their resource entries, as cbh_wire_flatten_requests makes them on the device.  Returns the number of
// inputs (the buffers are valid until the next call), -1 with *first_bad = the first malformed request.
static struct { std::vector<uint8_t> msg, flags; std::vector<uint64_t> moff, nbytes, first_byte; std::vector<uint32_t> ninputs, first_input; } g_q;
extern "C" long long hostsim_wire_split_requests(const uint8_t* bytes, const uint64_t* offsets, uint32_t n, const uint8_t* aux, const uint64_t* aoff,
                                                 const uint8_t** out_msg, const uint64_t** out_moff, const uint32_t** out_first_input,
                                                 const uint8_t** out_flags, uint32_t* first_bad) {
  WireReqArgs a{};
  a.req = bytes; a.roff = offsets; a.n = n; a.end = n ? (uint32_t)offsets[n] : 0; a.aux = aux; a.aoff = aoff; a.aux_end = (aoff && n) ? aoff[n] : 0;
  g_q.ninputs.assign(n + 1, 0); g_q.nbytes.assign(n + 1, 0); g_q.flags.assign(n + 1, 0);
  a.n_inputs = g_q.ninputs.data(); a.n_bytes = g_q.nbytes.data(); a.flags = g_q.flags.data();
  g_wqargs = &a;
  wire_launch(10, (n + CBH_BLOCK - 1) / CBH_BLOCK);
  g_q.first_input.assign(n + 1, 0); g_q.first_byte.assign(n + 1, 0);
  *first_bad = CBH_NONE;
  for (uint32_t r = 0; r < n; ++r) {
    if (g_q.ninputs[r] == CBH_WREQ_BAD) { *first_bad = r; return -1; }
    g_q.first_input[r + 1] = g_q.first_input[r] + g_q.ninputs[r];
    g_q.first_byte[r + 1] = g_q.first_byte[r] + g_q.nbytes[r];
  }
  const uint32_t total_inputs = g_q.first_input[n];
  g_q.msg.assign(g_q.first_byte[n] + 1, 0xEE); g_q.moff.assign((size_t)total_inputs + 1, 0xEEEEEEEEEEEEEEEEull);
  a.first_input = g_q.first_input.data(); a.first_byte = g_q.first_byte.data(); a.msg = g_q.msg.data(); a.moff = g_q.moff.data();
  if (n) wire_launch(11, (n + (CBH_BLOCK / 64) - 1) / (CBH_BLOCK / 64)); else g_q.moff[0] = 0;
  *out_msg = g_q.msg.data(); *out_moff = g_q.moff.data(); *out_first_input = g_q.first_input.data(); *out_flags = g_q.flags.data();
  return (long long)total_inputs;
}

// The device assembler on the batch the last hostsim_wire_flatten call built: results (input order) -> serialized CheckOutputs.
// out_bytes / out_off / out_flags are the caller's; returns the bytes needed (nothing is written beyond `cap`), < 0 on error.
extern "C" long long hostsim_wire_outputs(const void* blob, size_t len, const uint8_t* effect, const uint32_t* policy, const uint32_t* scope, const uint8_t* status,
                                          const uint64_t* edr, uint8_t* out_bytes, size_t cap, uint64_t* out_off, uint8_t* out_flags, int edr_is_grouped) {
  TableDev t{}; std::vector<uint32_t> meta;
  const uint8_t* base = static_cast<const uint8_t*>(blob);
  if (const char* e = cbh_parse_image(t, meta, base, base, len)) { g_err = e; return -1; }
  WireIndexHost wi;
  if (const char* e = cbh_wire_index_build(wi, base, len, meta)) { g_err = e; return -1; }
  const uint32_t n = g_w.n, nw = (n + 63) / 64;
  WireOutArgs a{};
  a.t_str_off = t.str_off; a.t_str_bytes = t.str_bytes; a.scope_sid = reinterpret_cast<const uint32_t*>(base + wi.scope_sid_offset); a.n_scopes = wi.n_scopes;
  a.n_policies = wi.n_policies; a.name_off = wi.name_off.data(); a.name_bytes = wi.name_bytes.data(); a.n_dr = wi.n_dr; a.n = n;
  a.msg = g_w.msg.data(); a.moff = g_w.moff.data(); a.dver_off = g_w.dver_off; a.dver_len = g_w.dver_len;
  a.req_u32 = g_w.req.data(); a.tuple_action = g_w.tuple_action.data(); a.in_span = g_w.in_span.data(); a.act_span = g_w.act_span.data();
  a.effect = effect; a.policy = policy; a.scope = scope; a.status = status; a.edr = edr;
  a.inv = (edr_is_grouped && g_w.grouped) ? g_w.inv.data() : nullptr;
  { const char* e = getenv("CBH_WIRE_LDS_CAP"); a.lds_cap = e ? (uint32_t)atoi(e) : 16384u; if (a.lds_cap > sizeof(cbh_dyn_lds)) a.lds_cap = sizeof(cbh_dyn_lds); }
  std::vector<uint32_t> sizes(n + 1, 0); std::vector<uint64_t> wavesum(nw + 1, 0), waveoff(nw + 1, 0);
  WireOutStats st{}; 
  a.sizes = sizes.data(); a.wavesum = wavesum.data(); a.waveoff = waveoff.data(); a.stats = &st; a.out_off = out_off; a.out_flags = out_flags;
  a.out_bias = (uint32_t)(reinterpret_cast<uintptr_t>(out_bytes) & 15u); a.out = out_bytes - a.out_bias;   // (as cbh_wire_outputs hands over a caller's page-locked buffer)
  g_woargs = &a;
  wire_launch(4, nw);
  wire_launch(5, 1);
  if (st.errors) { g_err = "device assembler: a policy / scope id out of range, or an output too large"; return -1; }
  if (st.total <= cap) wire_launch(6, nw);
  return (long long)st.total;
}
