// TEST INFRASTRUCTURE ONLY - never loaded by the product (cerbos_amd.capi loads
// libcerbos_hip.so and nothing else).
//
// Compiles the *device* source of the decision kernel (cerbos_amd/csrc/cbh_kernels.h) as
// plain host C++ by shimming the HIP keywords, and runs it one "lane" at a time.  This lets
// the CPU-only test tier (-m "not gpu") exercise the lowering + bytecode + kernel logic
// against the oracle and the reference's golden cases without a GPU.  It proves nothing
// about performance and is not a fallback: the GPU tier re-runs the same cases through the
// real library on an MI355X.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#define CBH_HOSTSIM 1
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(x)
struct uint4 { uint32_t x, y, z, w; };
struct Dim3 { uint32_t x = 0, y = 0, z = 0; };
static thread_local Dim3 threadIdx, blockIdx;
static inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; std::memcpy(&v, &d, 8); return v; }
static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p |= v; return o; }
using std::trunc;

#include "../../cerbos_amd/csrc/cbh_kernels.h"
#include "../../cerbos_amd/csrc/cbh_image.h"

static thread_local std::string g_err;

extern "C" const char* hostsim_last_error() { return g_err.c_str(); }

// gbits: [3][n_strings] glob match bits of the batch-local strings (computed by the caller with
// the Python simulation of the same automaton; the resolve kernel itself needs real LDS barriers).
extern "C" int hostsim_check(const void* blob, size_t len, const cbh_batch* in, const cbh_params* p,
                             cbh_result* out, uint64_t* gbits) {
  TableDev t{};
  std::vector<uint32_t> meta;
  const uint8_t* base = static_cast<const uint8_t*>(blob);
  if (const char* e = cbh_parse_image(t, meta, base, base, len)) { g_err = e; return -1; }
  if (in->n_columns != meta[CBH_M_NCOLUMNS]) { g_err = "n_columns mismatch"; return -1; }
  BatchDev b{};
  b.n_requests = in->n_requests; b.n_tuples = in->n_tuples; b.n_roles = in->n_roles;
  b.n_columns = in->n_columns; b.n_strings = in->n_strings; b.heap_len = in->heap_len;
  b.req_u32 = in->req_u32; b.roles = in->roles; b.tuple_req = in->tuple_req; b.tuple_action = in->tuple_action;
  b.col_tag = in->col_tag; b.col_val = in->col_val; b.heap_tag = in->heap_tag; b.heap_val = in->heap_val;
  b.str_off = in->str_off; b.str_bytes = in->str_bytes; b.str_flags = in->str_flags; b.gbits = gbits;
  OutDev o{out->effect, out->policy, out->scope, out->status, out->edr_mask};
  if (o.edr) std::memset(o.edr, 0, sizeof(uint64_t) * in->n_requests);
  const uint32_t nblocks = (in->n_tuples + CBH_BLOCK - 1) / CBH_BLOCK;
  for (uint32_t blk = 0; blk < nblocks; ++blk)
    for (uint32_t th = 0; th < CBH_BLOCK; ++th) {
      blockIdx.x = blk; threadIdx.x = th;
      cbh_check_kernel(t, b, o, p->now_ns, p->flags);
    }
  return 0;
}
