// Device code of libcerbos_hip.so: the glob-resolve kernel and the decision kernel.
// Included by cbh_engine.hip (the product) and, with the HIP keywords shimmed, by the
// test-only host simulation under tests/hostsim (functional checks without a GPU).
#pragma once
#include "cbh_interp.h"
#ifndef NFA_MAXW
#define NFA_MAXW 8
#endif

// ======================================================================== device code

__device__ __forceinline__ u32 hash4(u32 a, u32 b, u32 c, u32 d) {
  u32 h = a * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + b * 0x85EBCA77u;
  h = (h ^ (h >> 13)) + c * 0xC2B2AE3Du;
  h = (h ^ (h >> 16)) + d * 0x27D4EB2Fu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  return h;
}

__device__ inline bool dir_find(const TableDev& t, u32 k0, u32 k1, u32 k2, u32 k3, uint4& v) {
  u32 i = hash4(k0, k1, k2, k3) & t.hash_mask;
  for (u32 probe = 0; probe <= t.hash_mask; ++probe) {
    const CBH_G uint4* s = (const CBH_G uint4*)(&t.hash[i]);
    uint4 k = s[0];
    if (k.x == CBH_NONE) return false;
    if (k.x == k0 && k.y == k1 && k.z == k2 && k.w == k3) { v = s[1]; return true; }
    i = (i + 1) & t.hash_mask;
  }
  return false;
}

__device__ __forceinline__ u64 gbits_of(const TableDev& t, const BatchDev& b, u32 dim, u32 sid) {
  if (t.nfa_words[dim] == 0) return 0;   // no glob patterns in this dimension: nothing to look up
  if (sid < t.K) return t.gbits[(size_t)dim * t.K + sid];
  return b.gbits[(size_t)dim * b.n_strings + (sid - t.K)];
}
__device__ __forceinline__ bool pat_match(u32 pref, u32 sid, u64 bits) {
  if (pref == CBH_PAT_ANY) return true;
  return (pref & CBH_PAT_GLOB) ? ((bits >> (pref & 63u)) & 1ull) != 0 : pref == sid;
}

#define DIM_ACTION 0
#define DIM_ROLE 1
#define DIM_KIND 2
#define FLAG_RES 1u
#define FLAG_PRIN 2u
#define SP_OVERRIDE_PARENT 1u
#define SP_REQUIRE_CONSENT 2u
#define EFF_NO_MATCH 0u

__device__ __forceinline__ u32 chain_next(const TableDev& t, u32 si, u32 flagbit) {
  while (si != CBH_NONE && !(t.scope_flags[si] & flagbit)) si = t.scope_parent[si];
  return si;
}
// first scope of the chain for a request scope word (ruletable.go:848-882)
__device__ __forceinline__ u32 chain_first(const TableDev& t, u32 raw, u32 flagbit, bool lenient) {
  u32 si = raw & ~CBH_SCOPE_EXACT;
  const bool exact = (raw & CBH_SCOPE_EXACT) != 0;
  // (the scope's flags are read ONCE: read again by chain_next they were a second dependent round trip for every request)
  u32 f = t.scope_flags[si < t.n_scopes ? si : 0u];   // (unconditional, from a valid place: the word is the flatteners', but nothing checked it)
  if (si >= t.n_scopes || (!lenient && !(exact && (f & flagbit)))) return CBH_NONE;
  while (!(f & flagbit)) {
    si = t.scope_parent[si];
    if (si == CBH_NONE) break;
    f = t.scope_flags[si];
  }
  return si;
}

#if !defined(CBH_HOSTSIM) || defined(CBH_HOSTSIM_ENGINE)   /* (the kernel simulation gets the glob bits from its caller; the engine's does not) */
// --- glob NFA ----------------------------------------------------------------------------
#ifndef NFA_MAXW
#define NFA_MAXW 8
#endif
__global__ __launch_bounds__(CBH_BLOCK) void cbh_resolve_globs_kernel(TableDev t, BatchDev b) {
#ifdef CBH_HOSTSIM
  static unsigned char smem[(2 + 512) * NFA_MAXW * 8];
#else
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#endif
  u64* lds = reinterpret_cast<u64*>(smem);
  const u32 i = blockIdx.x * CBH_BLOCK + threadIdx.x;
  for (u32 dim = 0; dim < 3; ++dim) {
    const u32 nw = t.nfa_words[dim];
    if (nw == 0) continue;  // uniform
    // stage init | star | cls[256] | self[256] into LDS (coalesced 8-byte loads)
    const u32 total = (2 + 512) * nw;
    __syncthreads();
    for (u32 k = threadIdx.x; k < total; k += CBH_BLOCK) lds[k] = t.nfa[dim][k];
    __syncthreads();
    if (i >= b.n_strings) continue;
    if (!(b.str_flags[i] & (1u << dim))) { b.gbits[(size_t)dim * b.n_strings + i] = 0; continue; }
    const u64* init = lds; const u64* star = lds + nw; const u64* cls = lds + 2 * nw; const u64* self = cls + 256 * nw;
    u64 A[NFA_MAXW];
#pragma unroll
    for (int w = 0; w < NFA_MAXW; ++w) A[w] = (u32)w < nw ? init[w] : 0;
    u32 o, n;
    if (b.str_keys) { const u64 key = b.str_keys[i]; o = (u32)key; n = (u32)(key >> 32) & 0xFFFFu; }   // a batch the device flattened (cbh_wire.h)
    else { o = b.str_off[i]; n = b.str_off[i + 1] - o; }
    for (u32 k = 0; k < n; ++k) {
      const u32 ch = b.str_bytes[o + k];
      u64 carry = 0;
#pragma unroll
      for (int w = 0; w < NFA_MAXW; ++w) {
        if ((u32)w < nw) {
          const u64 adv = A[w] & cls[ch * nw + w];
          const u64 stay = A[w] & self[ch * nw + w];
          A[w] = (adv << 1) | carry | stay;
          carry = adv >> 63;
        }
      }
      // epsilon closure over star positions (a star may match the empty string)
      for (int it = 0; it < 8; ++it) {
        u64 c2 = 0; bool changed = false;
#pragma unroll
        for (int w = 0; w < NFA_MAXW; ++w) {
          if ((u32)w < nw) {
            const u64 s = A[w] & star[w];
            const u64 nx = A[w] | (s << 1) | c2;
            c2 = s >> 63;
            changed |= nx != A[w];
            A[w] = nx;
          }
        }
        if (!changed) break;
      }
    }
    // accept table follows the transition arrays in global memory
    const CBH_G u32* acc = (const CBH_G u32*)(t.nfa[dim] + (size_t)(2 + 512) * nw);
    const u32 nacc = acc[0];
    u64 out = 0;
    for (u32 k = 0; k < nacc; ++k) {
      const u32 bit = acc[2 + 2 * k], gi = acc[2 + 2 * k + 1];
      u64 word = 0;
#pragma unroll
      for (int w = 0; w < NFA_MAXW; ++w) if ((u32)w == (bit >> 6)) word = A[w];
      if ((word >> (bit & 63u)) & 1ull) out |= 1ull << gi;
    }
    b.gbits[(size_t)dim * b.n_strings + i] = out;
  }
}

#endif  // CBH_HOSTSIM (kernel simulation)

#include "cbh_check_wave.h"
#include "cbh_check_flat.h"
#include "cbh_check_walk2.h"
#include "cbh_wire.h"
#include "cbh_wire_req.h"
