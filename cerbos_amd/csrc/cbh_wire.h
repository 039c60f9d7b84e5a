// cbh_wire.h - device-side ingest (SURVEY §8 f1 on the GPU): serialized enginev1.CheckInput messages -> the batch arrays the
// decision kernels read, without the host touching a byte of the messages.
//
// What it replaces: the reference builds its request view per input on the CPU (check.go:536-554 checkInputToRequest, the
// protobuf decode in front of it); round 2 did the same walk in C++ (cbh_ingest.cpp flatten_slice: 63 % of the wire-inclusive
// wall time, the GPU idle meanwhile).  Here the raw bytes + offsets cross PCIe once and three launches build the batch in HBM:
//
//   cbh_wire_count_kernel   one lane per message: actions, roles, well-formedness of the top level; per-wave sums
//   cbh_wire_scan_kernel    one wave: exclusive offsets of the waves' action / role slices, totals, the call's default strings
//   cbh_wire_fill_kernel    one lane per message: every field of include/cerbos_hip.h's cbh_batch - request words, role and
//                           action ids, attribute columns, nested values in the tagged heap - plus where the strings the
//                           response needs sit in the message (the host assembler reads them instead of re-walking)
//
// Strings.  Equality on the device is id equality, so every string is interned: against the table's pool through a hash
// index built at table load (cbh_engine.hip wire_index_build), and - for strings the table does not know - into a
// batch-local open-addressing dictionary in HBM.  A dictionary slot is ONE 64-bit word {hash:16 | length:16 | byte offset
// of the first occurrence in the message buffer:32} claimed with one device-scope compare-and-swap: the word refers to
// message bytes that were uploaded before the launch and never change, so the claim publishes everything a later finder
// needs and no lane ever waits for another.  (A probe may read a stale EMPTY from its L1 / its XCD's L2 - the CAS that
// follows is performed at memory and returns the truth; a non-empty word never changes.)  The id of a batch-local string is
// K + its slot: the decision kernels find its bytes through BatchDev.str_keys (cbh_vm.h str_view, cbh_resolve_globs_kernel).
//
// Scope of the device path: messages the host flattener would split (> 64 actions), rewrite (resource kinds of the pre-0.30
// form, namer.go:213-218) or that nest containers deeper than CBH_WIRE_MAX_DEPTH are counted in WireStats.n_host and the
// caller takes the whole batch through libcerbos_ingest.so instead (product code, same result) - never a guess.
// Same message grammar, same last-field-wins rules, same tags and values as cbh_ingest.cpp; tests/test_wire_device.py holds
// the two against each other array by array.
#pragma once
#include <stddef.h>
#include "cbh_vm.h"

#define CBH_WIRE_MAX_KEYS 4u       /* keys of a column path */
#define CBH_WIRE_MAX_DEPTH 8u      /* containers nested deeper: host flattener */
#define CBH_WIRE_MAX_ROLES 255u
#define CBH_WIRE_MAX_STRLEN 0xFFFFu
#define CBH_WIRE_MAX_VALUE_ENTRIES 0xFFFFFu   /* heap entries of one attribute value */
#define CBH_WIRE_MAX_PROBES 256u
#define CBH_WIRE_CUR_COLS 16u      /* columns whose first key is found by the one pass per attribute map (LDS: 8 B per column and lane) */

#define CBH_WS_OK 0u
#define CBH_WS_BAD 1u    /* malformed message */
#define CBH_WS_HOST 2u   /* well formed, left to the host flattener */

#define CBH_WF_CONTAINER_IN_SENS 1u   /* a list / map sits in a column a classified leaf is sensitive to (BatchShape::plain_tags) */
#define CBH_WF_DICT_FULL 2u           /* the batch-local dictionary ran out of probes: the caller retries with a larger one */

#define CBH_WSPAN_N 6u   /* request id, principal id / version, resource kind / version / id: (offset, length) pairs per message */

struct WireCol { u32 root, nk; u32 key_off[CBH_WIRE_MAX_KEYS], key_len[CBH_WIRE_MAX_KEYS]; u32 key_hash0, pad; };   // key_hash0 = cbh_wire_hash of the first key

struct WireStats {
  u32 n_tuples, n_roles;        // scan kernel
  u32 max_actions, max_roles;   // count kernel
  u32 wide_lo, wide_hi;         // requests wider than cbh_walk2_kernel's shape lie in [wide_lo, wide_hi)
  u32 first_bad;                // smallest index of a malformed message (CBH_NONE: none)
  u32 n_host;                   // messages left to the host flattener
  u32 heap_used;                // heap entries the fill asked for (may exceed the capacity: then it is re-run)
  u32 flags;                    // CBH_WF_*
  u32 sid_empty, sid_dver, dscope_word, sid_claims;
  u32 max_block;                // count kernel: the most bytes any wave's 64 messages span, from the 16-byte boundary below the first
  u32 pad;
  // fill kernel: does the batch hold more than ONE route (kind, version, scope)?  The first route seen (its fingerprint, claimed by
  // one compare-and-swap) and a flag any wave sets that holds another: the routing kernels leave at once when it stays 0 - a stream of
  // one kind otherwise sends every wave's atomic to the same counter for ranks nobody uses
  u32 route_lo, route_hi, multi_route, pad2;
};

struct WireArgs {
  // the table
  const CBH_G u32* t_str_off; const CBH_G u8* t_str_bytes; u32 K; u32 t_flags;
  const CBH_G u64* tix; u32 tix_mask;          // [2 * (mask + 1)] slots of {hash << 32 | id + 1 (0 = empty), offset << 32 | length}
  const CBH_G u32* scope_of_sid;               // [K] scope index of a table string that is a scope, else CBH_NONE
  const CBH_G WireCol* cols; const CBH_G u8* col_keys; u32 n_cols; u32 sens_cols;
  // the messages: message i = msg[moff[i] .. moff[i + 1]); the call's default version / scope and "claims" follow the last one
  const CBH_G u8* msg; const CBH_G u64* moff; u32 n; u32 heap_cap;
  u32 lds_cap, pad2;   // bytes of dynamic LDS a wave of the fill kernel may stage its messages in (0: parse in place)
  u32 dver_off, dver_len, dscope_off, dscope_len, claims_off, pad1;
  u32 globals_off, globals_len;   // the call's globals, a serialized google.protobuf.Struct behind the messages (columns of root 4)   // ("claims": the key of the request view of a named JWT)
  // scratch
  CBH_G u32* cnt;         // [n] actions | roles << 8
  CBH_G u8* status;       // [n] CBH_WS_*
  CBH_G u32* wavesum;     // [waves][4] actions, roles, largest action count | largest role count << 16, bytes of the wave's block of messages
                          // (reduced by the one wave of the scan kernel: an atomic per wave on one statistics word costs a launch of
                          // 2 000 waves more than everything else it does)
  CBH_G u32* waveoff;     // [waves][2] exclusive
  CBH_G WireStats* stats;
  CBH_G WireStats* host_stats;   // page-locked host memory the device can write (or null): the scan kernel leaves a copy of `stats` there
  // the batch-local dictionary
  CBH_G u64* lix; u32 lix_mask; u32 pad0; CBH_G u32* lflags;   // flags: one byte per slot, OR-ed through the aligned dword
  // the batch
  CBH_G u32* req_u32; CBH_G u32* roles; CBH_G u32* tuple_action; CBH_G u8* col_tag; CBH_G u64* col_val;
  CBH_G u8* heap_tag; CBH_G u64* heap_val;
  CBH_G u32* in_span; CBH_G u32* act_span;
};

// ---- the string hash both sides use (the table index is built on the host) --------------------------------------
template <class P>
#ifndef CBH_HOSTSIM
__host__ __device__
#endif
static inline u32 cbh_wire_hash(P p, u32 n) {
  u32 h = 0x811C9DC5u ^ n;
  for (u32 i = 0; i < n; ++i) { h ^= (u32)p[i]; h *= 0x01000193u; }
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

// ---- atomics (device scope); the host simulation runs its fibers on one thread -----------------------------------
#ifdef CBH_HOSTSIM
static inline u64 w_cas64(u64* p, u64 expect, u64 v) { const u64 o = *p; if (o == expect) *p = v; return o; }
static inline u64 w_load64(const u64* p) { return *p; }
static inline u32 w_load32(const u32* p) { return *p; }
static inline u32 w_add32(u32* p, u32 v) { const u32 o = *p; *p += v; return o; }
static inline void w_or32(u32* p, u32 v) { *p |= v; }
static inline void w_max32(u32* p, u32 v) { if (v > *p) *p = v; }
static inline void w_min32(u32* p, u32 v) { if (v < *p) *p = v; }
#else
__device__ __forceinline__ u64 w_cas64(CBH_G u64* p, u64 expect, u64 v) {
  return (u64)atomicCAS((unsigned long long*)p, (unsigned long long)expect, (unsigned long long)v);
}
__device__ __forceinline__ u64 w_load64(const CBH_G u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 w_load32(const CBH_G u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // (from L2: this wave's L1 may hold the line from before another wave's atomic)
__device__ __forceinline__ u32 w_add32(CBH_G u32* p, u32 v) { return atomicAdd((unsigned int*)p, v); }
__device__ __forceinline__ void w_or32(CBH_G u32* p, u32 v) { atomicOr((unsigned int*)p, v); }
__device__ __forceinline__ void w_max32(CBH_G u32* p, u32 v) { atomicMax((unsigned int*)p, v); }
__device__ __forceinline__ void w_min32(CBH_G u32* p, u32 v) { atomicMin((unsigned int*)p, v); }
#endif

// The bytes a lane parses: MP is `const CBH_L u8*` - the fill kernel stages its wave's messages in LDS with one coalesced copy when
// they fit - or `const CBH_G u8*`, the message buffer itself.  The walk is a chain of DEPENDENT loads (a field's position follows from
// the length of the one before), so what it costs is round trips: an LDS read is several times shorter than one to L2, and a pointer
// whose address space the compiler knows gives ds_read / global_load instead of a flat access that pays the longer of both (measured:
// the staged block parsed through a generic pointer was no faster than the bytes in place, profiles/r03_wire_lds_ab.json).  And every
// round trip fetches EIGHT bytes (w_peek8: one ds_read_b64 / global_load_dwordx2 at any alignment): a field's tag and its length, or
// its varint value, come out of one window, a string is hashed and compared eight bytes at a time.  `m[off]`: off = byte offset in
// WireArgs.msg either way (the LDS view is biased by the block's start).  Every buffer read this way ends in CBH_WIRE_SLACK bytes
// nobody parses (the message buffer: the call's tail strings and the allocation's slack; the staged block: checked against lds_cap).
#define CBH_WIRE_SLACK 8u
typedef const CBH_G u8* WGlob;
typedef const CBH_L u8* WLds;
typedef WGlob WMsg;   // (where only the message buffer itself is ever read: the count / scan kernels, cbh_wire_req.h)

// (a template, not two overloads: the host pass of hipcc sees one pointer type where the device pass sees two address spaces)
template <class P>
__device__ __forceinline__ u64 w_peek8(P m, u32 p) {
  u64 v;
#ifdef CBH_HOSTSIM
  memcpy(&v, m + p, 8);
#else
  __builtin_memcpy(&v, m + p, 8);
#endif
  return v;
}

// ---- protobuf wire walking (the grammar of cbh_ingest.cpp next / entry / value / map_get) -------------------------
struct WSpan { u32 p, e; };   // byte offsets into WireArgs.msg
struct WField { u32 num, wt; u64 v; WSpan s; };
struct WVal { u32 kind; u64 v; WSpan s; };

// the seven payload bits of each of eight bytes (continuation bits already masked off) closed up into 56 bits
#if defined(CBH_WIRE_OOL) && !defined(CBH_HOSTSIM)
// CBH_WIRE_OOL (an experiment's build): the fold as a real function, called only where a lane meets a multi-byte varint - the fill
// kernel's code is two to three times the instruction cache and this fold sits inlined behind every one-byte fast path of it
__device__ __attribute__((noinline)) u64 w_fold7(u64 x) {
#else
__device__ __forceinline__ u64 w_fold7(u64 x) {
#endif
  x = ((x & 0x7F007F007F007F00ull) >> 1) | (x & 0x007F007F007F007Full);
  x = ((x & 0x3FFF00003FFF0000ull) >> 2) | (x & 0x00003FFF00003FFFull);
  x = ((x & 0x0FFFFFFF00000000ull) >> 4) | (x & 0x000000000FFFFFFFull);
  return x;
}
// a varint of at most eight bytes in the low bytes of `w`: its value; returns its length (0: it does not end within the window)
__device__ __forceinline__ u32 w_varint_win(u64 w, u64& out) {
  const u64 stop = ~w & 0x8080808080808080ull;
  if (!stop) return 0u;
  const u32 n = ((u32)__builtin_ctzll(stop) >> 3) + 1u;
  out = w_fold7(w & (~0ull >> (64u - 8u * n)) & 0x7F7F7F7F7F7F7F7Full);
  return n;
}
// ... and of nine or ten: `w` = its first eight bytes (every one with the continuation bit), the rest follows at m[p + 8].  A tenth
// byte contributes its lowest bit (the others fall off the 64), as in the byte-by-byte form of cbh_ingest.cpp; an eleventh is an error.
template <class MP>
__device__ __forceinline__ u32 w_varint_long(MP m, u32 p, u32 avail, u64 w, u64& out) {
  if (avail < 9u) return 0u;
  const u64 x = w_fold7(w & 0x7F7F7F7F7F7F7F7Full);
  const u32 w2 = (u32)w_peek8(m, p + 8u);
  if (!(w2 & 0x80u)) { out = x | ((u64)(w2 & 0x7Fu) << 56); return 9u; }
  if (avail >= 10u && !(w2 & 0x8000u)) { out = x | ((u64)(w2 & 0x7Fu) << 56) | ((u64)((w2 >> 8) & 1u) << 63); return 10u; }
  return 0u;
}
template <class MP>
__device__ __forceinline__ bool w_varint(MP m, WSpan& s, u64& out) {
  if (s.p >= s.e) return false;
  const u64 w = w_peek8(m, s.p);
  if (!(w & 0x80u)) { out = w & 0x7Fu; ++s.p; return true; }
  u32 n = w_varint_win(w, out);
  if (!n) n = w_varint_long(m, s.p, s.e - s.p, w, out);
  if (!n || n > s.e - s.p) return false;
  s.p += n;
  return true;
}

// Next field of a message; false at the end or on malformed input (`bad` set).  One window holds the tag and, for the two wire
// types almost every field has, what follows it - the length of a length-delimited field, a varint value.
template <class MP>
__device__ __forceinline__ bool w_next(MP m, WSpan& s, WField& f, bool& bad) {
  if (s.p >= s.e) return false;
  u64 w = w_peek8(m, s.p);
  u32 left = 8u;     // bytes of the window not yet used
  u64 key;
  if (!(w & 0x80u)) { key = w & 0x7Fu; ++s.p; w >>= 8; left = 7u; }
  else {
    const u32 n = w_varint_win(w, key);
    if (n && n <= s.e - s.p && n < 8u) { s.p += n; w >>= 8u * n; left = 8u - n; }
    else { if (!w_varint(m, s, key)) { bad = true; return false; } left = 0u; }
  }
  f.num = (u32)(key >> 3); f.wt = (u32)(key & 7u);
  if (f.wt == 2u || f.wt == 0u) {
    u64 v; bool got = false;
    if (left && s.p < s.e) {
      if (!(w & 0x80u)) { v = w & 0x7Fu; ++s.p; got = true; }
      else {
        // (the window's unused top bytes read as zero after the shift: a stop bit found there is not the varint's)
        const u32 n = w_varint_win(w | (~0ull << (8u * left)) , v);
        if (n && n <= left) { if (n > s.e - s.p) { bad = true; return false; } s.p += n; got = true; }
      }
    }
    if (!got && !w_varint(m, s, v)) { bad = true; return false; }
    if (f.wt == 0u) { f.v = v; return true; }
    if (v > (u64)(s.e - s.p)) { bad = true; return false; }
    f.s.p = s.p; f.s.e = s.p + (u32)v; s.p += (u32)v;
    return true;
  }
  if (f.wt == 1u) {
    if (s.e - s.p < 8u) { bad = true; return false; }
    f.v = w_peek8(m, s.p); s.p += 8u; return true;
  }
  if (f.wt == 5u) {
    if (s.e - s.p < 4u) { bad = true; return false; }
    f.v = (u32)w_peek8(m, s.p); s.p += 4u; return true;
  }
  bad = true; return false;
}

// map<string, google.protobuf.Value> entry: key = 1, value = 2 (last of each wins)
template <class MP>
__device__ __forceinline__ bool w_entry(MP m, WSpan e, WSpan& key, WSpan& val, bool& bad) {
  key.p = key.e = 0; val.p = val.e = 0;
  WField f;
  while (w_next(m, e, f, bad)) {
    if (f.num == 1u && f.wt == 2u) key = f.s;
    else if (f.num == 2u && f.wt == 2u) val = f.s;
  }
  return !bad;
}

// google.protobuf.Value oneof: null 1, number 2 (double), string 3, bool 4, struct 5, list 6; a field counts only with the
// wire type its declaration has; the last one present wins; an empty message is null.
template <class MP>
__device__ __forceinline__ bool w_value(MP m, WSpan s, WVal& out, bool& bad) {
  out.kind = 1u; out.v = 0; out.s.p = out.s.e = 0;
  WField f;
  while (w_next(m, s, f, bad)) {
    if (f.num >= 1u && f.num <= 6u) {
      const u32 want = (f.num == 2u) ? 1u : (f.num == 1u || f.num == 4u) ? 0u : 2u;
      if (f.wt == want) {
        out.kind = f.num;
        if (f.wt == 2u) { out.v = 0; out.s = f.s; } else { out.v = f.v; out.s.p = out.s.e = 0; }
      }
    }
  }
  return !bad;
}

// n bytes at a and at b (either in LDS or in global memory; both end in slack): eight at a time, all of them fetched before the
// first is looked at
template <class PA, class PB>
__device__ __forceinline__ bool w_bytes_eq(PA a, PB b, u32 n) {
  u64 diff = 0;
  u32 i = 0;
  for (; i + 8u <= n; i += 8u) diff |= w_peek8(a, i) ^ w_peek8(b, i);
  if (i < n) diff |= (w_peek8(a, i) ^ w_peek8(b, i)) & (~0ull >> (64u - 8u * (n - i)));
  return diff == 0;
}
// the string hash of cbh_wire_hash (below), over a message's bytes: eight per round trip
template <class MP>
__device__ __forceinline__ u32 w_hash(MP s, u32 n) {
  u32 h = 0x811C9DC5u ^ n;
  for (u32 i = 0; i < n; i += 8u) {
    u64 w = w_peek8(s, i);
    const u32 k = n - i < 8u ? n - i : 8u;
    for (u32 j = 0; j < k; ++j) { h ^= (u32)w & 0xFFu; h *= 0x01000193u; w >>= 8; }
  }
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

// Looks `key` up in the map field `fnum` of `msg` (last entry wins, as protobuf maps decode).
template <class MP>
__device__ __forceinline__ bool w_map_get(MP m, WSpan msg, u32 fnum, const CBH_G u8* key, u32 klen, WSpan& val, bool& bad) {
  WField f; bool found = false;
  while (w_next(m, msg, f, bad)) {
    if (f.num != fnum || f.wt != 2u) continue;
    WSpan k, v;
    if (!w_entry(m, f.s, k, v, bad)) return false;
    if (k.e - k.p == klen && w_bytes_eq(m + k.p, key, klen)) { val = v; found = true; }
  }
  return found;
}

// ---- interning -------------------------------------------------------------------------------------------------
// What the out-of-line helpers read of the launch's arguments, BY VALUE (registers): a reference to WireArgs would make the compiler
// keep the whole argument block in scratch memory and turn every a.field of the kernel into a scratch load.
struct WTab {
  const CBH_G u64* tix; const CBH_G u8* t_str_bytes; CBH_G u64* lix; CBH_G u32* lflags; const CBH_G u8* msg;
  u32 tix_mask, lix_mask, K;
};
__device__ __forceinline__ WTab w_tab(const WireArgs& a) {
  WTab t; t.tix = a.tix; t.t_str_bytes = a.t_str_bytes; t.lix = a.lix; t.lflags = a.lflags; t.msg = a.msg;
  t.tix_mask = a.tix_mask; t.lix_mask = a.lix_mask; t.K = a.K;
  return t;
}
#define CBH_WL_BAD 1u
#define CBH_WL_HOST 2u
#define CBH_WL_DICT_FULL 4u

// The table's index: open addressing over 16-byte slots {hash << 32 | id + 1, offset << 32 | length} (0: empty) - one load says
// whether a slot can be the string and where its bytes are, a second round trip compares them.
// id of a table string, CBH_NONE if the table does not hold it
template <class MP>
__device__ __forceinline__ u32 w_table_sid(const WTab& t, MP s, u32 len, u32 h) {
  typedef u64 v2 __attribute__((ext_vector_type(2)));
  for (u32 i = h & t.tix_mask, n = 0; n <= t.tix_mask; i = (i + 1u) & t.tix_mask, ++n) {
#ifndef CBH_HOSTSIM
    const v2 e = *(const CBH_G v2*)(t.tix + 2u * (size_t)i);
    const u64 e0 = e.x, e1 = e.y;
#else
    const u64 e0 = t.tix[2u * (size_t)i], e1 = t.tix[2u * (size_t)i + 1u];
#endif
    if (!e0) return CBH_NONE;
    if ((u32)(e0 >> 32) == h && (u32)e1 == len && w_bytes_eq(s, t.t_str_bytes + (u32)(e1 >> 32), len)) return (u32)e0 - 1u;
  }
  return CBH_NONE;
}

struct WLane { bool bad, host, dict_full; };

// string id of msg[off .. off + len): the table's id, or K + its slot in the batch-local dictionary (claimed if absent); in the
// high word CBH_WL_* flags.  `m[off]`: the string; `bias`: what to add to `off` for its offset in WireArgs.msg (the staged view of
// the fill kernel starts at the wave's block, not at the buffer)
template <class MP>
__device__ __attribute__((noinline)) u64 w_intern_fn(WTab t, MP m, u32 bias, u32 off, u32 len, u32 flag) {
  MP s = m + off;
  const u32 h = w_hash(s, len);
  const u32 id = w_table_sid(t, s, len, h);
  if (id != CBH_NONE) return id;
  if (len > CBH_WIRE_MAX_STRLEN) return (u64)t.K | ((u64)CBH_WL_HOST << 32);
  const u64 key = ((u64)((h >> 16) | 0x8000u) << 48) | ((u64)len << 32) | (u64)(off + bias);
  u32 i = h & t.lix_mask;
  for (u32 n = 0; n < CBH_WIRE_MAX_PROBES; ++n, i = (i + 1u) & t.lix_mask) {
    u64 cur = w_load64(t.lix + i);
    if (cur == 0) { const u64 prev = w_cas64(t.lix + i, 0, key); cur = prev == 0 ? key : prev; }
    if (cur == key || ((cur >> 32) == (key >> 32) && w_bytes_eq(s, t.msg + (u32)cur, len))) {
      if (flag) {
        const u32 sh = (i & 3u) * 8u;
        if (((t.lflags[i >> 2] >> sh) & flag) != flag) w_or32(t.lflags + (i >> 2), flag << sh);
      }
      return (u64)(t.K + i);
    }
  }
  return (u64)t.K | ((u64)CBH_WL_DICT_FULL << 32);
}
__device__ __forceinline__ void w_lane_flags(WLane& L, u32 fl) {
  if (fl & CBH_WL_BAD) L.bad = true;
  if (fl & CBH_WL_HOST) L.host = true;
  if (fl & CBH_WL_DICT_FULL) L.dict_full = true;
}
template <class MP>
__device__ __forceinline__ u32 w_intern(const WireArgs& a, MP m, u32 bias, u32 off, u32 len, u32 flag, WLane& L) {
  const u64 r = w_intern_fn(w_tab(a), m, bias, off, len, flag);
  w_lane_flags(L, (u32)(r >> 32));
  return (u32)r;
}

// scope word of a scope string (cbh_ingest.cpp scope_word, namer.go:77-87): the scope itself if the table knows it (bit 31
// set), else its nearest ancestor "a.b.c" -> "a.b", "a", "" the table knows, else 0
template <class MP>
__device__ __attribute__((noinline)) u32 w_scope_word_fn(WTab t, const CBH_G u32* scope_of_sid, MP m, u32 off, u32 len) {
  MP s = m + off;
  u32 id = w_table_sid(t, s, len, w_hash(s, len));
  if (id != CBH_NONE && scope_of_sid[id] != CBH_NONE) return scope_of_sid[id] | CBH_SCOPE_EXACT;
  for (u32 i = len; i-- > 0u;) {
    if (s[i] == '.' || i == 0u) {
      id = w_table_sid(t, s, i, w_hash(s, i));
      if (id != CBH_NONE && scope_of_sid[id] != CBH_NONE) return scope_of_sid[id];
    }
  }
  return 0u;
}
template <class MP>
__device__ __forceinline__ u32 w_scope_word(const WireArgs& a, MP m, u32 off, u32 len) { return w_scope_word_fn(w_tab(a), a.scope_of_sid, m, off, len); }

// namer.go:213-218 (cbh_ingest.cpp sanitize): does this resource kind have to be rewritten?  A name of the pre-0.30 form
// (segments "[A-Za-z][0-9A-Za-z_@.\-/]*" joined by ':') has every run of characters outside [0-9A-Za-z_.] replaced by one '_'.
__device__ __forceinline__ bool w_kind_ok_char(u32 c) { return (c - '0' < 10u) || ((c | 0x20u) - 'a' < 26u) || c == '_' || c == '.'; }
// (the common answer - every character is of [0-9A-Za-z_.] - from eight bytes per round trip)
template <class P>
__device__ __forceinline__ bool w_kind_is_plain(P s, u32 len) {
  bool plain = true;
  for (u32 i = 0; i < len && plain; i += 8u) {
    u64 w = w_peek8(s, i);
    const u32 k = len - i < 8u ? len - i : 8u;
    for (u32 j = 0; j < k; ++j) { plain = plain && w_kind_ok_char((u32)w & 0xFFu); w >>= 8; }
  }
  return plain;
}
template <class P>
__device__ __forceinline__ bool w_kind_needs_rewrite(P s, u32 len) {
  if (len == 0u || w_kind_is_plain(s, len)) return false;
  bool seg_start = true;
  for (u32 i = 0; i < len; ++i) {
    const u32 c = s[i];
    const bool alpha = ((c | 0x20u) - 'a' < 26u);
    if (seg_start) { if (!alpha) return false; seg_start = false; }
    else if (c == ':') seg_start = true;
    else if (!((c - '0' < 10u) || alpha || c == '_' || c == '@' || c == '.' || c == '-' || c == '/')) return false;
  }
  return !seg_start;
}
// The rewritten kind exists nowhere in the message, so only the table can name it: its id if the table holds the rewritten
// string (a kind some policy is written for), else CBH_NONE - that message is the host flattener's.
template <class MP>
__device__ __attribute__((noinline)) u32 w_rewritten_kind_sid_fn(WTab a, MP s, u32 len) {
  u32 n = 0; bool in_run = false;
  for (u32 i = 0; i < len; ++i) { const bool ok = w_kind_ok_char(s[i]); n += (ok || !in_run); in_run = !ok; }
  u32 h = 0x811C9DC5u ^ n; in_run = false;   // cbh_wire_hash over the rewritten bytes
  for (u32 i = 0; i < len; ++i) {
    const u32 c = s[i]; const bool ok = w_kind_ok_char(c);
    if (ok || !in_run) { h ^= ok ? c : (u32)'_'; h *= 0x01000193u; }
    in_run = !ok;
  }
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  for (u32 i = h & a.tix_mask, probes = 0; probes <= a.tix_mask; i = (i + 1u) & a.tix_mask, ++probes) {
    const u64 e0 = a.tix[2u * (size_t)i], e1 = a.tix[2u * (size_t)i + 1u];
    if (!e0) return CBH_NONE;
    if ((u32)(e0 >> 32) != h || (u32)e1 != n) continue;
    const u32 o = (u32)(e1 >> 32);
    bool same = true; u32 k = 0; in_run = false;
    for (u32 j = 0; j < len && same; ++j) {
      const u32 c = s[j]; const bool ok = w_kind_ok_char(c);
      if (ok || !in_run) { same = a.t_str_bytes[o + k] == (ok ? c : (u32)'_'); ++k; }
      in_run = !ok;
    }
    if (same) return (u32)e0 - 1u;
  }
  return CBH_NONE;
}
template <class MP>
__device__ __forceinline__ u32 w_rewritten_kind_sid(const WireArgs& a, MP s, u32 len) { return w_rewritten_kind_sid_fn(w_tab(a), s, len); }

// ---- wave helpers ------------------------------------------------------------------------------------------------
// exclusive prefix and total of a small per-lane count, by bit planes (reached by all 64 lanes)
__device__ __forceinline__ u32 w_wave_prefix(u32 x, u32 bits, u32 lane, u32& total) {
  u32 pre = 0; total = 0;
  const u64 lt = (1ull << lane) - 1ull;
  for (u32 b = 0; b < bits; ++b) {
    const u64 mk = wave_ballot(((x >> b) & 1u) != 0u);
    pre += (u32)__builtin_popcountll(mk & lt) << b;
    total += (u32)__builtin_popcountll(mk) << b;
  }
  return pre;
}
__device__ __forceinline__ u32 w_wave_max(u32 x, u32 bits) {   // largest x of the wave
  u32 best = 0; bool in = true;
  for (u32 b = bits; b-- > 0u;) {
    const u64 mk = wave_ballot(in && ((x >> b) & 1u));
    if (mk) { best |= 1u << b; in = in && ((x >> b) & 1u); }
  }
  return best;
}

#ifndef CBH_HOSTSIM
// (aligned(4): some of these accesses are only 8-byte aligned - a lane's run of per-wave records starts at an odd record when `per`
// is odd.  The 16-byte instruction is the same; the type must not promise the compiler an alignment the address does not have.)
typedef u32 w_v4 __attribute__((ext_vector_type(4), aligned(4)));
#define W_LOAD4(dst, p) { const w_v4 t_ = *(const CBH_G w_v4*)(p); (dst)[0] = t_.x; (dst)[1] = t_.y; (dst)[2] = t_.z; (dst)[3] = t_.w; }
#define W_STORE4(p, src) { w_v4 t_; t_.x = (src)[0]; t_.y = (src)[1]; t_.z = (src)[2]; t_.w = (src)[3]; *(CBH_G w_v4*)(p) = t_; }
#else
#define W_LOAD4(dst, p) { for (u32 q_ = 0; q_ < 4u; ++q_) (dst)[q_] = (p)[q_]; }
#define W_STORE4(p, src) { for (u32 q_ = 0; q_ < 4u; ++q_) (p)[q_] = (src)[q_]; }
#endif
// The wave's block of messages src[0 .. need) -> LDS at `stage`: asynchronous 16-byte copies straight into LDS (global_load_lds_dwordx4:
// destination = uniform base + lane * 16, no staging registers), every one of them in flight before the first has landed - a loop of
// load / store pairs waits for HBM once per kilobyte, and these bytes have just come up over the link: none of them is in L2.
__device__ __forceinline__ void w_stage_block(CBH_L u8* stage, const CBH_G u8* src, u32 need, u32 lane) {
#ifndef CBH_HOSTSIM
  for (u32 o = 0; o < need; o += 1024u)
    if (o + lane * 16u < need) __builtin_amdgcn_global_load_lds((const CBH_G void*)(src + o + lane * 16u), (CBH_L void*)(stage + o), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  for (u32 o = lane * 16u; o < need; o += 64u * 16u) for (u32 k = 0; k < 16u; ++k) stage[o + k] = src[o + k];
#endif
}

// ---- kernel 1: counts ----------------------------------------------------------------------------------------------
template <class MP>
__device__ __forceinline__ void w_top_level(MP m, WSpan s, WSpan& resource, WSpan& principal, WSpan& aux, WSpan& request_id,
                                            u32& n_actions, bool& bad) {
  resource.p = resource.e = principal.p = principal.e = aux.p = aux.e = request_id.p = request_id.e = 0; n_actions = 0;
  WField f;
  while (w_next(m, s, f, bad)) {
    if (f.wt != 2u) continue;
    if (f.num == 2u) resource = f.s; else if (f.num == 3u) principal = f.s; else if (f.num == 4u) ++n_actions;
    else if (f.num == 5u) aux = f.s; else if (f.num == 1u) request_id = f.s;
  }
}

// The count of one lane's message, read at m[o0 - bias .. o1 - bias)
template <class MP>
__device__ __forceinline__ void w_count_message(MP m, u32 p0, u32 p1, u32& na, u32& nr, bool& bad) {
  WSpan s; s.p = p0; s.e = p1;
  WSpan resource, principal, aux, rid;
  w_top_level(m, s, resource, principal, aux, rid, na, bad);
  WField f; WSpan p = principal;
  while (w_next(m, p, f, bad)) nr += (f.num == 3u && f.wt == 2u);
}

// (dynamic LDS: WireArgs.lds_cap bytes - a wave whose 64 messages fit stages them with one coalesced copy and counts there: the
// messages have just come up over the link, every first touch of a line is a trip to HBM, and the walk's loads depend on each other)
#ifdef CBH_HOSTSIM
static void cbh_wire_count_kernel(WireArgs a)
#else
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_count_kernel(WireArgs a)
#endif
{
  const u32 lane = threadIdx.x & 63u;
  const u32 i = blockIdx.x * CBH_BLOCK + threadIdx.x;
  const bool live = i < a.n;
  u32 na = 0, nr = 0, st = CBH_WS_OK;
  const u32 w0 = blockIdx.x * CBH_BLOCK, w1 = (w0 + CBH_BLOCK < a.n) ? w0 + CBH_BLOCK : a.n;
  const u64 lo64 = a.moff[w0 < a.n ? w0 : a.n], hi64 = a.moff[w1];
  const u32 lo16 = (u32)lo64 & ~15u;
  const bool staged = hi64 > lo64 && hi64 <= (u64)a.dver_off && (u64)((u32)hi64 - lo16) + 16u + CBH_WIRE_SLACK <= (u64)a.lds_cap;   // (uniform)
  if (staged) {
    const u32 need = (u32)hi64 - lo16;
    w_stage_block((CBH_L u8*)cbh_dyn_lds, a.msg + lo16, need, lane);
    __syncthreads();
  }
  if (live) {
    const u64 o0 = a.moff[i], o1 = a.moff[i + 1u];
    bool bad = o1 < o0 || o1 > (u64)a.dver_off;   // (dver_off = the end of the messages: nothing may point past it)
    if (!bad) {
      if (staged && o0 >= lo64 && o1 <= hi64) w_count_message((WLds)cbh_dyn_lds, (u32)o0 - lo16, (u32)o1 - lo16, na, nr, bad);
      else w_count_message((WGlob)a.msg, (u32)o0, (u32)o1, na, nr, bad);
    }
    if (bad) st = CBH_WS_BAD;
    else if (na > CBH_MAX_ACTIONS_PER_REQUEST || nr > CBH_WIRE_MAX_ROLES) st = CBH_WS_HOST;
    if (st != CBH_WS_OK) { na = 0; nr = 0; }
    a.cnt[i] = na | (nr << 8);
    a.status[i] = (u8)st;
  }
  u32 ta, tr;
  (void)w_wave_prefix(na, 7u, lane, ta);
  (void)w_wave_prefix(nr, 8u, lane, tr);
  const u32 wmax_a = w_wave_max(na, 7u), wmax_r = w_wave_max(nr, 8u);
  const u64 wide = wave_ballot(live && cbh_is_wide(na, nr));
  const u64 badm = wave_ballot(st == CBH_WS_BAD), hostm = wave_ballot(st == CBH_WS_HOST);
  if (lane == 0u) {
    const u32 w = blockIdx.x * (CBH_BLOCK / 64u) + threadIdx.x / 64u;
    // what the fill kernel would stage for this wave (offsets out of order: the call fails on them anyway)
    const u64 blk = hi64 >= lo64 ? hi64 - (lo64 & ~15ull) : 0xFFFFFFFFull;
    const u32 blk32 = blk > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)blk;
#ifndef CBH_HOSTSIM
    w_v4 rec; rec.x = ta; rec.y = tr; rec.z = wmax_a | (wmax_r << 16); rec.w = blk32;
    *(CBH_G w_v4*)(a.wavesum + 4u * w) = rec;
#else
    a.wavesum[4u * w] = ta; a.wavesum[4u * w + 1u] = tr; a.wavesum[4u * w + 2u] = wmax_a | (wmax_r << 16); a.wavesum[4u * w + 3u] = blk32;
#endif
    const u32 base = blockIdx.x * CBH_BLOCK + (threadIdx.x & ~63u);
    if (wide) {
      w_min32(&a.stats->wide_lo, base + (u32)__builtin_ctzll(wide));
      w_max32(&a.stats->wide_hi, base + 64u - (u32)__builtin_clzll(wide));
    }
    if (badm) w_min32(&a.stats->first_bad, base + (u32)__builtin_ctzll(badm));
    if (hostm) (void)w_add32(&a.stats->n_host, (u32)__builtin_popcountll(hostm));
  }
}

// ---- kernel 2: offsets of the waves' slices, totals, the call's default strings (one workgroup of CBH_WIRE_SCAN_THREADS) --------
// Every thread takes a run of consecutive waves; the runs' sums meet in LDS, thread 0 .. 63 of the first wave add them up by bit planes.
// (a lane's run is read and written sixteen pairs per round trip: one wave, so what it costs is how often it waits)
#define CBH_WIRE_SCAN_THREADS 64u
#ifdef CBH_HOSTSIM
static void cbh_wire_scan_kernel(WireArgs a)
#else
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_scan_kernel(WireArgs a)
#endif
{
  const u32 lane = threadIdx.x & 63u;
  const u32 nw = (a.n + 63u) / 64u;
  const u32 per = (nw + 63u) / 64u;
  const u32 lo = lane * per < nw ? lane * per : nw, hi = (lo + per < nw) ? lo + per : nw;
  u32 sa = 0, sr = 0, mx_a = 0, mx_r = 0, mx_blk = 0;
  {
    u32 w = lo;
    for (; w + 8u <= hi; w += 8u) {
      u32 x[32];
#pragma unroll
      for (u32 j = 0; j < 8u; ++j) W_LOAD4(x + 4u * j, a.wavesum + 4u * (w + j));
#pragma unroll
      for (u32 j = 0; j < 8u; ++j) {
        sa += x[4u * j]; sr += x[4u * j + 1u];
        const u32 ma = x[4u * j + 2u] & 0xFFFFu, mr = x[4u * j + 2u] >> 16, bk = x[4u * j + 3u];
        mx_a = ma > mx_a ? ma : mx_a; mx_r = mr > mx_r ? mr : mx_r; mx_blk = bk > mx_blk ? bk : mx_blk;
      }
    }
    for (; w < hi; ++w) {
      u32 x[4]; W_LOAD4(x, a.wavesum + 4u * w);
      sa += x[0]; sr += x[1];
      const u32 ma = x[2] & 0xFFFFu, mr = x[2] >> 16;
      mx_a = ma > mx_a ? ma : mx_a; mx_r = mr > mx_r ? mr : mx_r; mx_blk = x[3] > mx_blk ? x[3] : mx_blk;
    }
  }
  mx_a = w_wave_max(mx_a, 8u); mx_r = w_wave_max(mx_r, 9u); mx_blk = w_wave_max(mx_blk, 32u);
  u32 ta, tr;
  u32 pa = w_wave_prefix(sa, 32u, lane, ta), pr = w_wave_prefix(sr, 32u, lane, tr);
  {
    u32 w = lo;
    for (; w + 8u <= hi; w += 8u) {
      u32 x[32], y[16];
#pragma unroll
      for (u32 j = 0; j < 8u; ++j) W_LOAD4(x + 4u * j, a.wavesum + 4u * (w + j));
#pragma unroll
      for (u32 j = 0; j < 8u; ++j) { y[2u * j] = pa; y[2u * j + 1u] = pr; pa += x[4u * j]; pr += x[4u * j + 1u]; }
#pragma unroll
      for (u32 j = 0; j < 4u; ++j) W_STORE4(a.waveoff + 2u * w + 4u * j, y + 4u * j);
    }
    for (; w < hi; ++w) {
      a.waveoff[2u * w] = pa; a.waveoff[2u * w + 1u] = pr;
      pa += a.wavesum[4u * w]; pr += a.wavesum[4u * w + 1u];
    }
  }
  if (lane == 0u) {
    a.stats->n_tuples = ta; a.stats->n_roles = tr;
    a.stats->max_actions = mx_a; a.stats->max_roles = mx_r; a.stats->max_block = mx_blk;
    WLane L; L.bad = false; L.host = false; L.dict_full = false;
    WGlob m = (WGlob)a.msg; const u32 bias = 0u;
    const u32 v_empty = w_intern(a, m, bias, a.dver_off, 0u, 0u, L);
    const u32 v_dver = w_intern(a, m, bias, a.dver_off, a.dver_len, 0u, L);
    const u32 v_dscope = w_scope_word(a, m, a.dscope_off, a.dscope_len);
    const u32 v_claims = w_intern(a, m, bias, a.claims_off, 6u, 0u, L);
    a.stats->sid_empty = v_empty; a.stats->sid_dver = v_dver; a.stats->dscope_word = v_dscope; a.stats->sid_claims = v_claims;
    if (L.dict_full) w_or32(&a.stats->flags, CBH_WF_DICT_FULL);
    // what the host waits for, straight into its page-locked block (no copy engine, no further launch): the statistics as they stand -
    // the count kernel's from L2 (its launch has ended), this lane's own from its registers
    if (a.host_stats) {
      const CBH_G u32* src = (const CBH_G u32*)a.stats; CBH_G u32* dst = (CBH_G u32*)a.host_stats;
      u32 fl = 0;
      for (u32 k = 0; k < (u32)(sizeof(WireStats) / 4u); ++k) { const u32 v = w_load32(src + k); if (k == 9u) fl = v; dst[k] = v; }
      static_assert(offsetof(WireStats, flags) == 36, "word 9 of WireStats is `flags`");
      a.host_stats->n_tuples = ta; a.host_stats->n_roles = tr;
      a.host_stats->max_actions = mx_a; a.host_stats->max_roles = mx_r; a.host_stats->max_block = mx_blk;
      a.host_stats->sid_empty = v_empty; a.host_stats->sid_dver = v_dver; a.host_stats->dscope_word = v_dscope; a.host_stats->sid_claims = v_claims;
      a.host_stats->flags = fl | (L.dict_full ? CBH_WF_DICT_FULL : 0u);
    }
  }
}

// ---- kernel 3: the batch ---------------------------------------------------------------------------------------------
struct WFrame { WSpan rest; u32 slot; u32 is_map; };

template <class MP>
__device__ __forceinline__ u32 w_count_fields(MP m, WSpan s, u32 fnum, bool& bad) {
  u32 n = 0; WField f;
  while (w_next(m, s, f, bad)) n += (f.num == fnum && f.wt == 2u);
  return n;
}
__device__ __forceinline__ u64 w_container(u32 off, u32 n) { return ((u64)CBH_HEAP_BATCH << 62) | ((u64)off << 32) | (u64)n; }

// One container value - the entries of map field `fnum` of `body`, or the values (field 1) of a ListValue - and everything
// nested in it, depth first.  WRITE = false: returns the heap entries it needs.  WRITE = true: writes them at [base, ..)
// (the container's own entries first, each nested container's behind what was allocated before it) and returns the same.
struct WHeapDst { CBH_G u8* heap_tag; CBH_G u64* heap_val; u32 heap_cap; };
template <bool WRITE, class MP>
__device__ __attribute__((noinline)) u64 w_container_walk_fn(WTab t, WHeapDst a, MP m, u32 bias, WSpan body, u32 fnum, bool is_map, u32 base) {
  WLane L; L.bad = false; L.host = false; L.dict_full = false;
  auto intern = [&](u32 off, u32 len) -> u32 { const u64 r = w_intern_fn(t, m, bias, off, len, 0u); w_lane_flags(L, (u32)(r >> 32)); return (u32)r; };
  WFrame st[CBH_WIRE_MAX_DEPTH];
  u32 depth = 0;
  const u32 n0 = w_count_fields(m, body, fnum, L.bad);
  u32 used = is_map ? 2u * n0 : n0;
  st[0].rest = body; st[0].slot = base; st[0].is_map = is_map ? (0x80000000u | fnum) : fnum; depth = 1;
  while (depth && !L.bad) {
    WFrame& fr = st[depth - 1u];
    const u32 want = fr.is_map & 0x7FFFFFFFu; const bool mp = (fr.is_map >> 31) != 0u;
    WField f; bool got = false;
    while (w_next(m, fr.rest, f, L.bad)) if (f.num == want && f.wt == 2u) { got = true; break; }
    if (!got) { --depth; continue; }
    WSpan elem = f.s;
    if (mp) {
      WSpan k, v;
      if (!w_entry(m, f.s, k, v, L.bad)) break;
      const u32 kid = WRITE ? intern(k.p, k.e - k.p) : 0u;   // (the counting pass leaves the dictionary alone)
      if (WRITE && fr.slot < a.heap_cap) { a.heap_tag[fr.slot] = (u8)CBH_T_STRING; a.heap_val[fr.slot] = kid; }
      ++fr.slot;
      elem = v;
    }
    WVal v;
    u32 tag = CBH_T_NULL; u64 val = 0;
    if (w_value(m, elem, v, L.bad)) {
      if (v.kind == 2u) { tag = CBH_T_DOUBLE; val = v.v; }
      else if (v.kind == 3u) { tag = CBH_T_STRING; val = WRITE ? intern(v.s.p, v.s.e - v.s.p) : 0u; }
      else if (v.kind == 4u) { tag = CBH_T_BOOL; val = v.v ? 1u : 0u; }
      else if (v.kind >= 5u) {
        const bool cm = v.kind == 5u;
        const u32 n2 = w_count_fields(m, v.s, 1u, L.bad);
        const u32 off2 = base + used;
        used += cm ? 2u * n2 : n2;
        tag = cm ? CBH_T_MAP : CBH_T_LIST; val = w_container(off2, n2);
        if (depth == CBH_WIRE_MAX_DEPTH) { L.host = true; tag = CBH_T_NULL; val = 0; }
        else if (n2) {
          const u32 slot = fr.slot;
          if (WRITE && slot < a.heap_cap) { a.heap_tag[slot] = (u8)tag; a.heap_val[slot] = val; }
          ++st[depth - 1u].slot;
          st[depth].rest = v.s; st[depth].slot = off2; st[depth].is_map = cm ? (0x80000000u | 1u) : 1u;
          ++depth;
          if (used > CBH_WIRE_MAX_VALUE_ENTRIES) { L.host = true; break; }
          continue;
        }
      }
    }
    if (WRITE && fr.slot < a.heap_cap) { a.heap_tag[fr.slot] = (u8)tag; a.heap_val[fr.slot] = val; }
    ++fr.slot;
  }
  return (u64)used | ((u64)((L.bad ? CBH_WL_BAD : 0u) | (L.host ? CBH_WL_HOST : 0u) | (L.dict_full ? CBH_WL_DICT_FULL : 0u)) << 32);
}
template <bool WRITE, class MP>
__device__ __forceinline__ u32 w_container_walk(const WireArgs& a, MP m, u32 bias, WSpan body, u32 fnum, bool is_map, u32 base, WLane& L) {
  WHeapDst hd; hd.heap_tag = a.heap_tag; hd.heap_val = a.heap_val; hd.heap_cap = a.heap_cap;
  const u64 r = w_container_walk_fn<WRITE>(w_tab(a), hd, m, bias, body, fnum, is_map, base);
  w_lane_flags(L, (u32)(r >> 32));
  return (u32)r;
}


__device__ __forceinline__ u64 w_mix64(u64 h, u64 v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); h *= 0xFF51AFD7ED558CCDull; return h ^ (h >> 29); }
// the fingerprint of a request's route: what lanes must share to walk a bucket together (cbh_wire_route_kernel)
__device__ __forceinline__ u64 w_route_hash(u32 kind, u32 ver, u32 scope) {
  u64 h = w_mix64(0x243F6A8885A308D3ull, ((u64)kind << 32) | ver);
  h = w_mix64(h, (u64)scope);
  return h ? h : 1ull;
}
// What a lane knows about its message after the first passes, shared by the passes over the attribute columns.
struct WMsgParts { WSpan resource, principal, aux; };

// The fill for one lane (and, where the heap is reserved, for its wave).  MP: where the wave's messages are read - the block staged
// in LDS (m[0] = byte `bias` of WireArgs.msg) or the message buffer itself (bias 0).  The call's globals lie behind the messages and
// are always read in place (`mg`).  cur_p / cur_e: [CBH_WIRE_CUR_COLS][CBH_BLOCK] words of LDS.
template <class MP>
__device__ __forceinline__ void w_fill_body(const WireArgs& a, MP m, const u32 bias, bool live, const bool in_range, const u32 lane, const u32 i,
                                            CBH_L u32* cur_p, CBH_L u32* cur_e) {
  const u32 N = a.n;
  const WGlob mg = (WGlob)a.msg;
  const u32 c0 = in_range ? a.cnt[i] : 0u;
  const u32 na = live ? (c0 & 0xFFu) : 0u, nr = live ? (c0 >> 8) : 0u;
  u32 ta, tr;
  const u32 wv = blockIdx.x * (CBH_BLOCK / 64u) + threadIdx.x / 64u;
  const u32 act_off = a.waveoff[2u * wv] + w_wave_prefix(na, 7u, lane, ta);
  const u32 role_off = a.waveoff[2u * wv + 1u] + w_wave_prefix(nr, 8u, lane, tr);
  WLane L; L.bad = false; L.host = false; L.dict_full = false;
  WSpan resource, principal, aux, rid;
  resource.p = resource.e = principal.p = principal.e = aux.p = aux.e = rid.p = rid.e = 0;
  const u32 sid_empty = a.stats->sid_empty;
  u32 rt_kind = 0, rt_ver = 0, rt_scope = 0;   // the request's route (kind, resource version, resource scope word)
#define W_RQ(f) a.req_u32[(size_t)(f) * N + i]
#define W_SID(sp, fl) (((sp).e == (sp).p) ? sid_empty : w_intern(a, m, bias, (sp).p, (sp).e - (sp).p, (fl), L))
  if (live) {
    const u32 base0 = (u32)a.moff[i] - bias;
    WSpan s; s.p = base0; s.e = (u32)a.moff[i + 1u] - bias;
    WSpan pid, pver, pscope, kind, rver, rrid, rscope;
    pid.p = pid.e = pver.p = pver.e = pscope.p = pscope.e = kind.p = kind.e = rver.p = rver.e = rrid.p = rrid.e = rscope.p = rscope.e = 0;
    {   // top level: resource 2, principal 3, actions 4, aux_data 5, request_id 1
      WField f; u32 k = 0;
      while (w_next(m, s, f, L.bad)) {
        if (f.wt != 2u) continue;
        if (f.num == 2u) resource = f.s; else if (f.num == 3u) principal = f.s; else if (f.num == 5u) aux = f.s; else if (f.num == 1u) rid = f.s;
        else if (f.num == 4u) {
          if (k < na) {
            const u32 len = f.s.e - f.s.p;
            a.tuple_action[act_off + k] = w_intern(a, m, bias, f.s.p, len, CBH_SF_ACTION, L);   // (an empty action is a string like any other)
            a.act_span[2u * (act_off + k)] = len ? f.s.p - base0 : 0u; a.act_span[2u * (act_off + k) + 1u] = len;
          }
          ++k;
        }
      }
    }
    {   // Principal: id 1, policy_version 2, roles 3, attr 4, scope 5
      WSpan p = principal; WField f; u32 k = 0;
      while (w_next(m, p, f, L.bad)) {
        if (f.wt != 2u) continue;
        if (f.num == 1u) pid = f.s; else if (f.num == 2u) pver = f.s; else if (f.num == 5u) pscope = f.s;
        else if (f.num == 3u) { if (k < nr) a.roles[role_off + k] = w_intern(a, m, bias, f.s.p, f.s.e - f.s.p, CBH_SF_ROLE, L); ++k; }
      }
    }
    {   // Resource: kind 1, policy_version 2, id 3, attr 4, scope 5
      WSpan p = resource; WField f;
      while (w_next(m, p, f, L.bad)) {
        if (f.wt != 2u) continue;
        if (f.num == 1u) kind = f.s; else if (f.num == 2u) rver = f.s; else if (f.num == 3u) rrid = f.s; else if (f.num == 5u) rscope = f.s;
      }
    }
    {
      CBH_G u32* sp = a.in_span + (size_t)i * 2u * CBH_WSPAN_N;
#define W_SPAN(which, v) sp[2u * (which)] = ((v).e > (v).p) ? (v).p - base0 : 0u; sp[2u * (which) + 1u] = (v).e - (v).p
      W_SPAN(0, rid); W_SPAN(1, pid); W_SPAN(2, pver); W_SPAN(3, kind); W_SPAN(4, rver); W_SPAN(5, rrid);
#undef W_SPAN
    }
    // scope_value (namer.go:276-278): one leading '.' does not count
    WSpan psv = pscope, rsv = rscope;
    if (psv.e > psv.p && m[psv.p] == '.') ++psv.p;
    if (rsv.e > rsv.p && m[rsv.p] == '.') ++rsv.p;
    W_RQ(CBH_RQ_PRINCIPAL_ID) = W_SID(pid, 0u);
    W_RQ(CBH_RQ_P_SCOPE) = (pscope.e == pscope.p) ? a.stats->dscope_word : w_scope_word(a, m, psv.p, psv.e - psv.p);
    W_RQ(CBH_RQ_P_VERSION) = (pver.e == pver.p) ? a.stats->sid_dver : w_intern(a, m, bias, pver.p, pver.e - pver.p, 0u, L);
    if (w_kind_needs_rewrite(m + kind.p, kind.e - kind.p)) {
      rt_kind = w_rewritten_kind_sid(a, m + kind.p, kind.e - kind.p);
      if (rt_kind == CBH_NONE) L.host = true;
    } else rt_kind = w_intern(a, m, bias, kind.p, kind.e - kind.p, CBH_SF_KIND, L);
    rt_scope = (rscope.e == rscope.p) ? a.stats->dscope_word : w_scope_word(a, m, rsv.p, rsv.e - rsv.p);
    rt_ver = (rver.e == rver.p) ? a.stats->sid_dver : w_intern(a, m, bias, rver.p, rver.e - rver.p, 0u, L);
    W_RQ(CBH_RQ_KIND) = rt_kind; W_RQ(CBH_RQ_R_SCOPE) = rt_scope; W_RQ(CBH_RQ_R_VERSION) = rt_ver;
    W_RQ(CBH_RQ_ROLE_OFF) = role_off; W_RQ(CBH_RQ_ROLE_CNT) = nr;
    W_RQ(CBH_RQ_ACT_OFF) = act_off; W_RQ(CBH_RQ_ACT_CNT) = na;
    if (a.t_flags & CBH_MF_READS_REQUEST_STRINGS) {   // raw request strings only CEL programs read
      W_RQ(CBH_RQ_S_RESOURCE_ID) = W_SID(rrid, 0u);
      W_RQ(CBH_RQ_S_KIND) = W_SID(kind, 0u);
      W_RQ(CBH_RQ_S_P_SCOPE) = W_SID(psv, 0u);
      W_RQ(CBH_RQ_S_R_SCOPE) = W_SID(rsv, 0u);
      W_RQ(CBH_RQ_S_P_VERSION) = W_SID(pver, 0u);
      W_RQ(CBH_RQ_S_R_VERSION) = W_SID(rver, 0u);
    } else {
      for (u32 f = CBH_RQ_NCORE; f < CBH_RQ_NFIELDS; ++f) W_RQ(f) = 0u;
    }
  } else if (in_range) {
    for (u32 f = 0; f < CBH_RQ_NFIELDS; ++f) W_RQ(f) = 0u;
    W_RQ(CBH_RQ_ACT_OFF) = act_off; W_RQ(CBH_RQ_ROLE_OFF) = role_off;
    for (u32 k = 0; k < 2u * CBH_WSPAN_N; ++k) a.in_span[(size_t)i * 2u * CBH_WSPAN_N + k] = 0u;
  }
  // The first keys of the columns, found in ONE pass per root over its attribute map (Principal.attr, Resource.attr, AuxData.jwt,
  // the call's globals) instead of one pass per column: where column c's first key has its value in this lane's message
  // (last entry wins), or CBH_NONE.  The first CBH_WIRE_CUR_COLS columns; the rest (and auxData.jwts paths) walk on their own.
  // A key is compared by its hash first (WireCol.key_hash: cbh_wire_hash, at table load), its bytes only where that agrees.
  const u32 ncc = a.n_cols < CBH_WIRE_CUR_COLS ? a.n_cols : CBH_WIRE_CUR_COLS;
  for (u32 c = 0; c < ncc; ++c) cur_p[c * CBH_BLOCK + threadIdx.x] = CBH_NONE;
  if (live) {
    for (u32 root = 0; root < 5u; ++root) {
      if (root == 3u) continue;
      bool any = false;
      for (u32 c = 0; c < ncc; ++c) any = any || (a.cols[c].root == root && a.cols[c].nk != 0u);   // (uniform)
      if (!any) continue;
      WSpan s2; s2.p = a.globals_off; s2.e = a.globals_off + a.globals_len;
      if (root == 0u) s2 = principal; else if (root == 1u) s2 = resource; else if (root == 2u) s2 = aux;
      const u32 want = (root == 2u || root == 4u) ? 1u : 4u;
      auto pass = [&](auto mr) {   // the call's globals lie behind the messages: never in the staged block
        WField f;
        while (w_next(mr, s2, f, L.bad)) {
          if (f.num != want || f.wt != 2u) continue;
          WSpan k, v;
          if (!w_entry(mr, f.s, k, v, L.bad)) break;
          const u32 kl = k.e - k.p;
          const u32 kh = w_hash(mr + k.p, kl);
          for (u32 c = 0; c < ncc; ++c) {
            const CBH_G WireCol& col = a.cols[c];
            if (col.root == root && col.nk != 0u && col.key_len[0] == kl && col.key_hash0 == kh && w_bytes_eq(mr + k.p, a.col_keys + col.key_off[0], kl)) {
              cur_p[c * CBH_BLOCK + threadIdx.x] = v.p; cur_e[c * CBH_BLOCK + threadIdx.x] = v.e;
            }
          }
        }
      };
      if (root == 4u) pass(mg); else pass(m);
    }
  }
  // attribute columns: one per attribute path the table's programs read (wave-uniform loop: the heap is allocated per wave)
  bool sens_container = false;
  WSpan gl; gl.p = a.globals_off; gl.e = a.globals_off + a.globals_len;
  for (u32 c = 0; c < a.n_cols; ++c) {
    const WireCol col = a.cols[c];
    // (mc: where this column's bytes are read, bc: what m[0] is in WireArgs.msg - uniform per column)
    auto column = [&](auto mc, const u32 bc) {
      u32 tag = CBH_T_ABSENT; u64 val = 0;
      bool is_container = false, is_map = false; WSpan body; body.p = body.e = 0; u32 fnum = 1u, need = 0u;
      u32 shape = 0u;   // 0 a plain container; auxData.jwts (root 3): 1 one named JWT as {"claims": {...}}, 2 all of them name -> {"claims": {...}}
      if (live) {
        const WSpan root = col.root == 0u ? principal : col.root == 1u ? resource : col.root == 4u ? gl : aux;
        const u32 root_fnum = (col.root == 2u || col.root == 4u) ? 1u : 4u;
        bool done = false; WSpan cur; cur.p = cur.e = 0;
        u32 k0 = 1u;
        if (col.root == 3u) {
          // AuxData.jwts (field 2): map<string, JWT>, JWT.claims (field 1): map<string, Value>.  The request view is
          // name -> {"claims": {...}} (check.go:536-554), so the second key of a path must be "claims".
          WSpan jwt; jwt.p = jwt.e = 0;
          const CBH_G u8* k1 = a.col_keys + col.key_off[1];
          if (col.nk == 0u) { is_container = true; is_map = true; shape = 2u; body = aux; fnum = 2u; done = true; }
          else if (!w_map_get(mc, aux, 2u, a.col_keys + col.key_off[0], col.key_len[0], jwt, L.bad)) { tag = col.nk == 1u ? CBH_T_ABSENT : CBH_T_ERR; done = true; }
          else if (col.nk == 1u) { is_container = true; is_map = true; shape = 1u; body = jwt; fnum = 1u; done = true; }
          else if (!(col.key_len[1] == 6u && k1[0] == 'c' && k1[1] == 'l' && k1[2] == 'a' && k1[3] == 'i' && k1[4] == 'm' && k1[5] == 's')) { tag = col.nk == 2u ? CBH_T_ABSENT : CBH_T_ERR; done = true; }
          else if (col.nk == 2u) { is_container = true; is_map = true; body = jwt; fnum = 1u; done = true; }
          else if (!w_map_get(mc, jwt, 1u, a.col_keys + col.key_off[2], col.key_len[2], cur, L.bad)) { tag = col.nk == 3u ? CBH_T_ABSENT : CBH_T_ERR; done = true; }
          else k0 = 3u;
        }
        else if (col.nk == 0u) { is_container = true; is_map = true; body = root; fnum = root_fnum; done = true; }
        else if (c < ncc) {   // found (or not) by the pass above
          cur.p = cur_p[c * CBH_BLOCK + threadIdx.x]; cur.e = cur_e[c * CBH_BLOCK + threadIdx.x];
          if (cur.p == CBH_NONE) { cur.p = cur.e = 0; tag = col.nk == 1u ? CBH_T_ABSENT : CBH_T_ERR; done = true; }
        }
        else if (!w_map_get(mc, root, root_fnum, a.col_keys + col.key_off[0], col.key_len[0], cur, L.bad)) { tag = col.nk == 1u ? CBH_T_ABSENT : CBH_T_ERR; done = true; }
        for (u32 k = k0; k < col.nk && !done; ++k) {
          WVal v;
          if (!w_value(mc, cur, v, L.bad) || v.kind != 5u) { tag = CBH_T_ERR; done = true; break; }
          if (!w_map_get(mc, v.s, 1u, a.col_keys + col.key_off[k], col.key_len[k], cur, L.bad)) { tag = (k == col.nk - 1u) ? CBH_T_ABSENT : CBH_T_ERR; done = true; }
        }
        if (!done) {
          WVal v;
          if (!w_value(mc, cur, v, L.bad)) { tag = CBH_T_NULL; }
          else if (v.kind == 1u) tag = CBH_T_NULL;
          else if (v.kind == 2u) { tag = CBH_T_DOUBLE; val = v.v; }             // structpb: every number is a double
          else if (v.kind == 3u) { tag = CBH_T_STRING; val = w_intern(a, mc, bc, v.s.p, v.s.e - v.s.p, 0u, L); }
          else if (v.kind == 4u) { tag = CBH_T_BOOL; val = v.v ? 1u : 0u; }
          else { is_container = true; is_map = v.kind == 5u; body = v.s; fnum = 1u; }
        }
        if (is_container) {
          if (shape == 0u) need = w_container_walk<false>(a, mc, bc, body, fnum, is_map, 0u, L);
          else if (shape == 1u) need = 2u + w_container_walk<false>(a, mc, bc, body, 1u, true, 0u, L);
          else {   // every named JWT: (name, {"claims": ..}) pairs, then per JWT its two-entry wrapper and its claims
            WSpan s2 = body; WField f; need = 0u;
            while (w_next(mc, s2, f, L.bad)) {
              if (f.num != 2u || f.wt != 2u) continue;
              WSpan k, v;
              if (!w_entry(mc, f.s, k, v, L.bad)) break;
              need += 4u + w_container_walk<false>(a, mc, bc, v, 1u, true, 0u, L);
            }
          }
          if (need > CBH_WIRE_MAX_VALUE_ENTRIES) { L.host = true; need = 0u; is_container = false; tag = CBH_T_NULL; }
        }
      }
      if (wave_ballot(is_container) != 0) {   // one heap reservation for the wave's containers of this column
        u32 total;
        const u32 pre = w_wave_prefix(need, 20u, lane, total);
        u32 base = 0;
        if (lane == 0u && total) base = w_add32(&a.stats->heap_used, total);
        base = wave_readlane(base, 0u);
        if (is_container) {
          const u32 off = base + pre;
          const u32 sid_claims = a.stats->sid_claims;
          auto put = [&](u32 slot, u32 t, u64 v) { if (slot < a.heap_cap) { a.heap_tag[slot] = (u8)t; a.heap_val[slot] = v; } };
          if (shape == 0u) {
            (void)w_container_walk<true>(a, mc, bc, body, fnum, is_map, off, L);
            tag = is_map ? CBH_T_MAP : CBH_T_LIST; val = w_container(off, w_count_fields(mc, body, fnum, L.bad));
          } else if (shape == 1u) {
            put(off, CBH_T_STRING, sid_claims);
            put(off + 1u, CBH_T_MAP, w_container(off + 2u, w_count_fields(mc, body, 1u, L.bad)));
            (void)w_container_walk<true>(a, mc, bc, body, 1u, true, off + 2u, L);
            tag = CBH_T_MAP; val = w_container(off, 1u);
          } else {
            const u32 n_top = w_count_fields(mc, body, 2u, L.bad);
            u32 slot = off, next = off + 2u * n_top;
            WSpan s2 = body; WField f;
            while (w_next(mc, s2, f, L.bad)) {
              if (f.num != 2u || f.wt != 2u) continue;
              WSpan k, v;
              if (!w_entry(mc, f.s, k, v, L.bad)) break;
              put(slot, CBH_T_STRING, w_intern(a, mc, bc, k.p, k.e - k.p, 0u, L));
              put(slot + 1u, CBH_T_MAP, w_container(next, 1u));
              slot += 2u;
              put(next, CBH_T_STRING, sid_claims);
              put(next + 1u, CBH_T_MAP, w_container(next + 2u, w_count_fields(mc, v, 1u, L.bad)));
              next += 2u + w_container_walk<true>(a, mc, bc, v, 1u, true, next + 2u, L);
            }
            tag = CBH_T_MAP; val = w_container(off, n_top);
          }
          if (c < 32u && ((a.sens_cols >> c) & 1u)) sens_container = true;
        }
      }
      if (in_range) { a.col_tag[(size_t)c * N + i] = (u8)tag; a.col_val[(size_t)c * N + i] = val; }
    };
    if (col.root == 4u) column(mg, 0u); else column(m, bias);
  }
#undef W_RQ
#undef W_SID
  {   // one route or several (WireStats.multi_route)?  A wave asks once: its first request's route against the batch's first
    const u64 rh = live ? w_route_hash(rt_kind, rt_ver, rt_scope) : 0ull;
    const u64 livem = wave_ballot(live);
    if (livem) {
      const u32 lead = (u32)__builtin_ctzll(livem);
      const u64 lh = wave_readlane64(rh, lead);
      const u64 others = wave_ballot(live && rh != lh);
      if (lane == lead) {
        CBH_G u64* first = (CBH_G u64*)&a.stats->route_lo;
        u64 cur = w_load64(first);
        if (cur == 0) { const u64 prev = w_cas64(first, 0, lh); cur = prev == 0 ? lh : prev; }
        if ((others != 0 || cur != lh) && a.stats->multi_route == 0u) a.stats->multi_route = 1u;
      }
    }
  }
  const u64 badm = wave_ballot(L.bad), hostm = wave_ballot(L.host), fullm = wave_ballot(L.dict_full), sensm = wave_ballot(sens_container);
  if (lane == 0u) {
    const u32 base = blockIdx.x * CBH_BLOCK + (threadIdx.x & ~63u);
    if (badm) w_min32(&a.stats->first_bad, base + (u32)__builtin_ctzll(badm));
    if (hostm) (void)w_add32(&a.stats->n_host, (u32)__builtin_popcountll(hostm));
    const u32 fl = (fullm ? CBH_WF_DICT_FULL : 0u) | (sensm ? CBH_WF_CONTAINER_IN_SENS : 0u);
    if (fl) w_or32(&a.stats->flags, fl);
  }
}

// Dynamic LDS of the fill kernel: [cur_p: ncc x 64 words][cur_e: the same][the wave's messages: lds_cap bytes, if staged]
// (ncc = min(n_cols, CBH_WIRE_CUR_COLS); the host sizes the launch with cbh_wire_fill_lds)
#ifndef CBH_HOSTSIM
__host__ __device__
#endif
static inline u32 cbh_wire_fill_cur_bytes(u32 n_cols) { return 2u * (n_cols < CBH_WIRE_CUR_COLS ? n_cols : CBH_WIRE_CUR_COLS) * CBH_BLOCK * 4u; }

// Two kernels, one body.  cbh_wire_fill_lds_kernel: the wave's messages lie back to back - one coalesced copy into LDS (16-byte chunks
// from the 16-byte boundary below the first message) and every lane parses there; the host launches it when the LARGEST block of the
// call (WireStats.max_block, from the count kernel) fits what it is willing to give a wave, sized for exactly that block.
// cbh_wire_fill_kernel: the same parse in place, for calls of larger messages.
template <bool STAGED>
__device__ __forceinline__ void w_fill_kernel_body(const WireArgs& a) {
  const u32 lane = threadIdx.x & 63u;
  const u32 i = blockIdx.x * CBH_BLOCK + threadIdx.x;
  const u32 N = a.n;
  const bool in_range = i < N;
  bool live = in_range && a.status[i] == CBH_WS_OK;
  const u32 ncc = a.n_cols < CBH_WIRE_CUR_COLS ? a.n_cols : CBH_WIRE_CUR_COLS;
  CBH_L u32* cur_p = (CBH_L u32*)cbh_dyn_lds;
  CBH_L u32* cur_e = cur_p + ncc * CBH_BLOCK;
  if (STAGED) {
    CBH_L u8* stage = (CBH_L u8*)cbh_dyn_lds + cbh_wire_fill_cur_bytes(a.n_cols);
    const u32 w0 = blockIdx.x * CBH_BLOCK, w1 = (w0 + CBH_BLOCK < N) ? w0 + CBH_BLOCK : N;
    const u64 lo64 = a.moff[w0 < N ? w0 : N], hi64 = a.moff[w1];
    const u32 lo16 = (u32)lo64 & ~15u;
    // (the host sized lds_cap by the largest block; a block that still does not fit - offsets the count kernel saw otherwise - sits out)
    const bool fits = hi64 >= lo64 && hi64 <= (u64)a.dver_off && (u64)((u32)hi64 - lo16) + 16u + CBH_WIRE_SLACK <= (u64)a.lds_cap;   // (uniform)
    const u32 need = fits ? (u32)hi64 - lo16 : 0u;
    w_stage_block(stage, a.msg + lo16, need, lane);
    __syncthreads();   // (one wave per workgroup, uniform: the copy has landed before any lane parses)
    // a lane whose message does not lie inside the block (offsets that are not monotonic: the call fails anyway) sits out
    if (in_range) { const u64 o0 = a.moff[i], o1 = a.moff[i + 1u]; live = live && fits && o0 >= lo64 && o1 >= o0 && o1 <= hi64; }
    w_fill_body(a, (WLds)stage, lo16, live, in_range, lane, i, cur_p, cur_e);
  } else {
    w_fill_body(a, (WGlob)a.msg, 0u, live, in_range, lane, i, cur_p, cur_e);
  }
}
#ifdef CBH_HOSTSIM
static void cbh_wire_fill_kernel(WireArgs a) { w_fill_kernel_body<false>(a); }
static void cbh_wire_fill_lds_kernel(WireArgs a) { w_fill_kernel_body<true>(a); }
#else
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_fill_kernel(WireArgs a) { w_fill_kernel_body<false>(a); }
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_fill_lds_kernel(WireArgs a) { w_fill_kernel_body<true>(a); }
#endif

// ============================================================================================================================
// The way back (SURVEY §8 f2 on the GPU): serialized enginev1.CheckOutput messages written by the device, for a batch the
// device flattened.  The reference builds them in checkWithAuditTrail / setEffect (check.go:64-94, 513-530); cbh_ingest.cpp
// assemble_outputs does it per input on a host thread (90 % of the device road's wall time once the flattening had moved to
// the GPU, profiles/r03_e2e_wire_inclusive.json).  Three launches - sizes, one-wave scan, bytes - and one copy back:
//   CheckOutput { 1 request_id, 2 resource_id, 3 actions map<string, ActionEffect{1 effect, 2 policy, 3 scope}>, 4 effective_derived_roles* }
// byte for byte what cbi_assemble_wire_pb writes (tests hold the two against each other): an action named twice keeps its
// first place and takes the later result unless the earlier one is a DENY and the later is not (check.go:513-530); policy keys
// as namer.PolicyKeyFromFQN gives them (namer.go:95-134), resource kinds / versions of the pre-0.30 form rewritten on the fly.
#define CBH_WO_UNSUPPORTED 1u    /* = CBI_OUT_UNSUPPORTED (include/cerbos_ingest.h) */
#define CBH_WO_CEL_ERROR 2u      /* = CBI_OUT_CEL_ERROR */
#define CBH_WO_WANTS_TRACE 16u   /* = CBI_OUT_WANTS_TRACE */
#define CBH_WO_MAX_OUTPUT 0xFFFFFFu   /* bytes of one CheckOutput (larger: the call fails) */

struct WireOutStats { u64 total; u32 errors; u32 pad; };   // errors: bit 0 a policy / scope id out of range, bit 1 an output too large
struct WireOutArgs {
  const CBH_G u32* t_str_off; const CBH_G u8* t_str_bytes;
  const CBH_G u32* scope_sid; u32 n_scopes; u32 n_policies;
  const CBH_G u32* name_off; const CBH_G u8* name_bytes; u32 n_dr; u32 n;   // names: policy keys (CBH_P_TABLE ids), then derived roles
  const CBH_G u8* msg; const CBH_G u64* moff; u32 dver_off, dver_len;
  const CBH_G u32* req_u32; const CBH_G u32* tuple_action; const CBH_G u32* in_span; const CBH_G u32* act_span;
  const CBH_G u8* effect; const CBH_G u32* policy; const CBH_G u32* scope; const CBH_G u8* status; const CBH_G u64* edr;
  CBH_G u32* sizes; CBH_G u64* wavesum; CBH_G u64* waveoff; CBH_G WireOutStats* stats;
  CBH_G u8* out; CBH_G u64* out_off; CBH_G u8* out_flags;
  u32 lds_cap, out_bias;   // bytes of dynamic LDS a wave of the write kernel may stage its outputs in; output byte g lies at out[out_bias + g]
                           // (`out` 16-byte aligned, out_bias < 16: the block the kernel writes may be the CALLER's page-locked buffer)
  const CBH_G u32* inv;   // a batch grouped by route (cbh_wire_route_kernel): input -> position of its per-request results; else null
  CBH_G WireOutStats* host_stats;   // page-locked host memory the device can write (or null): the scan kernel leaves {total, errors} there
};

template <bool WRITE, class P = CBH_G u8*> struct WSink {
  static constexpr bool writes = WRITE;
  P w; u32 n;
  __device__ __forceinline__ void byte(u32 b) { if (WRITE) w[n] = (u8)b; ++n; }
  __device__ __forceinline__ void varint(u64 v) { while (v >= 0x80u) { byte(((u32)v & 0x7Fu) | 0x80u); v >>= 7; } byte((u32)v); }
  // (eight bytes per load and store; the last ones one by one: what lies behind them is another lane's)
  __device__ __forceinline__ void bytes(const CBH_G u8* p, u32 len) {
    if (WRITE) {
      u32 i = 0;
      for (; i + 8u <= len; i += 8u) {
        const u64 v = w_peek8(p, i);
#ifdef CBH_HOSTSIM
        memcpy(w + n + i, &v, 8);
#else
        __builtin_memcpy(w + n + i, &v, 8);
#endif
      }
      if (i < len) { u64 v = w_peek8(p, i); for (; i < len; ++i) { w[n + i] = (u8)v; v >>= 8; } }
    }
    n += len;
  }
  // namer.go:213-218: a name of the pre-0.30 form has every run of characters outside [0-9A-Za-z_.] replaced by one '_'
  __device__ __forceinline__ void rewritten(const CBH_G u8* p, u32 len) {
    bool in_run = false;
    for (u32 i = 0; i < len; ++i) { const u32 c = p[i]; const bool ok = w_kind_ok_char(c); if (ok || !in_run) byte(ok ? c : (u32)'_'); in_run = !ok; }
  }
  __device__ __forceinline__ void lit(const char* s, u32 len) { for (u32 i = 0; i < len; ++i) byte((u32)(u8)s[i]); }
};
__device__ __forceinline__ u32 w_varint_size(u64 v) { u32 n = 1; while (v >= 0x80u) { v >>= 7; ++n; } return n; }

// One of the strings a policy key is made of - the resource's kind / version, the principal's id / version - as the key shows it
// (sanitised): looked at once per input, not once per action and pass.
struct WKeyPart { const CBH_G u8* p; u32 len; u32 out_len; bool rewrite, known; };
__device__ __forceinline__ void w_key_part(WKeyPart& k) {
  if (k.known) return;
  k.known = true;
  k.rewrite = w_kind_needs_rewrite(k.p, k.len);
  k.out_len = k.len;
  if (k.rewrite) { WSink<false> c; c.w = nullptr; c.n = 0; c.rewritten(k.p, k.len); k.out_len = c.n; }
}
struct WKeyParts { WKeyPart kind, rver, pid, pver; };

// the policy key of a device policy word (cbh_ingest.cpp policy_key): its length ...
__device__ __forceinline__ u32 w_policy_key_len(const WireOutArgs& a, WKeyParts& kp, u32 word, u32& errors) {
  const u32 k = word >> 28, ident = word & 0x0FFFFFFFu;
  if (k == CBH_P_EMPTY) return 0u;
  if (k == CBH_P_NO_MATCH) return 8u;
  if (k == CBH_P_NO_MATCH_SCOPE_PERMISSIONS) return 30u;
  if (k == CBH_P_TABLE) {
    if (ident >= a.n_policies) { errors |= 1u; return 0u; }
    return a.name_off[ident + 1u] - a.name_off[ident];
  }
  if (k == CBH_P_RESOURCE || k == CBH_P_PRINCIPAL) {
    if (ident >= a.n_scopes) { errors |= 1u; return 0u; }
    const bool rp = k == CBH_P_RESOURCE;
    WKeyPart& name = rp ? kp.kind : kp.pid; WKeyPart& ver = rp ? kp.rver : kp.pver;
    w_key_part(name); w_key_part(ver);
    const u32 sid = a.scope_sid[ident], sl = a.t_str_off[sid + 1u] - a.t_str_off[sid];
    return (rp ? 9u : 10u) + name.out_len + 2u + ver.out_len + (sl ? 1u + sl : 0u);
  }
  errors |= 1u;
  return 0u;
}
// ... and its bytes (after w_policy_key_len of the same word: the parts are known, the ids in range)
template <class SINK>
__device__ __forceinline__ void w_policy_key(const WireOutArgs& a, SINK& o, const WKeyParts& kp, u32 word) {
  const u32 k = word >> 28, ident = word & 0x0FFFFFFFu;
  if (k == CBH_P_NO_MATCH) { o.lit("NO_MATCH", 8); return; }
  if (k == CBH_P_NO_MATCH_SCOPE_PERMISSIONS) { o.lit("NO_MATCH_FOR_SCOPE_PERMISSIONS", 30); return; }
  if (k == CBH_P_TABLE) {
    if (ident < a.n_policies) o.bytes(a.name_bytes + a.name_off[ident], a.name_off[ident + 1u] - a.name_off[ident]);
    return;
  }
  if ((k == CBH_P_RESOURCE || k == CBH_P_PRINCIPAL) && ident < a.n_scopes) {
    const bool rp = k == CBH_P_RESOURCE;
    const WKeyPart& name = rp ? kp.kind : kp.pid; const WKeyPart& ver = rp ? kp.rver : kp.pver;
    if (rp) o.lit("resource.", 9); else o.lit("principal.", 10);
    if (name.rewrite) o.rewritten(name.p, name.len); else o.bytes(name.p, name.len);
    o.lit(".v", 2);
    if (ver.rewrite) o.rewritten(ver.p, ver.len); else o.bytes(ver.p, ver.len);
    const u32 sid = a.scope_sid[ident], so = a.t_str_off[sid], sl = a.t_str_off[sid + 1u] - so;
    if (sl) { o.byte('/'); o.bytes(a.t_str_bytes + so, sl); }
  }
}

// The CheckOutput of input i into `o`; returns its CBH_WO_* flags.  What the output repeats of its input and what the decision
// kernels left for its actions is fetched FIRST, all loads in flight together - the input's spans as three 16-byte loads, and for an
// input of at most four actions (nearly all) their ids, effects, policies, scopes, states and name spans - so that what follows is
// arithmetic and copies, not a chain of dependent loads.
#define CBH_WO_FAST_ACTIONS 4u
template <class SINK>
__device__ __forceinline__ u32 w_output(const WireOutArgs& a, u32 i, SINK& o, u32& errors) {
  const u32 N = a.n;
  const CBH_G u8* m = a.msg + (u32)a.moff[i];
  u32 sp[2u * CBH_WSPAN_N];
  {
    const CBH_G u32* q = a.in_span + (size_t)i * 2u * CBH_WSPAN_N;
#ifndef CBH_HOSTSIM
    typedef u32 v4 __attribute__((ext_vector_type(4)));
    const v4 x0 = *(const CBH_G v4*)q, x1 = *(const CBH_G v4*)(q + 4), x2 = *(const CBH_G v4*)(q + 8);
    sp[0] = x0.x; sp[1] = x0.y; sp[2] = x0.z; sp[3] = x0.w; sp[4] = x1.x; sp[5] = x1.y; sp[6] = x1.z; sp[7] = x1.w;
    sp[8] = x2.x; sp[9] = x2.y; sp[10] = x2.z; sp[11] = x2.w;
#else
    for (u32 k = 0; k < 2u * CBH_WSPAN_N; ++k) sp[k] = q[k];
#endif
  }
  const u32 act_off = a.req_u32[(size_t)CBH_RQ_ACT_OFF * N + i], na = a.req_u32[(size_t)CBH_RQ_ACT_CNT * N + i];
  const u64 edr = a.edr[a.inv ? a.inv[i] : i];
  // per action (the first CBH_WO_FAST_ACTIONS in registers; a longer list reads the rest where it needs them)
  u32 f_id[CBH_WO_FAST_ACTIONS], f_eff[CBH_WO_FAST_ACTIONS], f_st[CBH_WO_FAST_ACTIONS], f_pol[CBH_WO_FAST_ACTIONS], f_sc[CBH_WO_FAST_ACTIONS];
  u32 f_no[CBH_WO_FAST_ACTIONS], f_nl[CBH_WO_FAST_ACTIONS];
#pragma unroll
  for (u32 k = 0; k < CBH_WO_FAST_ACTIONS; ++k) {
    const u32 t = act_off + (k < na ? k : 0u);   // (an index that exists; what it gives is not looked at for k >= na)
    const bool in = k < na;
    f_id[k] = in ? a.tuple_action[t] : 0u; f_eff[k] = in ? (u32)a.effect[t] : 0u; f_st[k] = in ? (u32)a.status[t] : 0u;
    f_pol[k] = in ? a.policy[t] : 0u; f_sc[k] = in ? a.scope[t] : 0u;
    f_no[k] = in ? a.act_span[2u * t] : 0u; f_nl[k] = in ? a.act_span[2u * t + 1u] : 0u;
  }
  auto id_of = [&](u32 k) -> u32 { return k < CBH_WO_FAST_ACTIONS ? f_id[k] : a.tuple_action[act_off + k]; };
  auto eff_of = [&](u32 k) -> u32 { return k < CBH_WO_FAST_ACTIONS ? f_eff[k] : (u32)a.effect[act_off + k]; };
  if (sp[1]) { o.byte(0x0Au); o.varint(sp[1]); o.bytes(m + sp[0], sp[1]); }        // 1 request_id
  if (sp[11]) { o.byte(0x12u); o.varint(sp[11]); o.bytes(m + sp[10], sp[11]); }    // 2 resource_id
  u32 flags = 0;
  for (u32 k = 0; k < na; ++k) {
    const u32 st = k < CBH_WO_FAST_ACTIONS ? f_st[k] : (u32)a.status[act_off + k];
    flags |= st == CBH_ST_UNSUPPORTED ? CBH_WO_UNSUPPORTED : st == CBH_ST_CEL_ERROR ? CBH_WO_CEL_ERROR : st == CBH_ST_WANTS_TRACE ? CBH_WO_WANTS_TRACE : 0u;
  }
  WKeyParts kp;
  kp.kind.p = m + sp[6]; kp.kind.len = sp[7]; kp.pid.p = m + sp[2]; kp.pid.len = sp[3];
  kp.rver.p = m + sp[8]; kp.rver.len = sp[9]; kp.pver.p = m + sp[4]; kp.pver.len = sp[5];
  if (kp.rver.len == 0u) { kp.rver.p = a.msg + a.dver_off; kp.rver.len = a.dver_len; }
  if (kp.pver.len == 0u) { kp.pver.p = a.msg + a.dver_off; kp.pver.len = a.dver_len; }
  kp.kind.known = kp.pid.known = kp.rver.known = kp.pver.known = false;
  kp.kind.rewrite = kp.pid.rewrite = kp.rver.rewrite = kp.pver.rewrite = false;
  kp.kind.out_len = kp.pid.out_len = kp.rver.out_len = kp.pver.out_len = 0u;
  for (u32 k = 0; k < na; ++k) {
    const u32 id = id_of(k);
    bool first = true;
    for (u32 q = 0; q < k && first; ++q) first = id_of(q) != id;
    if (!first) continue;   // named before: that entry carries the result
    u32 j = k;
    for (u32 q = k + 1u; q < na; ++q)
      if (id_of(q) == id && (eff_of(q) == CBH_EFFECT_DENY || eff_of(j) != CBH_EFFECT_DENY)) j = q;
    const bool kf = k < CBH_WO_FAST_ACTIONS, jf = j < CBH_WO_FAST_ACTIONS;
    const u32 name_o = kf ? f_no[k] : a.act_span[2u * (act_off + k)], name_l = kf ? f_nl[k] : a.act_span[2u * (act_off + k) + 1u];
    const u32 effect = eff_of(j), word = jf ? f_pol[j] : a.policy[act_off + j], sc = jf ? f_sc[j] : a.scope[act_off + j];
    const u32 pol_l = w_policy_key_len(a, kp, word, errors);
    u32 scope_o = 0, scope_l = 0;
    if (sc != CBH_NONE) {
      if (sc >= a.n_scopes) errors |= 1u;
      else { const u32 sid = a.scope_sid[sc]; scope_o = a.t_str_off[sid]; scope_l = a.t_str_off[sid + 1u] - scope_o; }
    }
    const u32 eff_len = (effect ? 1u + w_varint_size(effect) : 0u) + (pol_l ? 1u + w_varint_size(pol_l) + pol_l : 0u) + (scope_l ? 1u + w_varint_size(scope_l) + scope_l : 0u);
    const u32 ent_len = 1u + w_varint_size(name_l) + name_l + 1u + w_varint_size(eff_len) + eff_len;
    o.byte(0x1Au); o.varint(ent_len);                       // 3 actions: map entry
    o.byte(0x0Au); o.varint(name_l); o.bytes(m + name_o, name_l);
    o.byte(0x12u); o.varint(eff_len);
    if (effect) { o.byte(0x08u); o.varint(effect); }
    if (pol_l) { o.byte(0x12u); o.varint(pol_l); if (SINK::writes) w_policy_key(a, o, kp, word); else o.n += pol_l; }
    if (scope_l) { o.byte(0x1Au); o.varint(scope_l); o.bytes(a.t_str_bytes + scope_o, scope_l); }
  }
  u64 left = a.n_dr >= 64u ? edr : edr & ((1ull << a.n_dr) - 1ull);
  while (left) {                                              // 4 effective_derived_roles
    const u32 d = (u32)__builtin_ctzll(left); left &= left - 1ull;
    const u32 no = a.name_off[a.n_policies + d], nl = a.name_off[a.n_policies + d + 1u] - no;
    o.byte(0x22u); o.varint(nl); o.bytes(a.name_bytes + no, nl);
  }
  return flags;
}

__device__ __forceinline__ u64 w_wave_prefix64(u64 x, u32 bits, u32 lane, u64& total) {
  u64 pre = 0; total = 0;
  const u64 lt = (1ull << lane) - 1ull;
  for (u32 b = 0; b < bits; ++b) {
    const u64 mk = wave_ballot(((x >> b) & 1ull) != 0ull);
    pre += (u64)__builtin_popcountll(mk & lt) << b;
    total += (u64)__builtin_popcountll(mk) << b;
  }
  return pre;
}

#ifdef CBH_HOSTSIM
static void cbh_wire_out_size_kernel(WireOutArgs a)
#else
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_out_size_kernel(WireOutArgs a)
#endif
{
  const u32 lane = threadIdx.x & 63u;
  const u32 i = blockIdx.x * CBH_BLOCK + threadIdx.x;
  u32 sz = 0, errors = 0;
  if (i < a.n) {
    WSink<false> o; o.w = nullptr; o.n = 0;
    const u32 fl = w_output(a, i, o, errors);
    sz = o.n;
    if (sz > CBH_WO_MAX_OUTPUT) { errors |= 2u; sz = 0; }
    a.sizes[i] = sz; a.out_flags[i] = (u8)fl;
  }
  u32 total;
  (void)w_wave_prefix(sz, 24u, lane, total);
  const u64 errm = wave_ballot(errors != 0u);
  if (lane == 0u) a.wavesum[blockIdx.x * (CBH_BLOCK / 64u) + threadIdx.x / 64u] = total;
  if (errm && errors) w_or32(&a.stats->errors, errors);
}

#ifdef CBH_HOSTSIM
static void cbh_wire_out_scan_kernel(WireOutArgs a)
#else
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_out_scan_kernel(WireOutArgs a)
#endif
{
  const u32 lane = threadIdx.x & 63u;
  const u32 nw = (a.n + 63u) / 64u, per = (nw + 63u) / 64u;
  const u32 lo = lane * per < nw ? lane * per : nw, hi = (lo + per < nw) ? lo + per : nw;
  u64 s = 0;
  {   // (sixteen sums per round trip, as cbh_wire_scan_kernel)
    u32 w = lo;
    for (; w + 16u <= hi; w += 16u) {
      u32 x[32];
#pragma unroll
      for (u32 j = 0; j < 8u; ++j) W_LOAD4(x + 4u * j, (const CBH_G u32*)(a.wavesum + w) + 4u * j);
#pragma unroll
      for (u32 j = 0; j < 16u; ++j) s += (u64)x[2u * j] | ((u64)x[2u * j + 1u] << 32);
    }
    for (; w < hi; ++w) s += a.wavesum[w];
  }
  u64 total;
  u64 p = w_wave_prefix64(s, 40u, lane, total);
  {
    u32 w = lo;
    for (; w + 16u <= hi; w += 16u) {
      u32 x[32];
#pragma unroll
      for (u32 j = 0; j < 8u; ++j) W_LOAD4(x + 4u * j, (const CBH_G u32*)(a.wavesum + w) + 4u * j);
#pragma unroll
      for (u32 j = 0; j < 16u; ++j) { const u64 c = (u64)x[2u * j] | ((u64)x[2u * j + 1u] << 32); x[2u * j] = (u32)p; x[2u * j + 1u] = (u32)(p >> 32); p += c; }
#pragma unroll
      for (u32 j = 0; j < 8u; ++j) W_STORE4((CBH_G u32*)(a.waveoff + w) + 4u * j, x + 4u * j);
    }
    for (; w < hi; ++w) { a.waveoff[w] = p; p += a.wavesum[w]; }
  }
  if (lane == 0u) {
    a.stats->total = total; a.out_off[a.n] = total;
    if (a.host_stats) {   // what the host waits for, and the error bits cleared for the next results of this batch
      a.host_stats->total = total; a.host_stats->errors = w_load32(&a.stats->errors); a.host_stats->pad = 0u;
      a.stats->errors = 0u;
    }
  }
}

#ifdef CBH_HOSTSIM
static void cbh_wire_out_write_kernel(WireOutArgs a)
#else
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_out_write_kernel(WireOutArgs a)
#endif
{
  const u32 lane = threadIdx.x & 63u;
  const u32 i = blockIdx.x * CBH_BLOCK + threadIdx.x;
  const u32 sz = i < a.n ? a.sizes[i] : 0u;
  u32 total;
  const u32 pre = w_wave_prefix(sz, 24u, lane, total);
  const u64 woff = a.waveoff[blockIdx.x * (CBH_BLOCK / 64u) + threadIdx.x / 64u];
  const u64 off = woff + pre;
  if (i < a.n) a.out_off[i] = off;
  u32 errors = 0;
  const u32 skew = (u32)(woff + a.out_bias) & 15u;
  if (total + skew + 16u <= a.lds_cap) {
    // The wave's outputs lie back to back: every lane writes its bytes into LDS (at the block's own 16-byte skew) and the
    // wave copies the block out in 16-byte stores - one byte per store instruction and lane otherwise.
#ifndef CBH_HOSTSIM
    typedef CBH_L u8* LP;
#else
    typedef u8* LP;
#endif
    if (i < a.n && sz) { WSink<true, LP> o; o.w = (LP)cbh_dyn_lds + skew + pre; o.n = 0; (void)w_output(a, i, o, errors); }
    __syncthreads();   // (one wave per workgroup, uniform branch)
    const u32 end = skew + total;   // the block occupies LDS bytes [skew, end); global byte g = woff - skew + (LDS byte)
    CBH_G u8* gbase = a.out + (woff + a.out_bias - skew);
    for (u32 o16 = lane * 16u; o16 < end; o16 += 64u * 16u) {
      if (o16 >= skew && o16 + 16u <= end) {
#ifndef CBH_HOSTSIM
        typedef u32 v4 __attribute__((ext_vector_type(4)));
        *(CBH_G v4*)(gbase + o16) = *(const CBH_L v4*)((const CBH_L u8*)cbh_dyn_lds + o16);
#else
        for (u32 k = 0; k < 16u; ++k) gbase[o16 + k] = cbh_dyn_lds[o16 + k];
#endif
      } else {   // the first / last chunk: only the block's own bytes (the neighbours' belong to other waves)
        for (u32 k = 0; k < 16u; ++k) if (o16 + k >= skew && o16 + k < end) gbase[o16 + k] = ((LP)cbh_dyn_lds)[o16 + k];
      }
    }
  } else if (i < a.n && sz) {
    WSink<true> o; o.w = a.out + a.out_bias + off; o.n = 0;
    (void)w_output(a, i, o, errors);
  }
}

// ============================================================================================================================
// Routing on the device.  The decision kernels walk the table wave by wave and merge the lanes that stand at the same scope
// of the same (version, kind): a wave whose 64 requests are of 40 different routes walks 40 buckets.  The host flattener
// therefore orders requests by route (cbh_ingest.cpp sort_batch: kind, version, scope, role count, role signature); a batch the
// device flattened arrives in INPUT order.  Three launches group it: cbh_wire_route_kernel gives every request its route - a
// 64-bit fingerprint of those five facts, found or claimed in a small open-addressing table by one compare-and-swap (two
// routes that collide merely share a group: grouping is for speed, never for meaning) - and its rank inside the route, drawn
// with ONE returning atomic per distinct route and wave; cbh_wire_route_scan_kernel turns the routes' counts into their
// starts; cbh_wire_gather_kernel writes the per-request arrays (request words, attribute columns) in grouped order.  Tuples
// stay where they are (a request carries its ACT_OFF along), so results come back in input tuple order as before; only the
// per-request derived-role mask is indexed by grouped position (WireRouteArgs.inv maps an input to it).
#define CBH_WIRE_ROUTE_SLOTS 16384u   /* routes a call can tell apart (a fuller table: the batch stays in input order) */

struct WireRouteArgs {
  u32 n, n_cols; u32 pad[2];
  const CBH_G u32* req_u32; const CBH_G u32* roles; const CBH_G u8* col_tag; const CBH_G u64* col_val;   // as the fill kernel left them
  CBH_G u64* rt_key;     // [CBH_WIRE_ROUTE_SLOTS] fingerprints (0 = empty)
  CBH_G u32* rt_cnt;     // [CBH_WIRE_ROUTE_SLOTS + 2] requests per route; after the scan: the routes' starts; [SLOTS] = routes in use, [SLOTS + 1] = overflow flag
  CBH_G u32* slot;       // [n] route of a request
  CBH_G u32* rank;       // [n] its rank inside the route
  CBH_G u32* inv;        // [n] input -> grouped position
  CBH_G u32* req_out; CBH_G u8* col_tag_out; CBH_G u64* col_val_out;
  const CBH_G u32* multi;   // WireStats.multi_route of the fill that wrote the batch (null: not known - route everything)
  // page-locked host memory the device can write (or null): the scan kernel leaves {routes in use, overflow flag} in host_routes and -
  // it runs behind the fill - a copy of the fill's statistics in host_stats
  CBH_G u32* host_routes; const CBH_G WireStats* stats; CBH_G WireStats* host_stats;
};


#ifdef CBH_HOSTSIM
static void cbh_wire_route_kernel(WireRouteArgs a)
#else
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_route_kernel(WireRouteArgs a)
#endif
{
  const u32 lane = threadIdx.x & 63u;
  const u32 i = blockIdx.x * CBH_BLOCK + threadIdx.x;
  const u32 N = a.n;
  const bool live = i < N;
  if (a.multi && w_load32(a.multi) == 0u) return;   // one route: nothing to group (the counters stay 0: "no route in use")
  u64 h = 0;
  if (live) {
    const u32 kind = a.req_u32[(size_t)CBH_RQ_KIND * N + i], ver = a.req_u32[(size_t)CBH_RQ_R_VERSION * N + i], scope = a.req_u32[(size_t)CBH_RQ_R_SCOPE * N + i];
    // (kind, version, scope): what lanes must share to walk a bucket together; the role lists - the host sort's secondary key - are
    // left out: they differ from request to request, are decided per lane by class masks, and would only scatter a kind's requests
    h = w_route_hash(kind, ver, scope);
  }
  // One lane per distinct route of the wave goes to the table (a stream of one kind would otherwise send every lane's
  // compare-and-swap to the same word) and draws the ranks of all the wave's requests of that route with one returning add.
  u32 my_slot = CBH_NONE, my_rank = 0;
  u64 todo = wave_ballot(live);
  while (todo) {
    const u32 lead = (u32)__builtin_ctzll(todo);
    const u64 lh = wave_readlane64(h, lead);
    const u64 same = wave_ballot(live && h == lh);
    u32 ls = CBH_NONE, base = 0;
    if (lane == lead) {
      u32 s = (u32)(lh >> 20) & (CBH_WIRE_ROUTE_SLOTS - 1u);
      for (u32 p = 0; p < 64u; ++p, s = (s + 1u) & (CBH_WIRE_ROUTE_SLOTS - 1u)) {
        u64 cur = w_load64(a.rt_key + s);
        if (cur == 0) { const u64 prev = w_cas64(a.rt_key + s, 0, lh); cur = prev == 0 ? lh : prev; }
        if (cur == lh) { ls = s; break; }
      }
      if (ls != CBH_NONE) base = w_add32(a.rt_cnt + ls, (u32)__builtin_popcountll(same));
    }
    ls = wave_readlane(ls, lead); base = wave_readlane(base, lead);
    if (live && h == lh) { my_slot = ls; my_rank = base + (u32)__builtin_popcountll(same & ((1ull << lane) - 1ull)); }
    todo &= ~same;
  }
  const u64 lost = wave_ballot(live && my_slot == CBH_NONE);
  if (live) { a.slot[i] = my_slot; a.rank[i] = my_rank; }
  if (lost && lane == 0u) w_or32(a.rt_cnt + CBH_WIRE_ROUTE_SLOTS + 1u, 1u);   // the table is too full to place a route: no grouping this time
}

#ifdef CBH_HOSTSIM
static void cbh_wire_route_scan_kernel(WireRouteArgs a)
#else
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_route_scan_kernel(WireRouteArgs a)
#endif
{
  const u32 lane = threadIdx.x & 63u;
  const u32 per = CBH_WIRE_ROUTE_SLOTS / 64u;
  if (a.host_stats && a.stats) {   // (the fill's launch has ended: its statistics are in memory)
    const CBH_G u32* src = (const CBH_G u32*)a.stats; CBH_G u32* dst = (CBH_G u32*)a.host_stats;
    for (u32 k = lane; k < (u32)(sizeof(WireStats) / 4u); k += 64u) dst[k] = w_load32(src + k);
  }
  if (a.multi && w_load32(a.multi) == 0u) {   // (cbh_wire_route_kernel did not run: every counter is 0, "routes in use" too)
    if (a.host_routes && lane == 0u) { a.host_routes[0] = 0u; a.host_routes[1] = 0u; }
    return;
  }
  u32 s = 0, used = 0;
  for (u32 k = 0; k < per; ++k) { const u32 c = a.rt_cnt[lane * per + k]; s += c; used += c != 0u; }
  u32 total, total_used;
  u32 p = w_wave_prefix(s, 32u, lane, total);
  (void)w_wave_prefix(used, 15u, lane, total_used);
  for (u32 k = 0; k < per; ++k) { const u32 c = a.rt_cnt[lane * per + k]; a.rt_cnt[lane * per + k] = p; p += c; }
  if (lane == 0u) {
    a.rt_cnt[CBH_WIRE_ROUTE_SLOTS] = total_used;
    if (a.host_routes) { a.host_routes[0] = total_used; a.host_routes[1] = w_load32(a.rt_cnt + CBH_WIRE_ROUTE_SLOTS + 1u); }
  }
}

#ifdef CBH_HOSTSIM
static void cbh_wire_gather_kernel(WireRouteArgs a)
#else
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_gather_kernel(WireRouteArgs a)
#endif
{
  const u32 i = blockIdx.x * CBH_BLOCK + threadIdx.x;
  const u32 N = a.n;
  // (a route found no slot, or the stream is of one route: nothing to group - the host reads the same two words)
  if (i >= N || a.rt_cnt[CBH_WIRE_ROUTE_SLOTS + 1u] != 0u || a.rt_cnt[CBH_WIRE_ROUTE_SLOTS] <= 1u) return;
  const u32 pos = a.rt_cnt[a.slot[i]] + a.rank[i];
  a.inv[i] = pos;
  for (u32 f = 0; f < CBH_RQ_NFIELDS; ++f) a.req_out[(size_t)f * N + pos] = a.req_u32[(size_t)f * N + i];
  for (u32 c = 0; c < a.n_cols; ++c) { a.col_tag_out[(size_t)c * N + pos] = a.col_tag[(size_t)c * N + i]; a.col_val_out[(size_t)c * N + pos] = a.col_val[(size_t)c * N + i]; }
}

#if !defined(CBH_HOSTSIM) || defined(CBH_HOSTSIM_ENGINE)
// A few words of device memory to page-locked host memory the device can write (the batch's block): what the host has to know
// between two launches - totals, flags, the shape of the batch - without a copy engine in between.  A small device-to-host copy
// queues behind whatever bulk copy is on the link (the slices of cbh_wire_check_pb upload 12 MB each): measured, a slice's first
// answers left only after the LAST slice's messages had arrived.  The host reads the words after synchronising with the stream.
struct WirePublishArgs { const CBH_G u32* src; CBH_G u32* dst; u32 n_words; u32 pad; };
__global__ __launch_bounds__(64) void cbh_wire_publish_kernel(WirePublishArgs a) {
  for (u32 k = threadIdx.x; k < a.n_words; k += 64u) a.dst[k] = a.src[k];
}
// derived-role masks back in input order (cbh_result_download of a grouped batch).  (Arguments in a struct, as everywhere here: a
// kernel whose SIGNATURE carries address-space qualified pointers has one mangled name in the device pass and another on the host.)
struct WireUnsortArgs { const CBH_G u64* edr_grouped; const CBH_G u32* inv; CBH_G u64* edr_input; u32 n; u32 pad; };
__global__ __launch_bounds__(256) void cbh_wire_unsort_edr_kernel(WireUnsortArgs a) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) a.edr_input[i] = a.edr_grouped[a.inv[i]];
}
// the other way round: something known per input (the trail group of a request's inputs) to where the grouped batch keeps that input
struct WireScatterArgs { const CBH_G u32* by_input; const CBH_G u32* inv; CBH_G u32* by_position; u32 n; u32 pad; };
__global__ __launch_bounds__(256) void cbh_wire_scatter_u32_kernel(WireScatterArgs a) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) a.by_position[a.inv[i]] = a.by_input[i];
}
#endif
