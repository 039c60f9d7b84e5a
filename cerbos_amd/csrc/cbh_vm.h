// Device-side value model and CEL bytecode interpreter for the decision kernel.
// CDNA4 / gfx950 only.  One lane evaluates one (principal, resource, action) tuple; the
// operand stack, locals and iteration slots live in LDS laid out [slot][lane] so a
// wave's accesses to one slot hit 64 consecutive banks-worth of addresses (conflict free).
//
// Semantics restate the CEL behaviour the reference gets from cel-go v0.30.0 through
// internal/conditions/cel.go:65-107 (see oracle/celeval.py for the CPU restatement and
// SURVEY.md Appendix B for the rules): cross-type numeric comparison, int64 overflow
// errors, error absorption in && / ||, missing-key errors, leaf errors -> false.
#pragma once
#ifndef CBH_HOSTSIM
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "../../include/cerbos_hip.h"
#include "cbh_blob.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

#define CBH_BLOCK 256
#define CBH_STACK_DEPTH 10
#define CBH_MAX_LOCALS 4
#define CBH_MAX_ITERS 2

struct TableDev {
  const u32* str_off; const u8* str_bytes;
  const u32* scope_parent; const u32* scope_flags;
  const CbhHashSlot* hash; u32 hash_mask;
  const u32* rows; u32 n_rows;
  const u32* rprows; u32 n_rprows;
  const u32* pool;
  const u32* dr; u32 n_dr;
  const u32* code;
  const u8* const_tag; const u64* const_val;
  const u8* theap_tag; const u64* theap_val;
  const u64* gbits; u32 K;
  const u64* nfa[3]; u32 nfa_words[3];
  u32 flags;
};

struct BatchDev {
  u32 n_requests, n_tuples, n_roles, n_columns, n_strings, heap_len;
  const u32* req_u32; const u32* roles; const u32* tuple_req; const u32* tuple_action;
  const u8* col_tag; const u64* col_val;
  const u8* heap_tag; const u64* heap_val;
  const u32* str_off; const u8* str_bytes; const u8* str_flags;
  u64* gbits; // [3][n_strings], written by the resolve kernel
};

struct OutDev { u8* effect; u32* policy; u32* scope; u8* status; u64* edr; };

struct Val { u32 t; u64 v; };

struct Lane {           // per-lane evaluation state that programs can observe
  u32 req;
  u64 edr;              // effective derived roles of the scope being processed
  u32 status;           // CBH_ST_* accumulated
  bool edr_err;         // strict mode: derived roles of this scope failed to evaluate
};

struct Ctx {
  const TableDev& t; const BatchDev& b;
  i64 now_ns; u32 flags; u32 tid;
  u64* s_val; u8* s_tag;       // operand stack   [CBH_STACK_DEPTH][CBH_BLOCK]
  u64* l_val; u8* l_tag;       // locals          [CBH_MAX_LOCALS][CBH_BLOCK]
  u64* it_cont; u32* it_idx; u32* it_state; // iteration slots [CBH_MAX_ITERS][CBH_BLOCK]
};

__device__ __forceinline__ Val mk(u32 t, u64 v) { Val x; x.t = t; x.v = v; return x; }
__device__ __forceinline__ Val mk_err() { return mk(CBH_T_ERR, 0); }
__device__ __forceinline__ Val mk_bool(bool b) { return mk(CBH_T_BOOL, b ? 1u : 0u); }
__device__ __forceinline__ double as_f64(u64 v) { return __longlong_as_double((long long)v); }
__device__ __forceinline__ u64 f64_bits(double d) { return (u64)__double_as_longlong(d); }
__device__ __forceinline__ bool is_num(u32 t) { return t == CBH_T_INT || t == CBH_T_UINT || t == CBH_T_DOUBLE; }

// ---- strings -------------------------------------------------------------------------
__device__ __forceinline__ void str_span(const Ctx& c, u32 sid, const u8*& p, u32& n) {
  if (sid < c.t.K) {
    u32 o = c.t.str_off[sid];
    n = c.t.str_off[sid + 1] - o;
    p = c.t.str_bytes + o;
  } else {
    u32 i = sid - c.t.K;
    u32 o = c.b.str_off[i];
    n = c.b.str_off[i + 1] - o;
    p = c.b.str_bytes + o;
  }
}

__device__ inline int str_cmp(const Ctx& c, u32 a, u32 b) {
  if (a == b) return 0;
  const u8 *pa, *pb; u32 na, nb;
  str_span(c, a, pa, na); str_span(c, b, pb, nb);
  u32 n = na < nb ? na : nb;
  for (u32 i = 0; i < n; ++i) {
    if (pa[i] != pb[i]) return pa[i] < pb[i] ? -1 : 1;
  }
  return na < nb ? -1 : (na > nb ? 1 : 0);
}

// number of code points (CEL size(string))
__device__ inline u32 str_codepoints(const Ctx& c, u32 sid) {
  const u8* p; u32 n; str_span(c, sid, p, n);
  u32 k = 0;
  for (u32 i = 0; i < n; ++i) k += ((p[i] & 0xC0) != 0x80);
  return k;
}

__device__ inline bool str_find(const Ctx& c, u32 hay, u32 needle, int mode /*0 starts,1 ends,2 contains*/) {
  const u8 *ph, *pn; u32 nh, nn;
  str_span(c, hay, ph, nh); str_span(c, needle, pn, nn);
  if (nn > nh) return false;
  u32 lo = 0, hi = nh - nn;
  if (mode == 0) hi = 0;
  if (mode == 1) lo = hi;
  for (u32 s = lo; s <= hi; ++s) {
    u32 j = 0;
    while (j < nn && ph[s + j] == pn[j]) ++j;
    if (j == nn) return true;
  }
  return false;
}

// ---- heap ----------------------------------------------------------------------------
__device__ __forceinline__ u32 cont_sel(u64 v) { return (u32)(v >> 62); }
__device__ __forceinline__ u32 cont_off(u64 v) { return (u32)((v >> 32) & 0x3FFFFFFFu); }
__device__ __forceinline__ u32 cont_len(u64 v) { return (u32)v; }

__device__ __forceinline__ Val heap_get(const Ctx& c, u32 sel, u32 idx) {
  if (sel == CBH_HEAP_TABLE) return mk(c.t.theap_tag[idx], c.t.theap_val[idx]);
  if (sel == CBH_HEAP_BATCH) return mk(c.b.heap_tag[idx], c.b.heap_val[idx]);
  return mk(CBH_T_STRING, c.b.roles[idx]);
}

// ---- numeric comparison (exact across int64 / uint64 / double) -------------------------
// returns -1, 0, 1 or 2 (unordered: NaN)
__device__ inline int cmp_i64_f64(i64 i, double d) {
  if (d != d) return 2;
  if (d >= 9223372036854775808.0) return -1;
  if (d < -9223372036854775808.0) return 1;
  double tr = trunc(d);
  i64 t = (i64)tr;
  if (i != t) return i < t ? -1 : 1;
  double fr = d - tr;
  return fr > 0 ? -1 : (fr < 0 ? 1 : 0);
}
__device__ inline int cmp_u64_f64(u64 u, double d) {
  if (d != d) return 2;
  if (d >= 18446744073709551616.0) return -1;
  if (d < 0) return 1;
  double tr = trunc(d);
  u64 t = (u64)tr;
  if (u != t) return u < t ? -1 : 1;
  double fr = d - tr;
  return fr > 0 ? -1 : (fr < 0 ? 1 : 0);
}
__device__ inline int num_cmp(Val a, Val b) {
  if (a.t == CBH_T_DOUBLE && b.t == CBH_T_DOUBLE) {
    double x = as_f64(a.v), y = as_f64(b.v);
    if (x != x || y != y) return 2;
    return x < y ? -1 : (x > y ? 1 : 0);
  }
  if (a.t == CBH_T_INT) {
    i64 x = (i64)a.v;
    if (b.t == CBH_T_INT) { i64 y = (i64)b.v; return x < y ? -1 : (x > y ? 1 : 0); }
    if (b.t == CBH_T_UINT) { if (x < 0) return -1; u64 ux = (u64)x; return ux < b.v ? -1 : (ux > b.v ? 1 : 0); }
    return cmp_i64_f64(x, as_f64(b.v));
  }
  if (a.t == CBH_T_UINT) {
    if (b.t == CBH_T_UINT) return a.v < b.v ? -1 : (a.v > b.v ? 1 : 0);
    if (b.t == CBH_T_INT) { i64 y = (i64)b.v; if (y < 0) return 1; return a.v < (u64)y ? -1 : (a.v > (u64)y ? 1 : 0); }
    return cmp_u64_f64(a.v, as_f64(b.v));
  }
  // a double, b int/uint
  int r = (b.t == CBH_T_INT) ? cmp_i64_f64((i64)b.v, as_f64(a.v)) : cmp_u64_f64(b.v, as_f64(a.v));
  return r == 2 ? 2 : -r;
}

// scalar equality (cel-go Equal): mismatched types are simply unequal.
// `deep` receives true when a container comparison needs more than one nesting level.
__device__ inline bool scalar_equal(const Ctx& c, Val a, Val b) {
  if (is_num(a.t) && is_num(b.t)) return num_cmp(a, b) == 0;
  if (a.t != b.t) return false;
  switch (a.t) {
    case CBH_T_NULL: return true;
    case CBH_T_BOOL: case CBH_T_STRING: case CBH_T_TIMESTAMP: case CBH_T_DURATION: return a.v == b.v;
    default: return false;
  }
}

__device__ inline bool val_equal(const Ctx& c, Lane& L, Val a, Val b) {
  if (a.t == CBH_T_LIST && b.t == CBH_T_LIST) {
    u32 n = cont_len(a.v);
    if (n != cont_len(b.v)) return false;
    for (u32 i = 0; i < n; ++i) {
      Val x = heap_get(c, cont_sel(a.v), cont_off(a.v) + i);
      Val y = heap_get(c, cont_sel(b.v), cont_off(b.v) + i);
      if (x.t == CBH_T_LIST || x.t == CBH_T_MAP || y.t == CBH_T_LIST || y.t == CBH_T_MAP) {
        L.status |= CBH_ST_UNSUPPORTED;  // nested container equality is not on the device
        return false;
      }
      if (!scalar_equal(c, x, y)) return false;
    }
    return true;
  }
  if (a.t == CBH_T_MAP && b.t == CBH_T_MAP) { L.status |= CBH_ST_UNSUPPORTED; return false; }
  return scalar_equal(c, a, b);
}

// ordering: returns -1/0/1, 2 = unordered (NaN -> every comparison false), 3 = no such overload
__device__ inline int val_compare(const Ctx& c, Val a, Val b) {
  if (is_num(a.t) && is_num(b.t)) return num_cmp(a, b);
  if (a.t != b.t) return 3;
  switch (a.t) {
    case CBH_T_BOOL: return a.v < b.v ? -1 : (a.v > b.v ? 1 : 0);
    case CBH_T_STRING: return str_cmp(c, (u32)a.v, (u32)b.v);
    case CBH_T_TIMESTAMP: case CBH_T_DURATION: { i64 x = (i64)a.v, y = (i64)b.v; return x < y ? -1 : (x > y ? 1 : 0); }
    default: return 3;
  }
}

// map lookup by string key: entries are (key, value) pairs
__device__ inline bool map_find(const Ctx& c, Val m, Val key, Val& out) {
  u32 n = cont_len(m.v), off = cont_off(m.v), sel = cont_sel(m.v);
  for (u32 i = 0; i < n; ++i) {
    Val k = heap_get(c, sel, off + 2 * i);
    if (scalar_equal(c, k, key)) { out = heap_get(c, sel, off + 2 * i + 1); return true; }
  }
  return false;
}

// ---- time ----------------------------------------------------------------------------
__device__ inline i64 days_from_civil(i64 y, u32 m, u32 d) {
  y -= m <= 2;
  const i64 era = (y >= 0 ? y : y - 399) / 400;
  const u32 yoe = (u32)(y - era * 400);
  const u32 doy = (153 * (m > 2 ? m - 3 : m + 9) + 2) / 5 + d - 1;
  const u32 doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + (i64)doe - 719468;
}
__device__ __forceinline__ bool dig(u8 ch) { return ch >= '0' && ch <= '9'; }

// RFC 3339 (Go time.Parse(time.RFC3339)): returns 0 ok, 1 parse error, 2 outside the i64-ns range
__device__ inline int parse_timestamp(const u8* p, u32 n, i64& out_ns) {
  if (n < 20) return 1;
  for (int i = 0; i < 19; ++i) {
    bool d = dig(p[i]);
    if (i == 4 || i == 7) { if (p[i] != '-') return 1; }
    else if (i == 10) { if (p[i] != 'T') return 1; }
    else if (i == 13 || i == 16) { if (p[i] != ':') return 1; }
    else if (!d) return 1;
  }
  u32 Y = (p[0] - '0') * 1000 + (p[1] - '0') * 100 + (p[2] - '0') * 10 + (p[3] - '0');
  u32 M = (p[5] - '0') * 10 + (p[6] - '0'), D = (p[8] - '0') * 10 + (p[9] - '0');
  u32 h = (p[11] - '0') * 10 + (p[12] - '0'), mi = (p[14] - '0') * 10 + (p[15] - '0'), s = (p[17] - '0') * 10 + (p[18] - '0');
  if (M < 1 || M > 12 || D < 1 || h > 23 || mi > 59 || s > 59) return 1;
  const u32 mdays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  u32 dim = mdays[M - 1];
  if (M == 2 && ((Y % 4 == 0 && Y % 100 != 0) || Y % 400 == 0)) dim = 29;
  if (D > dim) return 1;
  u32 i = 19; i64 frac = 0;
  if (p[i] == '.') {
    ++i; u32 nd = 0;
    while (i < n && dig(p[i])) { if (nd < 9) { frac = frac * 10 + (p[i] - '0'); ++nd; } ++i; }
    if (nd == 0 && !(i > 20)) return 1;
    while (nd < 9) { frac *= 10; ++nd; }
  }
  if (i >= n) return 1;
  i64 off = 0;
  if (p[i] == 'Z') { if (i + 1 != n) return 1; }
  else if (p[i] == '+' || p[i] == '-') {
    if (i + 6 != n || !dig(p[i + 1]) || !dig(p[i + 2]) || p[i + 3] != ':' || !dig(p[i + 4]) || !dig(p[i + 5])) return 1;
    u32 oh = (p[i + 1] - '0') * 10 + (p[i + 2] - '0'), om = (p[i + 4] - '0') * 10 + (p[i + 5] - '0');
    if (oh > 23 || om > 59) return 1;
    off = (i64)(oh * 3600 + om * 60); if (p[i] == '-') off = -off;
  } else return 1;
  i64 secs = days_from_civil(Y, M, D) * 86400 + (i64)(h * 3600 + mi * 60 + s) - off;
  if (secs < -62135596800LL || secs > 253402300799LL) return 1;   // CEL timestamp range
  if (secs < -9223372036LL || secs > 9223372035LL) return 2;      // not representable in i64 ns
  out_ns = secs * 1000000000LL + frac;
  return 0;
}

// Go time.ParseDuration: returns 0 ok, 1 error
__device__ inline int parse_duration(const u8* p, u32 n, i64& out_ns) {
  u32 i = 0; bool neg = false;
  if (n == 0) return 1;
  if (p[0] == '-' || p[0] == '+') { neg = p[0] == '-'; i = 1; }
  if (i == n) return 1;
  if (n - i == 1 && p[i] == '0') { out_ns = 0; return 0; }
  u64 d = 0;
  while (i < n) {
    u64 v = 0, f = 0; double scale = 1; bool pre = false, post = false;
    if (!(p[i] == '.' || dig(p[i]))) return 1;
    u32 pl = i;
    while (i < n && dig(p[i])) {
      if (v > ((1ull << 63) - 1) / 10) return 1;
      v = v * 10 + (p[i] - '0');
      if (v > (1ull << 63)) return 1;
      ++i;
    }
    pre = pl != i;
    if (i < n && p[i] == '.') {
      ++i; u32 pl2 = i; bool overflow = false;
      while (i < n && dig(p[i])) {
        if (!overflow) {
          if (f > ((1ull << 63) - 1) / 10) overflow = true;
          else {
            u64 y = f * 10 + (p[i] - '0');
            if (y > (1ull << 63)) overflow = true; else { f = y; scale *= 10; }
          }
        }
        ++i;
      }
      post = pl2 != i;
    }
    if (!pre && !post) return 1;
    u32 us = i;
    while (i < n && p[i] != '.' && !dig(p[i])) ++i;
    u32 ul = i - us; if (ul == 0) return 1;
    u64 unit = 0; const u8* q = p + us;
    if (ul == 2 && q[0] == 'n' && q[1] == 's') unit = 1;
    else if (ul == 2 && q[0] == 'u' && q[1] == 's') unit = 1000;
    else if (ul == 3 && q[0] == 0xC2 && q[1] == 0xB5 && q[2] == 's') unit = 1000;
    else if (ul == 3 && q[0] == 0xCE && q[1] == 0xBC && q[2] == 's') unit = 1000;
    else if (ul == 2 && q[0] == 'm' && q[1] == 's') unit = 1000000;
    else if (ul == 1 && q[0] == 's') unit = 1000000000ull;
    else if (ul == 1 && q[0] == 'm') unit = 60000000000ull;
    else if (ul == 1 && q[0] == 'h') unit = 3600000000000ull;
    else return 1;
    if (v > (1ull << 63) / unit) return 1;
    v *= unit;
    if (f > 0) {
      v += (u64)((double)f * ((double)unit / scale));
      if (v > (1ull << 63)) return 1;
    }
    d += v;
    if (d > (1ull << 63)) return 1;
  }
  if (neg) { out_ns = (i64)(0 - d); return 0; }
  if (d > (1ull << 63) - 1) return 1;
  out_ns = (i64)d;
  return 0;
}

// dotted IPv4 "a.b.c.d" (Go netip / net.ParseIP accept no leading zeros > 1 digit... net.ParseCIDR and
// net.ParseIP reject leading zeros since Go 1.17): returns false on any deviation.
__device__ inline bool parse_ipv4(const u8* p, u32 n, u32& out) {
  u32 v = 0, parts = 0, i = 0;
  while (parts < 4) {
    if (i >= n || !dig(p[i])) return false;
    u32 x = 0, nd = 0, st = i;
    while (i < n && dig(p[i])) { x = x * 10 + (p[i] - '0'); ++nd; ++i; if (x > 255) return false; }
    if (nd > 1 && p[st] == '0') return false;
    v = (v << 8) | x; ++parts;
    if (parts < 4) { if (i >= n || p[i] != '.') return false; ++i; }
  }
  if (i != n) return false;
  out = v; return true;
}

// ---- interpreter -----------------------------------------------------------------------
#define ST(i) c.s_tag[(i) * CBH_BLOCK + c.tid]
#define SV(i) c.s_val[(i) * CBH_BLOCK + c.tid]
#define PUSHV(x) do { Val _x = (x); ST(sp) = (u8)_x.t; SV(sp) = _x.v; ++sp; } while (0)
#define TOPV(k) mk(ST(sp - 1 - (k)), SV(sp - 1 - (k)))

__device__ inline Val arith(u32 op, Val a, Val b) {
  if (a.t == CBH_T_DOUBLE && b.t == CBH_T_DOUBLE) {
    double x = as_f64(a.v), y = as_f64(b.v), r;
    switch (op) {
      case OP_ADD: r = x + y; break;
      case OP_SUB: r = x - y; break;
      case OP_MUL: r = x * y; break;
      case OP_DIV: r = x / y; break;
      default: return mk_err();
    }
    return mk(CBH_T_DOUBLE, f64_bits(r));
  }
  if (a.t == CBH_T_INT && b.t == CBH_T_INT) {
    i64 x = (i64)a.v, y = (i64)b.v, r;
    switch (op) {
      case OP_ADD: if (__builtin_add_overflow(x, y, &r)) return mk_err(); break;
      case OP_SUB: if (__builtin_sub_overflow(x, y, &r)) return mk_err(); break;
      case OP_MUL: if (__builtin_mul_overflow(x, y, &r)) return mk_err(); break;
      case OP_DIV: if (y == 0 || (x == INT64_MIN && y == -1)) return mk_err(); r = x / y; break;
      case OP_MOD: if (y == 0 || (x == INT64_MIN && y == -1)) return mk_err(); r = x % y; break;
      default: return mk_err();
    }
    return mk(CBH_T_INT, (u64)r);
  }
  if (a.t == CBH_T_UINT && b.t == CBH_T_UINT) {
    u64 x = a.v, y = b.v, r;
    switch (op) {
      case OP_ADD: if (__builtin_add_overflow(x, y, &r)) return mk_err(); break;
      case OP_SUB: if (y > x) return mk_err(); r = x - y; break;
      case OP_MUL: if (__builtin_mul_overflow(x, y, &r)) return mk_err(); break;
      case OP_DIV: if (y == 0) return mk_err(); r = x / y; break;
      case OP_MOD: if (y == 0) return mk_err(); r = x % y; break;
      default: return mk_err();
    }
    return mk(CBH_T_UINT, r);
  }
  // timestamps / durations
  i64 x = (i64)a.v, y = (i64)b.v, r;
  if (op == OP_ADD) {
    if ((a.t == CBH_T_TIMESTAMP && b.t == CBH_T_DURATION) || (a.t == CBH_T_DURATION && b.t == CBH_T_TIMESTAMP)) {
      if (__builtin_add_overflow(x, y, &r)) return mk_err();
      return mk(CBH_T_TIMESTAMP, (u64)r);
    }
    if (a.t == CBH_T_DURATION && b.t == CBH_T_DURATION) {
      if (__builtin_add_overflow(x, y, &r)) return mk_err();
      return mk(CBH_T_DURATION, (u64)r);
    }
  }
  if (op == OP_SUB) {
    if (a.t == CBH_T_TIMESTAMP && b.t == CBH_T_TIMESTAMP) { if (__builtin_sub_overflow(x, y, &r)) return mk_err(); return mk(CBH_T_DURATION, (u64)r); }
    if (a.t == CBH_T_TIMESTAMP && b.t == CBH_T_DURATION) { if (__builtin_sub_overflow(x, y, &r)) return mk_err(); return mk(CBH_T_TIMESTAMP, (u64)r); }
    if (a.t == CBH_T_DURATION && b.t == CBH_T_DURATION) { if (__builtin_sub_overflow(x, y, &r)) return mk_err(); return mk(CBH_T_DURATION, (u64)r); }
  }
  return mk_err();  // no such overload (incl. string/list concatenation: not on the device)
}

// iteration-slot state word: bit0 saw-error, bits 8.. count of true predicates
// Runs the program at `pc`.  Returns 0 = false, 1 = true, 2 = strict-mode evaluation error.
__device__ inline int run_program(const Ctx& c, Lane& L, u32 pc) {
  int sp = 0;
  const bool strict = (c.flags & CBH_F_STRICT_EVALUATION) != 0;
  for (u32 steps = 0; steps < 200000u; ++steps) {
    const u32 w = c.t.code[pc++];
    const u32 op = w & 0xFFu, a = w >> 8;
    switch (op) {
      case OP_RET: {
        return (sp > 0 && ST(sp - 1) == CBH_T_BOOL && SV(sp - 1) != 0) ? 1 : 0;
      }
      case OP_CONST: PUSHV(mk(c.t.const_tag[a], c.t.const_val[a])); break;
      case OP_COL: {
        size_t ix = (size_t)a * c.b.n_requests + L.req;
        u32 t = c.b.col_tag[ix];
        if (t == CBH_T_ABSENT) PUSHV(mk_err()); else PUSHV(mk(t, c.b.col_val[ix]));
        break;
      }
      case OP_HASCOL: {
        u32 t = c.b.col_tag[(size_t)a * c.b.n_requests + L.req];
        if (t == CBH_T_ERR) PUSHV(mk_err()); else PUSHV(mk_bool(t != CBH_T_ABSENT));
        break;
      }
      case OP_REQSTR: PUSHV(mk(CBH_T_STRING, c.b.req_u32[(size_t)a * c.b.n_requests + L.req])); break;
      case OP_ROLES: {
        u64 off = c.b.req_u32[(size_t)CBH_RQ_ROLE_OFF * c.b.n_requests + L.req];
        u64 cnt = c.b.req_u32[(size_t)CBH_RQ_ROLE_CNT * c.b.n_requests + L.req];
        PUSHV(mk(CBH_T_LIST, ((u64)CBH_HEAP_ROLES << 62) | (off << 32) | cnt));
        break;
      }
      case OP_SELECT: case OP_HASSEL: {
        Val m = TOPV(0), out;
        if (m.t != CBH_T_MAP) { ST(sp - 1) = CBH_T_ERR; break; }
        bool f = map_find(c, m, mk(CBH_T_STRING, a), out);
        if (op == OP_HASSEL) { ST(sp - 1) = CBH_T_BOOL; SV(sp - 1) = f; }
        else if (!f) ST(sp - 1) = CBH_T_ERR;
        else { ST(sp - 1) = (u8)out.t; SV(sp - 1) = out.v; }
        break;
      }
      case OP_INDEX: {
        Val i = TOPV(0), m = TOPV(1), out = mk_err(); --sp;
        if (m.t == CBH_T_ERR || i.t == CBH_T_ERR) { /* error */ }
        else if (m.t == CBH_T_MAP) { if (!map_find(c, m, i, out)) out = mk_err(); }
        else if (m.t == CBH_T_LIST && is_num(i.t)) {
          i64 k = -1;
          if (i.t == CBH_T_INT) k = (i64)i.v;
          else if (i.t == CBH_T_UINT) k = i.v < (1ull << 62) ? (i64)i.v : -1;
          else { double d = as_f64(i.v); if (d == trunc(d) && d >= 0 && d < 4e18) k = (i64)d; }
          if (k >= 0 && (u64)k < cont_len(m.v)) out = heap_get(c, cont_sel(m.v), cont_off(m.v) + (u32)k);
        }
        ST(sp - 1) = (u8)out.t; SV(sp - 1) = out.v;
        break;
      }
      case OP_EQ: case OP_NE: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        if (x.t == CBH_T_ERR || y.t == CBH_T_ERR) { ST(sp - 1) = CBH_T_ERR; break; }
        bool e = val_equal(c, L, x, y);
        ST(sp - 1) = CBH_T_BOOL; SV(sp - 1) = (op == OP_EQ) ? e : !e;
        break;
      }
      case OP_LT: case OP_LE: case OP_GT: case OP_GE: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        if (x.t == CBH_T_ERR || y.t == CBH_T_ERR) { ST(sp - 1) = CBH_T_ERR; break; }
        int r = val_compare(c, x, y);
        if (r == 3) { ST(sp - 1) = CBH_T_ERR; break; }
        bool res = false;
        if (r != 2) res = (op == OP_LT) ? r < 0 : (op == OP_LE) ? r <= 0 : (op == OP_GT) ? r > 0 : r >= 0;
        ST(sp - 1) = CBH_T_BOOL; SV(sp - 1) = res;
        break;
      }
      case OP_IN: {
        Val cont = TOPV(0), x = TOPV(1); --sp;
        if (x.t == CBH_T_ERR || cont.t == CBH_T_ERR) { ST(sp - 1) = CBH_T_ERR; break; }
        bool found = false;
        if (cont.t == CBH_T_LIST) {
          u32 n = cont_len(cont.v);
          for (u32 i = 0; i < n && !found; ++i) found = val_equal(c, L, x, heap_get(c, cont_sel(cont.v), cont_off(cont.v) + i));
        } else if (cont.t == CBH_T_MAP) {
          Val tmp; found = map_find(c, cont, x, tmp);
        } else { ST(sp - 1) = CBH_T_ERR; break; }
        ST(sp - 1) = CBH_T_BOOL; SV(sp - 1) = found;
        break;
      }
      case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_MOD: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        Val r = (x.t == CBH_T_ERR || y.t == CBH_T_ERR) ? mk_err() : arith(op, x, y);
        if (r.t == CBH_T_ERR && x.t != CBH_T_ERR && y.t != CBH_T_ERR &&
            (x.t == CBH_T_STRING || x.t == CBH_T_LIST) && x.t == y.t && op == OP_ADD)
          L.status |= CBH_ST_UNSUPPORTED;  // concatenation allocates: not on the device
        ST(sp - 1) = (u8)r.t; SV(sp - 1) = r.v;
        break;
      }
      case OP_NEG: {
        Val x = TOPV(0);
        if (x.t == CBH_T_INT) { if ((i64)x.v == INT64_MIN) ST(sp - 1) = CBH_T_ERR; else SV(sp - 1) = (u64)(-(i64)x.v); }
        else if (x.t == CBH_T_DOUBLE) SV(sp - 1) = f64_bits(-as_f64(x.v));
        else ST(sp - 1) = CBH_T_ERR;
        break;
      }
      case OP_NOT: {
        if (ST(sp - 1) == CBH_T_BOOL) SV(sp - 1) = SV(sp - 1) ? 0 : 1; else ST(sp - 1) = CBH_T_ERR;
        break;
      }
      case OP_JF: if (ST(sp - 1) == CBH_T_BOOL && SV(sp - 1) == 0) pc = a; break;
      case OP_JT: if (ST(sp - 1) == CBH_T_BOOL && SV(sp - 1) != 0) pc = a; break;
      case OP_AND: case OP_OR: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        const u64 absorbing = (op == OP_OR) ? 1 : 0;
        bool xb = x.t == CBH_T_BOOL, yb = y.t == CBH_T_BOOL;
        if ((xb && x.v == absorbing) || (yb && y.v == absorbing)) { ST(sp - 1) = CBH_T_BOOL; SV(sp - 1) = absorbing; }
        else if (xb && yb) { ST(sp - 1) = CBH_T_BOOL; SV(sp - 1) = 1 - absorbing; }
        else ST(sp - 1) = CBH_T_ERR;
        break;
      }
      case OP_JTERN: {
        Val g = TOPV(0); --sp;
        const u32 end_pc = c.t.code[pc];
        if (g.t != CBH_T_BOOL) { PUSHV(mk_err()); pc = end_pc; }
        else if (g.v) pc += 1;
        else pc = a;
        break;
      }
      case OP_JMP: pc = a; break;
      case OP_POP: --sp; break;
      case OP_LEAF: {
        Val x = TOPV(0);
        if (x.t == CBH_T_ERR) {
          L.status |= CBH_ST_CEL_ERROR;
          if (strict) return 2;
        }
        ST(sp - 1) = CBH_T_BOOL; SV(sp - 1) = (x.t == CBH_T_BOOL && x.v) ? 1 : 0;
        break;
      }
      case OP_SIZE: {
        Val x = TOPV(0);
        if (x.t == CBH_T_STRING) { ST(sp - 1) = CBH_T_INT; SV(sp - 1) = str_codepoints(c, (u32)x.v); }
        else if (x.t == CBH_T_LIST || x.t == CBH_T_MAP) { ST(sp - 1) = CBH_T_INT; SV(sp - 1) = cont_len(x.v); }
        else ST(sp - 1) = CBH_T_ERR;
        break;
      }
      case OP_STARTSWITH: case OP_ENDSWITH: case OP_CONTAINS: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        if (x.t != CBH_T_STRING || y.t != CBH_T_STRING) { ST(sp - 1) = CBH_T_ERR; break; }
        bool r = str_find(c, (u32)x.v, (u32)y.v, op == OP_STARTSWITH ? 0 : (op == OP_ENDSWITH ? 1 : 2));
        ST(sp - 1) = CBH_T_BOOL; SV(sp - 1) = r;
        break;
      }
      case OP_TIMESTAMP: {
        Val x = TOPV(0);
        if (x.t == CBH_T_TIMESTAMP) break;
        if (x.t != CBH_T_STRING) { ST(sp - 1) = CBH_T_ERR; break; }
        const u8* p; u32 n; str_span(c, (u32)x.v, p, n);
        i64 ns; int rc = parse_timestamp(p, n, ns);
        if (rc == 2) L.status |= CBH_ST_UNSUPPORTED;
        if (rc != 0) { ST(sp - 1) = CBH_T_ERR; break; }
        ST(sp - 1) = CBH_T_TIMESTAMP; SV(sp - 1) = (u64)ns;
        break;
      }
      case OP_DURATION: {
        Val x = TOPV(0);
        if (x.t == CBH_T_DURATION) break;
        if (x.t != CBH_T_STRING) { ST(sp - 1) = CBH_T_ERR; break; }
        const u8* p; u32 n; str_span(c, (u32)x.v, p, n);
        i64 ns;
        if (parse_duration(p, n, ns) != 0) { ST(sp - 1) = CBH_T_ERR; break; }
        ST(sp - 1) = CBH_T_DURATION; SV(sp - 1) = (u64)ns;
        break;
      }
      case OP_TIMESINCE: {
        Val x = TOPV(0); i64 r;
        if (x.t != CBH_T_TIMESTAMP || __builtin_sub_overflow(c.now_ns, (i64)x.v, &r)) { ST(sp - 1) = CBH_T_ERR; break; }
        ST(sp - 1) = CBH_T_DURATION; SV(sp - 1) = (u64)r;
        break;
      }
      case OP_NOW: PUSHV(mk(CBH_T_TIMESTAMP, (u64)c.now_ns)); break;
      case OP_EDRHAS: {
        if (L.edr_err) PUSHV(mk_err()); else PUSHV(mk_bool((L.edr >> a) & 1));
        break;
      }
      case OP_LOCAL: PUSHV(mk(c.l_tag[a * CBH_BLOCK + c.tid], c.l_val[a * CBH_BLOCK + c.tid])); break;
      case OP_ITER_BEGIN: {
        Val x = TOPV(0); --sp;
        const u32 w2 = c.t.code[pc++];
        if (x.t != CBH_T_LIST && x.t != CBH_T_MAP) {
          // not iterable: the macro yields an error; park it as the folded result
          c.it_state[a * CBH_BLOCK + c.tid] = 0x80000000u | (w2 & 0xFF);
          c.it_idx[a * CBH_BLOCK + c.tid] = 0; c.it_cont[a * CBH_BLOCK + c.tid] = 0;
          pc = w2 >> 8;
          break;
        }
        c.it_cont[a * CBH_BLOCK + c.tid] = x.v;  // payload (sel/off/len)
        c.it_idx[a * CBH_BLOCK + c.tid] = 0;
        c.it_state[a * CBH_BLOCK + c.tid] = (w2 & 0xFF) | ((x.t == CBH_T_MAP) ? 0x40000000u : 0u);
        break;
      }
      case OP_ITER_NEXT: {
        // next word: end_pc ; following word: local slots (v1 | v2 << 8 | nvars << 16)
        const u32 end_pc = c.t.code[pc++]; const u32 lw = c.t.code[pc++];
        const u64 cont = c.it_cont[a * CBH_BLOCK + c.tid];
        const u32 i = c.it_idx[a * CBH_BLOCK + c.tid];
        const u32 st = c.it_state[a * CBH_BLOCK + c.tid];
        if (i >= cont_len(cont)) { pc = end_pc; break; }
        const bool is_map = (st & 0x40000000u) != 0;
        const u32 l1 = lw & 0xFF, l2 = (lw >> 8) & 0xFF, nv = (lw >> 16) & 0xFF;
        Val k, v;
        if (is_map) { k = heap_get(c, cont_sel(cont), cont_off(cont) + 2 * i); v = heap_get(c, cont_sel(cont), cont_off(cont) + 2 * i + 1); }
        else { k = mk(CBH_T_INT, i); v = heap_get(c, cont_sel(cont), cont_off(cont) + i); }
        if (nv == 2) {
          c.l_tag[l1 * CBH_BLOCK + c.tid] = (u8)k.t; c.l_val[l1 * CBH_BLOCK + c.tid] = k.v;
          c.l_tag[l2 * CBH_BLOCK + c.tid] = (u8)v.t; c.l_val[l2 * CBH_BLOCK + c.tid] = v.v;
        } else {
          Val e = is_map ? k : v;
          c.l_tag[l1 * CBH_BLOCK + c.tid] = (u8)e.t; c.l_val[l1 * CBH_BLOCK + c.tid] = e.v;
        }
        c.it_idx[a * CBH_BLOCK + c.tid] = i + 1;
        break;
      }
      case OP_ITER_ACC: {
        const u32 loop_pc = c.t.code[pc++]; const u32 end_pc = c.t.code[pc++];
        Val x = TOPV(0); --sp;
        u32 st = c.it_state[a * CBH_BLOCK + c.tid];
        const u32 kind = st & 0xFF;
        pc = loop_pc;
        if (x.t != CBH_T_BOOL) {
          if (kind == IT_EXISTS_ONE) { st |= 0x80000000u; pc = end_pc; }  // errors propagate
          else st |= 0x100u;                                              // remembered, may be absorbed
        } else if (kind == IT_ALL) { if (!x.v) { st |= 0x200u; pc = end_pc; } }
        else if (kind == IT_EXISTS) { if (x.v) { st |= 0x200u; pc = end_pc; } }
        else if (x.v) st += 0x10000u;
        c.it_state[a * CBH_BLOCK + c.tid] = st;
        break;
      }
      case OP_ITER_END: {
        const u32 st = c.it_state[a * CBH_BLOCK + c.tid];
        const u32 kind = st & 0xFF;
        if (st & 0x80000000u) { PUSHV(mk_err()); break; }
        if (kind == IT_ALL) { if (st & 0x200u) PUSHV(mk_bool(false)); else if (st & 0x100u) PUSHV(mk_err()); else PUSHV(mk_bool(true)); }
        else if (kind == IT_EXISTS) { if (st & 0x200u) PUSHV(mk_bool(true)); else if (st & 0x100u) PUSHV(mk_err()); else PUSHV(mk_bool(false)); }
        else PUSHV(mk_bool(((st >> 16) & 0x3FFFu) == 1));
        break;
      }
      case OP_TOINT: {
        Val x = TOPV(0);
        if (x.t == CBH_T_INT) break;
        if (x.t == CBH_T_UINT) { if (x.v > (u64)INT64_MAX) ST(sp - 1) = CBH_T_ERR; else ST(sp - 1) = CBH_T_INT; break; }
        if (x.t == CBH_T_DOUBLE) {
          double d = as_f64(x.v);
          if (d != d || d >= 9223372036854775807.0 || d <= -9223372036854775808.0) ST(sp - 1) = CBH_T_ERR;
          else { ST(sp - 1) = CBH_T_INT; SV(sp - 1) = (u64)(i64)d; }
          break;
        }
        if (x.t == CBH_T_TIMESTAMP) { i64 ns = (i64)x.v; i64 s = ns / 1000000000LL; if (ns % 1000000000LL < 0) --s; ST(sp - 1) = CBH_T_INT; SV(sp - 1) = (u64)s; break; }
        if (x.t == CBH_T_DURATION) { ST(sp - 1) = CBH_T_INT; break; }
        if (x.t == CBH_T_STRING) L.status |= CBH_ST_UNSUPPORTED;
        ST(sp - 1) = CBH_T_ERR;
        break;
      }
      case OP_TODOUBLE: {
        Val x = TOPV(0);
        if (x.t == CBH_T_DOUBLE) break;
        if (x.t == CBH_T_INT) { ST(sp - 1) = CBH_T_DOUBLE; SV(sp - 1) = f64_bits((double)(i64)x.v); break; }
        if (x.t == CBH_T_UINT) { ST(sp - 1) = CBH_T_DOUBLE; SV(sp - 1) = f64_bits((double)x.v); break; }
        if (x.t == CBH_T_STRING) L.status |= CBH_ST_UNSUPPORTED;
        ST(sp - 1) = CBH_T_ERR;
        break;
      }
      case OP_INIPRANGE: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        if (x.t != CBH_T_STRING || y.t != CBH_T_STRING) { ST(sp - 1) = CBH_T_ERR; break; }
        const u8 *pi, *pc2; u32 ni, nc;
        str_span(c, (u32)x.v, pi, ni); str_span(c, (u32)y.v, pc2, nc);
        bool v6 = false;
        for (u32 i = 0; i < ni; ++i) v6 |= pi[i] == ':';
        for (u32 i = 0; i < nc; ++i) v6 |= pc2[i] == ':';
        if (v6) { L.status |= CBH_ST_UNSUPPORTED; ST(sp - 1) = CBH_T_ERR; break; }   // IPv6: host only
        u32 slash = nc;
        for (u32 i = 0; i < nc; ++i) if (pc2[i] == '/') { slash = i; break; }
        u32 ip, net, bits = 0, nd = 0;
        bool ok = slash < nc && parse_ipv4(pi, ni, ip) && parse_ipv4(pc2, slash, net);
        for (u32 i = slash + 1; ok && i < nc; ++i) { if (!dig(pc2[i]) || nd >= 2) ok = false; else { bits = bits * 10 + (pc2[i] - '0'); ++nd; } }
        if (ok && (nd == 0 || bits > 32 || (nd == 2 && pc2[slash + 1] == '0'))) ok = false;
        if (!ok) { ST(sp - 1) = CBH_T_ERR; break; }
        u32 mask = bits == 0 ? 0u : (0xFFFFFFFFu << (32 - bits));
        ST(sp - 1) = CBH_T_BOOL; SV(sp - 1) = ((ip & mask) == (net & mask));
        break;
      }
      case OP_HASINTERSECTION: case OP_ISSUBSET: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        if (x.t != CBH_T_LIST || y.t != CBH_T_LIST) { ST(sp - 1) = CBH_T_ERR; break; }
        // hasIntersection(x, y): some element of x is in y.  x.isSubset(y): every element of x is in y.
        bool any = false, all = true;
        for (u32 i = 0; i < cont_len(x.v); ++i) {
          Val e = heap_get(c, cont_sel(x.v), cont_off(x.v) + i);
          bool in = false;
          for (u32 j = 0; j < cont_len(y.v) && !in; ++j) in = val_equal(c, L, e, heap_get(c, cont_sel(y.v), cont_off(y.v) + j));
          any |= in; all &= in;
        }
        ST(sp - 1) = CBH_T_BOOL; SV(sp - 1) = (op == OP_HASINTERSECTION) ? any : all;
        break;
      }
      case OP_UNSUPPORTED:
      default:
        L.status |= CBH_ST_UNSUPPORTED;
        PUSHV(mk_err());
        break;
    }
  }
  L.status |= CBH_ST_UNSUPPORTED;  // step budget exhausted
  return 0;
}
