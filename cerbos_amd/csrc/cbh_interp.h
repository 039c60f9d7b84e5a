// Wave-uniform CEL bytecode interpreter.
//
// Every lane of a wave that calls run_uniform() executes the SAME program: the program
// counter, the operand-stack pointer and the opcode dispatch are wave-uniform (they live on
// the scalar unit: no exec-mask juggling per instruction), only the values are per lane.
// Lanes the program does not apply to ride along predicated off (`active` = false).
//
// Because control flow is uniform, nothing inside an expression short-circuits: `a && b`
// evaluates both sides and combines them with CEL's error absorption, a ternary evaluates
// both branches and selects, comprehensions iterate until every live lane is finished.  CEL
// is side-effect free, so results are unchanged.  The places where the reference's laziness
// IS observable - which leaf of an all/any/none tree gets evaluated, hence which CEL errors
// are recorded and which errors deny in strict mode (check.go:697-749, 823-837) - are kept
// exact with a per-lane "tree-live" predicate (TREE_* instructions).
#pragma once
#include "cbh_vm.h"

// ---- wave primitives ---------------------------------------------------------------------
// Discipline: every lane of the wave reaches every call (callers keep finished lanes
// predicated off instead of returning), so the exec mask is full at each of them.
#ifndef CBH_HOSTSIM
__device__ __forceinline__ u64 wave_ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ u32 wave_readlane(u32 v, u32 lane) { return (u32)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ u32 uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
#else
// hostsim.cpp provides wave_ballot / wave_readlane
static inline u32 uniform(u32 v) { return v; }
#endif
__device__ __forceinline__ u32 first_lane(u64 m) { return (u32)__builtin_ctzll(m); }
// Read of table data at a wave-uniform address.  The table image is immutable while kernels run,
// so it is read through the constant address space: the compiler emits s_load_dword (scalar
// cache, result in an SGPR) instead of a 64-lane vector load of one address.
#ifndef CBH_HOSTSIM
// The address goes through readfirstlane: it pins the value to SGPRs and keeps the optimiser from
// tracing the pointer back to its global-memory origin and demoting the load to a vector load.
__device__ __forceinline__ unsigned long long uniform_addr(unsigned long long v) {
  return ((unsigned long long)uniform((u32)(v >> 32)) << 32) | uniform((u32)v);
}
template <typename P> __device__ __forceinline__ u32 uload(P p) {   // P: pointer to u32 in any address space
  typedef const __attribute__((address_space(4))) u32* cptr;
  return *(cptr)uniform_addr((unsigned long long)(p));
}
#else
static inline u32 uload(const u32* p) { return *p; }
#endif
// a pointer every lane holds identically, moved to scalar registers (e.g. after it travelled through
// memory into a non-inlined function, where the compiler can no longer tell it is uniform)
template <typename P> __device__ __forceinline__ P uniform_ptr(P p) {
  const unsigned long long v = (unsigned long long)p;
  return (P)(((unsigned long long)uniform((u32)(v >> 32)) << 32) | uniform((u32)v));
}
__device__ __forceinline__ u64 wave_readlane64(u64 v, u32 lane) {
  return (u64)wave_readlane((u32)v, lane) | ((u64)wave_readlane((u32)(v >> 32), lane) << 32);
}

#define ST(i) c.s_tag[(i) * CBH_BLOCK + c.tid]
#define SV(i) c.s_val[(i) * CBH_BLOCK + c.tid]
#define PUSHV(x) do { Val _x = (x); ST(sp) = (u8)_x.t; SV(sp) = _x.v; ++sp; } while (0)
#define TOPV(k) mk(ST(sp - 1 - (k)), SV(sp - 1 - (k)))
#define SETTOP(x) do { Val _x = (x); ST(sp - 1) = (u8)_x.t; SV(sp - 1) = _x.v; } while (0)

// iteration-slot state word (per lane): bits 0..7 kind, bit 8 saw-error, bit 9 decided
// (short-circuit value reached), bit 10 lane-live-at-entry, bit 11 iterating (not exhausted /
// decided), bit 30 container is a map, bit 31 the macro itself is an error; bits 16..29 count.
#define ITS_ERRSEEN 0x100u
#define ITS_DECIDED 0x200u
#define ITS_ENTRY_LIVE 0x400u
#define ITS_RUNNING 0x800u
#define ITS_MAP 0x40000000u
#define ITS_FAIL 0x80000000u

// One record of the trace pass's log (cerbos_hip.h cbh_trace); records beyond the capacity are counted, not stored.
__device__ inline void trace_log(const OutDev& o, u32 w0, u32 w1, u32 w2, u32 w3, u64 v, u64 mask) {
  if (o.trace_cnt == nullptr) return;
#ifndef CBH_HOSTSIM
  const u32 i = __hip_atomic_fetch_add(o.trace_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  const u32 i = (*o.trace_cnt)++;
#endif
  if (i >= o.trace_cap) return;
  CBH_G u32* r = o.trace_rec + (size_t)i * CBH_TRACE_RECORD_WORDS;
  r[0] = w0; r[1] = w1; r[2] = w2; r[3] = w3; r[4] = (u32)v; r[5] = (u32)(v >> 32); r[6] = (u32)mask; r[7] = (u32)(mask >> 32);
}

// An operation fails: the slot becomes an error whose payload (cbh_vm.h mk_errc) is that of the operand that already was
// one - the left before the right, as cel-go reports them - else `code`.  Only the trace instantiation keeps payloads.
#define FAILTOP1(xv, code) do { if (TRACE) SV(sp - 1) = (xv).t == CBH_T_ERR ? (xv).v : (u64)(code); ST(sp - 1) = CBH_T_ERR; } while (0)
#define FAILTOP2(xv, yv, code) do { if (TRACE) SV(sp - 1) = (xv).t == CBH_T_ERR ? (xv).v : (yv).t == CBH_T_ERR ? (yv).v : (u64)(code); \
                                    ST(sp - 1) = CBH_T_ERR; } while (0)
// trace instantiation: the first error a comprehension absorbed, two dwords per slot behind the iteration state words
// a rope (cbh_vm.h) met by an operation that does not read ropes: the tuple is the caller's engine's, never a wrong answer
#define ROPE_UNSUPPORTED2(xv, yv) if ((xv).t == CBH_T_ROPE || (yv).t == CBH_T_ROPE) { if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }
#define IT_ERR_LO(slot) c.it_state[(CBH_MAX_ITERS + 2 * (slot)) * CBH_BLOCK + c.tid]
#define IT_ERR_HI(slot) c.it_state[(CBH_MAX_ITERS + 2 * (slot) + 1) * CBH_BLOCK + c.tid]

// Runs the program at wave-uniform `pc` for the lanes with active=true.
// Per lane result: 0 = false, 1 = true, 2 = strict-mode evaluation error (inactive lanes: 0).
// TRACE (cbh_trace_batch): error values carry which error they are, a failing leaf / variable / output expression is
// logged with `tctx` (word 1 of the record) and `tmask` (the actions an output belongs to); `tfailed` = the derived roles
// that failed in strict mode (check.go:593-610 names them in the error).
template <bool TRACE>
__device__ __forceinline__ u32 run_uniform_impl(const KernelArgs* ka, const VmLds lds, u32 req, u64 edr, bool edr_err, u32 pc, bool active,
                                                u32 tctx, u64 tmask, u64 tfailed) {
  const Ctx c = ctx_from_memory(uniform_ptr(ka), lds);   // a real call: work from the arguments in memory
  Lane L; L.req = req; L.edr = edr; L.status = 0; L.edr_err = edr_err;   // by value: see compare_op_slow
  int sp = 0;
  const bool strict = (c.flags & CBH_F_STRICT_EVALUATION) != 0;
  bool live = active;     // lane still evaluates leaves (tree-live && not aborted)
  int result = 0;
  u32 tree_saved = 0;     // bit d = `live` saved at tree depth d
  u32 tree_acc = 0;       // bit d = accumulator at tree depth d
  int tree_depth = 0;
  u32 ap = 0;             // this lane's arena: entries [0, ap) hold the lists the program has built so far (cbh_vm.h)
  pc = uniform(pc);
  const CBH_G u32* code = uniform_ptr(c.t.code);
  for (u32 steps = 0; steps < 1000000u; ++steps) {
    const u32 w = uload(&code[pc]); ++pc;
    const u32 op = w & 0xFFu, a = w >> 8;
    const bool live_in = live;        // status bits raised by a lane that is not live are dropped
    const u32 status_in = L.status;
    switch (op) {
      case OP_RET: {
        if (active && result != 2) result = (sp > 0 && ST(sp - 1) == CBH_T_BOOL && SV(sp - 1) != 0) ? 1 : 0;
        return (u32)result | (L.status << 8);
      }
      case OP_CONST: PUSHV(mk(c.t.const_tag[a], c.t.const_val[a])); break;
      case OP_COL: {
        Val v = load_operand(c, L, 1, a);
        if (TRACE && v.t == CBH_T_ERR) v = mk_errc(CBH_ERR_ATTR_MISSING, a);
        PUSHV(v);
        break;
      }
      case OP_HASCOL: {
        u32 t = c.b.col_tag[(size_t)a * c.b.n_requests + L.req];
        if (t == CBH_T_ERR) PUSHV(mk_errc(CBH_ERR_ATTR_MISSING, a)); else PUSHV(mk_bool(t != CBH_T_ABSENT));
        break;
      }
      case OP_REQSTR: PUSHV(load_operand(c, L, 2, a)); break;
      case OP_ROLES: {
        u64 off = c.b.req_u32[(size_t)CBH_RQ_ROLE_OFF * c.b.n_requests + L.req];
        u64 cnt = c.b.req_u32[(size_t)CBH_RQ_ROLE_CNT * c.b.n_requests + L.req];
        PUSHV(mk(CBH_T_LIST, ((u64)CBH_HEAP_ROLES << 62) | (off << 32) | cnt));
        break;
      }
      case OP_SELECT: case OP_HASSEL: {
        Val m = TOPV(0), out;
        if (m.t != CBH_T_MAP) { FAILTOP1(m, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        bool f = map_find(c, m, mk(CBH_T_STRING, a), out);
        if (op == OP_HASSEL) SETTOP(mk_bool(f));
        else if (!f) SETTOP(mk_errc(CBH_ERR_NO_SUCH_KEY, a));
        else SETTOP(out);
        break;
      }
      case OP_INDEX: {
        Val i = TOPV(0), m = TOPV(1), out = mk_err(); --sp;
        if (i.t == CBH_T_ROPE || m.t == CBH_T_ROPE) { if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }   // a rope as a key: not read here
        if (m.t == CBH_T_ERR) out = m;
        else if (i.t == CBH_T_ERR) out = i;
        else if (m.t == CBH_T_MAP) { if (!map_find(c, m, i, out)) out = i.t == CBH_T_STRING ? mk_errc(CBH_ERR_NO_SUCH_KEY, (u32)i.v) : mk_err(); }
        else if (m.t != CBH_T_LIST) out = mk_errc(CBH_ERR_NO_SUCH_OVERLOAD);
        else if (is_num(i.t)) {
          i64 k = -1;
          if (i.t == CBH_T_INT) k = (i64)i.v;
          else if (i.t == CBH_T_UINT) k = i.v < (1ull << 62) ? (i64)i.v : -1;
          else { double d = as_f64(i.v); if (d == trunc(d) && d >= 0 && d < 4e18) k = (i64)d; }
          if (k >= 0 && (u64)k < cont_len(m.v)) out = heap_get(c, cont_sel(m.v), cont_off(m.v) + (u32)k);
        }
        SETTOP(out);
        break;
      }
      case OP_EQ: case OP_NE: case OP_LT: case OP_LE: case OP_GT: case OP_GE: case OP_IN: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        if (x.t == CBH_T_ROPE || y.t == CBH_T_ROPE) {   // ropes are compared by content, out of line (cbh_vm.h rope_op)
          if (x.t == CBH_T_ERR || y.t == CBH_T_ERR) { FAILTOP2(x, y, 0); break; }
          if (op == OP_EQ || op == OP_NE || (op == OP_IN && x.t == CBH_T_ROPE)) {
            const SlowVal r = rope_op(ka, lds, op == OP_EQ ? ROPE_EQ : op == OP_NE ? ROPE_NE : ROPE_IN, x, y);
            SETTOP(mk(r.t, r.v));
          } else {   // an ordering of ropes, or membership IN a rope: left to the caller's engine / no such overload
            if (op != OP_IN && is_strlike(x.t) && is_strlike(y.t) && live) L.status |= CBH_ST_UNSUPPORTED;
            SETTOP(mk_errc(CBH_ERR_NO_SUCH_OVERLOAD));
          }
          break;
        }
        SETTOP(compare_op(c, L, op, x, y));
        break;
      }
      case OP_LEAF_BIN: {   // fused leaf: operands straight from columns / constants
        const u32 a0 = uload(&code[pc]), a1 = uload(&code[pc + 1]); pc += 2;
        Val x = load_operand(c, L, (a >> 8) & 0xF, a0);
        Val y = load_operand(c, L, (a >> 12) & 0xF, a1);
        Val r = compare_op(c, L, a & 0xFF, x, y);
        if (r.t == CBH_T_ERR && live) {
          L.status |= CBH_ST_CEL_ERROR;
          if (strict) { result = 2; live = false; }
        }
        PUSHV(mk_bool(r.t == CBH_T_BOOL && r.v));
        break;
      }
      case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_MOD: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        Val r = x.t == CBH_T_ERR ? x : y.t == CBH_T_ERR ? y : arith(op, x, y);
        if (r.t == CBH_T_ERR && x.t != CBH_T_ERR && y.t != CBH_T_ERR &&
            (x.t == CBH_T_STRING || x.t == CBH_T_LIST) && x.t == y.t && op == OP_ADD && live)
          L.status |= CBH_ST_UNSUPPORTED;  // concatenation allocates: not on the device
        SETTOP(r);
        break;
      }
      case OP_NEG: {
        Val x = TOPV(0);
        if (x.t == CBH_T_INT) { if ((i64)x.v == INT64_MIN) SETTOP(mk_errc(CBH_ERR_INT_OVERFLOW)); else SV(sp - 1) = (u64)(-(i64)x.v); }
        else if (x.t == CBH_T_DOUBLE) SV(sp - 1) = f64_bits(-as_f64(x.v));
        else FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD);
        break;
      }
      case OP_NOT: {
        if (ST(sp - 1) == CBH_T_BOOL) SV(sp - 1) = SV(sp - 1) ? 0 : 1; else { const Val x = TOPV(0); FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD); }
        break;
      }
      case OP_AND: case OP_OR: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        const u64 absorbing = (op == OP_OR) ? 1 : 0;
        bool xb = x.t == CBH_T_BOOL, yb = y.t == CBH_T_BOOL;
        if ((xb && x.v == absorbing) || (yb && y.v == absorbing)) SETTOP(mk_bool(absorbing != 0));
        else if (xb && yb) SETTOP(mk_bool(absorbing == 0));
        else if (TRACE && (x.t == CBH_T_ERR || !xb)) FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD);   // the first error in evaluation order
        else FAILTOP1(y, CBH_ERR_NO_SUCH_OVERLOAD);
        break;
      }
      case OP_TERN: {     // pop else, then, guard
        Val e = TOPV(0), t = TOPV(1), g = TOPV(2); sp -= 2;
        if (g.t != CBH_T_BOOL) FAILTOP1(g, CBH_ERR_NO_SUCH_OVERLOAD); else SETTOP(g.v ? t : e);
        break;
      }
      case OP_JMP: pc = a; break;
      case OP_POP: --sp; break;
      case OP_LEAF: {   // trace programs: a = trace string id of the expression + 1
        Val x = TOPV(0);
        if (x.t == CBH_T_ERR && live) {
          L.status |= CBH_ST_CEL_ERROR;
          if (TRACE && a != 0) trace_log(ka->o, L.req, CBH_TR_ERROR | tctx, a - 1, (u32)x.v, x.v, 0);
          if (strict) { result = 2; live = false; }
        }
        SETTOP(mk_bool(x.t == CBH_T_BOOL && x.v));
        break;
      }
      // ---- condition trees (check.go:697-749): children in order, later children are not
      //      evaluated (no error recorded / no strict abort) once the outcome is known
      case OP_TREE_BEGIN: {   // a = kind (0 all, 1 any, 2 none)
        const u32 bit = 1u << tree_depth;
        tree_saved = live ? (tree_saved | bit) : (tree_saved & ~bit);
        tree_acc = (a == 0) ? (tree_acc | bit) : (tree_acc & ~bit);   // all starts true, any/none false
        ++tree_depth;
        break;
      }
      case OP_TREE_ACC: {     // pop child result (plain bool)
        const u32 bit = 1u << (tree_depth - 1);
        const bool v = SV(sp - 1) != 0; --sp;
        if (live) {
          if (a == 0) { if (!v) { tree_acc &= ~bit; live = false; } }
          else if (v) { tree_acc |= bit; live = false; }
        }
        break;
      }
      case OP_TREE_END: {
        --tree_depth;
        const u32 bit = 1u << tree_depth;
        const bool acc = (tree_acc & bit) != 0;
        if (result != 2) live = (tree_saved & bit) != 0;
        PUSHV(mk_bool(a == 2 ? !acc : acc));
        break;
      }
      case OP_SIZE: {
        Val x = TOPV(0);
        if (x.t == CBH_T_STRING) SETTOP(mk(CBH_T_INT, str_codepoints(c, (u32)x.v)));
        else if (x.t == CBH_T_ROPE) { const SlowVal r = rope_op(ka, lds, ROPE_SIZE, x, x); SETTOP(mk(r.t, r.v)); }
        else if (x.t == CBH_T_LIST || x.t == CBH_T_MAP) SETTOP(mk(CBH_T_INT, cont_len(x.v)));
        else FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD);
        break;
      }
      case OP_STARTSWITH: case OP_ENDSWITH: case OP_CONTAINS: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        if (!is_strlike(x.t) || !is_strlike(y.t)) { FAILTOP2(x, y, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        if (x.t == CBH_T_ROPE || y.t == CBH_T_ROPE) {
          const SlowVal r = rope_op(ka, lds, op == OP_STARTSWITH ? ROPE_STARTS : (op == OP_ENDSWITH ? ROPE_ENDS : ROPE_CONTAINS), x, y);
          SETTOP(mk(r.t, r.v));
          break;
        }
        SETTOP(mk_bool(str_find(c, (u32)x.v, (u32)y.v, op == OP_STARTSWITH ? 0 : (op == OP_ENDSWITH ? 1 : 2))));
        break;
      }
      case OP_INDEXOF: {   // a = 0 first / 1 last
        Val y = TOPV(0), x = TOPV(1); --sp;
        ROPE_UNSUPPORTED2(x, y);
        if (x.t != CBH_T_STRING || y.t != CBH_T_STRING) { FAILTOP2(x, y, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        SETTOP(mk(CBH_T_INT, (u64)str_index_of(c, (u32)x.v, (u32)y.v, a != 0)));
        break;
      }
      case OP_STREQ_CASE: {   // a = mode a | mode b << 2 | ne << 4
        Val y = TOPV(0), x = TOPV(1); --sp;
        const u32 ma = a & 3u, mb = (a >> 2) & 3u;
        ROPE_UNSUPPORTED2(x, y);
        // a mapped side must be a string (no such overload otherwise); an unmapped side of another type is simply unequal
        if (x.t == CBH_T_ERR || y.t == CBH_T_ERR || (ma && x.t != CBH_T_STRING) || (mb && y.t != CBH_T_STRING)) { FAILTOP2(x, y, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        const bool eq = x.t == CBH_T_STRING && y.t == CBH_T_STRING && str_eq_case(c, (u32)x.v, ma, (u32)y.v, mb);
        SETTOP(mk_bool(eq != ((a >> 4) & 1u)));
        break;
      }
      case OP_MATCHES: {   // next word = the pattern's tables
        const u32 off = uload(&code[pc]); ++pc;
        Val x = TOPV(0);
        ROPE_UNSUPPORTED2(x, x);
        if (x.t != CBH_T_STRING) { FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        SETTOP(mk_bool(regex_match(c, off, (u32)x.v)));
        break;
      }
      case OP_HIER: {   // a = predicate
        Val y = TOPV(0), x = TOPV(1); --sp;
        ROPE_UNSUPPORTED2(x, y);
        if (x.t == CBH_T_STRING && y.t == CBH_T_STRING) { SETTOP(mk_bool(hier_pred(c, a, (u32)x.v, (u32)y.v))); break; }
        // hierarchy(list of strings) is valid CEL the device does not evaluate; anything else is "no such overload"
        if ((x.t == CBH_T_LIST || y.t == CBH_T_LIST) && x.t != CBH_T_ERR && y.t != CBH_T_ERR && live) L.status |= CBH_ST_UNSUPPORTED;
        FAILTOP2(x, y, CBH_ERR_NO_SUCH_OVERLOAD);
        break;
      }
      case OP_HIERCOMMON: {   // hierarchy(a).commonAncestors(hierarchy(b)) == hierarchy(c)
        Val z = TOPV(0), y = TOPV(1), x = TOPV(2); sp -= 2;
        if (x.t == CBH_T_ROPE || y.t == CBH_T_ROPE || z.t == CBH_T_ROPE) { if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }
        if (x.t == CBH_T_STRING && y.t == CBH_T_STRING && z.t == CBH_T_STRING) { SETTOP(mk_bool(hier_common_eq(c, (u32)x.v, (u32)y.v, (u32)z.v))); break; }
        if ((x.t == CBH_T_LIST || y.t == CBH_T_LIST || z.t == CBH_T_LIST) && x.t != CBH_T_ERR && y.t != CBH_T_ERR && z.t != CBH_T_ERR && live) L.status |= CBH_ST_UNSUPPORTED;
        if (x.t == CBH_T_ERR || y.t == CBH_T_ERR) { SETTOP(x.t == CBH_T_ERR ? x : y); break; }
        FAILTOP1(z, CBH_ERR_NO_SUCH_OVERLOAD);
        break;
      }
      case OP_IPFN: {   // cel-go ext.Network on request strings (a: cbh_blob.h)
        if (a == 10) {   // cidr(c).containsIP(ip): netip.ParsePrefix + netip.ParseAddr + Prefix.Contains
          Val y = TOPV(0), x = TOPV(1); --sp;
          ROPE_UNSUPPORTED2(x, y);
          if (x.t != CBH_T_STRING || y.t != CBH_T_STRING) { FAILTOP2(x, y, CBH_ERR_NO_SUCH_OVERLOAD); break; }
          gbytes pc2, pi; u32 nc, ni; u64 hi, lo;
          str_span(c, (u32)x.v, pc2, nc); str_span(c, (u32)y.v, pi, ni);
          if (parse_addr(pi, ni, hi, lo) == 0) { SETTOP(mk_err()); break; }
          const int r = ip_in_range(pi, ni, pc2, nc);
          if (r == -2 && live) L.status |= CBH_ST_UNSUPPORTED;
          if (r < 0) { SETTOP(mk_err()); break; }
          SETTOP(mk_bool(r == 1));
          break;
        }
        Val x = TOPV(0);
        ROPE_UNSUPPORTED2(x, x);
        if (x.t != CBH_T_STRING) { FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        gbytes p; u32 n; u64 hi, lo;
        str_span(c, (u32)x.v, p, n);
        const u32 fam = parse_addr(p, n, hi, lo);
        if (a == 0 || a == 8 || a == 9) { SETTOP(mk_bool(fam != 0 && (a == 0 || fam == (a == 8 ? 4u : 6u)))); break; }
        if (fam == 0) { SETTOP(mk_err()); break; }   // ip(s) / ip.isCanonical(s): "IP Address ... parse error"
        if (a == 1) { SETTOP(mk(CBH_T_INT, (u64)fam)); break; }
        if (a == 7) { SETTOP(mk_bool(addr_is_canonical(p, n, fam, hi, lo))); break; }
        SETTOP(mk_bool(addr_is(a, fam, hi, lo)));
        break;
      }
      case OP_STRVIEW: {   // a: 0 substring(from), 1 substring(from, to), 2 charAt(i), 3 trim()
        const u32 nargs = a == 1 ? 2u : (a == 3 ? 0u : 1u);
        Val to = mk(CBH_T_INT, 0), from = mk(CBH_T_INT, 0);
        if (nargs == 2) { to = TOPV(0); from = TOPV(1); } else if (nargs == 1) from = TOPV(0);
        sp -= (int)nargs;
        Val x = TOPV(0);
        if (x.t == CBH_T_ROPE) { if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }
        if (x.t != CBH_T_STRING || from.t != CBH_T_INT || to.t != CBH_T_INT) {
          if (x.t == CBH_T_ERR) break;
          if (from.t == CBH_T_ERR) { SETTOP(from); break; }
          if (to.t == CBH_T_ERR) { SETTOP(to); break; }
          FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD); break;
        }
        gbytes p; u32 n; str_span(c, (u32)x.v, p, n);
        u32 b0 = 0, b1 = n;   // the window in bytes
        bool bad = false;
        if (a == 3) {   // strings.TrimSpace: unicode.IsSpace
          auto space_at = [&](u32 i, u32& len) -> bool {   // is there a space starting at byte i?
            const u8 ch = p[i];
            len = 1;
            if (ch == ' ' || (ch >= 9 && ch <= 13)) return true;
            if (ch == 0xC2 && i + 1 < n && (p[i + 1] == 0x85 || p[i + 1] == 0xA0)) { len = 2; return true; }
            if (i + 2 < n) {
              const u8 c1 = p[i + 1], c2 = p[i + 2];
              len = 3;
              if (ch == 0xE1 && c1 == 0x9A && c2 == 0x80) return true;
              if (ch == 0xE2 && c1 == 0x80 && ((c2 >= 0x80 && c2 <= 0x8A) || c2 == 0xA8 || c2 == 0xA9 || c2 == 0xAF)) return true;
              if (ch == 0xE2 && c1 == 0x81 && c2 == 0x9F) return true;
              if (ch == 0xE3 && c1 == 0x80 && c2 == 0x80) return true;
            }
            return false;
          };
          u32 len;
          while (b0 < b1 && space_at(b0, len)) b0 += len;
          for (;;) {   // from the end: step back over one code point and test it
            if (b1 <= b0) break;
            u32 s0 = b1 - 1;
            while (s0 > b0 && (p[s0] & 0xC0u) == 0x80u) --s0;
            if (!space_at(s0, len) || s0 + len != b1) break;
            b1 = s0;
          }
        } else {
          u32 cps = 0;
          for (u32 i = 0; i < n; ++i) cps += (p[i] & 0xC0u) != 0x80u;
          const i64 f = (i64)from.v, t = a == 1 ? (i64)to.v : (a == 2 ? (i64)from.v + 1 : (i64)cps);
          if (a == 2) bad = f < 0 || f > (i64)cps;          // charAt(size) is ""
          else bad = f < 0 || f > (i64)cps || t < 0 || t > (i64)cps || f > t;
          if (!bad) {
            const u32 tt = (a == 2 && f == (i64)cps) ? (u32)cps : (u32)t;
            u32 cp = 0; b0 = n; b1 = n;
            for (u32 i = 0; i <= n; ++i) {   // byte offsets of code points f and tt
              if (i == n || (p[i] & 0xC0u) != 0x80u) {
                if (cp == (u32)f && b0 == n) b0 = i;
                if (cp == tt) { b1 = i; break; }
                ++cp;
              }
            }
          }
        }
        if (bad) { SETTOP(mk_err()); break; }
        if (b0 > CBH_ROPE_WINDOW_MAX || b1 - b0 > CBH_ROPE_WINDOW_MAX || ap >= CBH_ARENA_ENTRIES) { if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }
        arena_put(c, ap, mk(CBH_T_STRING, rope_window((u32)x.v, b0, b1 - b0)));
        SETTOP(mk_rope(ap, 1));
        ++ap;
        break;
      }
      case OP_STRREPLACE: {   // s.replace(old, new): every occurrence, left to right, not overlapping (strings.ReplaceAll)
        Val nw = TOPV(0), old = TOPV(1), x = TOPV(2); sp -= 2;
        if (x.t == CBH_T_ROPE || old.t == CBH_T_ROPE || nw.t == CBH_T_ROPE) { if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }
        if (x.t != CBH_T_STRING || old.t != CBH_T_STRING || nw.t != CBH_T_STRING) {
          if (x.t == CBH_T_ERR) break;
          if (old.t == CBH_T_ERR) { SETTOP(old); break; }
          if (nw.t == CBH_T_ERR) { SETTOP(nw); break; }
          FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD); break;
        }
        gbytes p, q; u32 n, m; str_span(c, (u32)x.v, p, n); str_span(c, (u32)old.v, q, m);
        if (m == 0 || n > CBH_ROPE_WINDOW_MAX) { if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }   // (an empty `old` matches between all code points)
        const u32 start = ap;
        u32 from = 0, i = 0;
        bool full = false;
        while (i + m <= n) {
          u32 j = 0;
          while (j < m && p[i + j] == q[j]) ++j;
          if (j < m) { ++i; continue; }
          if (ap + 2 > CBH_ARENA_ENTRIES) { full = true; break; }
          if (i > from) arena_put(c, ap++, mk(CBH_T_STRING, rope_window((u32)x.v, from, i - from)));
          arena_put(c, ap++, nw);
          i += m; from = i;
        }
        if (!full && ap + 1 > CBH_ARENA_ENTRIES) full = true;
        if (full) { ap = start; if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }
        if (from < n || ap == start) arena_put(c, ap++, mk(CBH_T_STRING, rope_window((u32)x.v, from, n - from)));
        SETTOP(mk_rope(start, ap - start));
        break;
      }
      case OP_TIMESTAMP: {
        Val x = TOPV(0);
        ROPE_UNSUPPORTED2(x, x);
        if (x.t == CBH_T_TIMESTAMP) break;
        if (x.t != CBH_T_STRING) { FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        gbytes p; u32 n; str_span(c, (u32)x.v, p, n);
        i64 ns = 0; int rc = parse_timestamp(p, n, ns);
        if (rc == 2 && live) L.status |= CBH_ST_UNSUPPORTED;
        if (rc != 0) { SETTOP(mk_err()); break; }
        SETTOP(mk(CBH_T_TIMESTAMP, (u64)ns));
        break;
      }
      case OP_DURATION: {
        Val x = TOPV(0);
        ROPE_UNSUPPORTED2(x, x);
        if (x.t == CBH_T_DURATION) break;
        if (x.t != CBH_T_STRING) { FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        gbytes p; u32 n; str_span(c, (u32)x.v, p, n);
        i64 ns = 0;
        if (parse_duration(p, n, ns) != 0) { SETTOP(mk_err()); break; }
        SETTOP(mk(CBH_T_DURATION, (u64)ns));
        break;
      }
      case OP_TIMESINCE: {
        Val x = TOPV(0); i64 r;
        if (x.t != CBH_T_TIMESTAMP) { FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        if (__builtin_sub_overflow(c.now_ns, (i64)x.v, &r)) { SETTOP(mk_err()); break; }
        SETTOP(mk(CBH_T_DURATION, (u64)r));
        break;
      }
      case OP_NOW: PUSHV(mk(CBH_T_TIMESTAMP, (u64)c.now_ns)); break;
      case OP_TS_GETTER: {    // a = getter kind; next word = the zone's offset in seconds (resolved at lowering)
        const u32 zw = uload(&code[pc]); ++pc;
        Val x = TOPV(0);
        i64 r = 0;
        i64 off_s = (i64)(int)((zw & 0x80000000u) ? zw : (zw & 0x3FFFFFFFu));
        if ((zw & 0xC0000000u) == 0x40000000u && x.t == CBH_T_TIMESTAMP) {
          // an IANA zone: [n, from_0, offset_0, ...] in the table heap, seconds, ascending, covering 1900 .. 2100 (celc.py
          // _named_zone_table); the offset in force at the timestamp = the last entry not after it
          const u32 at = zw & 0x3FFFFFFFu;
          const u32 nz = (u32)c.t.theap_val[at];
          const i64 ns = (i64)x.v;
          const i64 sec = ns >= 0 ? ns / 1000000000ll : -((-ns + 999999999ll) / 1000000000ll);
          if (nz == 0 || sec < (i64)c.t.theap_val[at + 1] || sec >= 4102444800ll) { if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }
          u32 lo = 0, hi = nz;   // entries [lo, hi): from_lo <= sec
          while (hi - lo > 1) { const u32 mid = (lo + hi) / 2; if ((i64)c.t.theap_val[at + 1 + 2 * mid] <= sec) lo = mid; else hi = mid; }
          off_s = (i64)c.t.theap_val[at + 2 + 2 * lo];
        }
        if (!ts_getter(x, a, off_s, r)) { FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        SETTOP(mk(CBH_T_INT, (u64)r));
        break;
      }
      case OP_EDRHAS: {
        if (L.edr_err) PUSHV((TRACE && (tfailed >> 56) == 0) ? mk(CBH_T_ERR, (u64)CBH_ERR_EDR_FAILED | (tfailed << 8)) : mk_err());
        else PUSHV(mk_bool((L.edr >> a) & 1));
        break;
      }
      case OP_EDREQ: {   // runtime.effectiveDerivedRoles == [constant names]
        if (L.edr_err) PUSHV((TRACE && (tfailed >> 56) == 0) ? mk(CBH_T_ERR, (u64)CBH_ERR_EDR_FAILED | (tfailed << 8)) : mk_err());
        else PUSHV(mk_bool(!(a >> 31) && L.edr == c.t.const_val[a & 0x7FFFFFFFu]));
        break;
      }
      case OP_LOCAL: PUSHV(mk(c.l_tag[a * CBH_BLOCK + c.tid], c.l_val[a * CBH_BLOCK + c.tid])); break;
      // ---- comprehensions: wave-uniform loop, per-lane progress
      case OP_ITER_BEGIN: {   // a = slot; next word = kind
        Val x = TOPV(0); --sp;
        const u32 kind = uload(&code[pc]); ++pc;
        u32 st = kind | (live ? ITS_ENTRY_LIVE : 0);
        u32 start = 0;   // filter / map: where the result list begins in the lane's arena
        if (x.t != CBH_T_LIST && x.t != CBH_T_MAP) {
          st |= ITS_FAIL;
          if (TRACE) { const u64 e = x.t == CBH_T_ERR ? x.v : (u64)CBH_ERR_NO_SUCH_OVERLOAD; IT_ERR_LO(a) = (u32)e; IT_ERR_HI(a) = (u32)(e >> 32); }
          x.v = 0;
        }
        else {
          if (live) st |= ITS_RUNNING;
          if (x.t == CBH_T_MAP) st |= ITS_MAP;   // lanes that are not live sit the loop out
          if (kind == IT_FILTER || kind == IT_MAP || kind == IT_MAP_FILTER) {
            // the result has at most as many elements as the range: reserved up front, so that whatever the loop body builds
            // lands behind it and the result stays contiguous
            const u32 n = cont_len(x.v);
            if (n > CBH_ARENA_ENTRIES - ap) { if (live) L.status |= CBH_ST_UNSUPPORTED; st |= ITS_FAIL; st &= ~ITS_RUNNING; if (TRACE) { IT_ERR_LO(a) = 0; IT_ERR_HI(a) = 0; } }
            else { start = ap; ap += n; }
          }
        }
        c.it_cont[a * CBH_BLOCK + c.tid] = x.v;
        // index word: [31:26] the result's place in the arena, [25:20] the arena's fill when the loop starts, [19:0] the index
        c.it_idx[a * CBH_BLOCK + c.tid] = (start << 26) | (ap << 20);
        c.it_state[a * CBH_BLOCK + c.tid] = st;
        break;
      }
      case OP_ITER_NEXT: {    // a = slot; next words: end_pc, locals (v1 | v2 << 8 | nvars << 16)
        const u32 end_pc = uload(&code[pc]), lw = uload(&code[pc + 1]); pc += 2;
        const u64 cont = c.it_cont[a * CBH_BLOCK + c.tid];
        const u32 iw = c.it_idx[a * CBH_BLOCK + c.tid], i = iw & 0xFFFFFu;
        u32 st = c.it_state[a * CBH_BLOCK + c.tid];
        bool more = (st & ITS_RUNNING) && i < cont_len(cont) && i < 0xFFFFFu;
        // what the previous turn of the body built is dead now - unless the body's value IS what it built (map)
        if ((st & ITS_RUNNING) && (st & 0xFF) != IT_MAP && (st & 0xFF) != IT_MAP_FILTER) ap = (iw >> 20) & 63u;
        if (!more) st &= ~ITS_RUNNING;
        c.it_state[a * CBH_BLOCK + c.tid] = st;
        live = more && (st & ITS_ENTRY_LIVE) && result != 2;
        if (wave_ballot(more) == 0) { pc = end_pc; break; }
        if (more) {
          const bool is_map = (st & ITS_MAP) != 0;
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
          c.it_idx[a * CBH_BLOCK + c.tid] = (iw & 0xFFF00000u) | (i + 1);
        }
        break;
      }
      case OP_ITER_ACC: {     // a = slot; next word = loop_pc | local of the loop variable << 30 (filter keeps its value)
        const u32 lw2 = uload(&code[pc]); ++pc;
        const u32 loop_pc = lw2 & 0x3FFFFFFFu;
        Val x = TOPV(0); --sp;
        u32 st = c.it_state[a * CBH_BLOCK + c.tid];
        Val guard = mk_bool(true);   // map with a filter: the body left the predicate under the element
        if ((st & 0xFF) == IT_MAP_FILTER) { guard = TOPV(0); --sp; }   // (wave-uniform: the kind is the program's)
        if ((st & ITS_RUNNING) && ((st & 0xFF) == IT_FILTER || (st & 0xFF) == IT_MAP || (st & 0xFF) == IT_MAP_FILTER)) {
          // filter: a predicate that is not a bool fails the macro (`pred ? acc + [x] : acc`); map: so does a failing element -
          // of a turn the predicate lets through
          const bool is_map = (st & 0xFF) != IT_FILTER;
          const bool bad_guard = guard.t != CBH_T_BOOL, skipped = !bad_guard && !guard.v;
          if (!bad_guard && !skipped && is_map && x.t == CBH_T_ROPE) {   // a rope never becomes a list element (cbh_vm.h)
            if (live) L.status |= CBH_ST_UNSUPPORTED;
            if (TRACE) { IT_ERR_LO(a) = 0; IT_ERR_HI(a) = 0; }
            st |= ITS_FAIL; st &= ~ITS_RUNNING;
          } else
          if (bad_guard || (!skipped && (x.t == CBH_T_ERR || (!is_map && x.t != CBH_T_BOOL)))) {
            const Val& bad = bad_guard ? guard : x;
            if (TRACE) { const u64 e = bad.t == CBH_T_ERR ? bad.v : (u64)CBH_ERR_NO_SUCH_OVERLOAD; IT_ERR_LO(a) = (u32)e; IT_ERR_HI(a) = (u32)(e >> 32); }
            st |= ITS_FAIL; st &= ~ITS_RUNNING;
          } else if (!skipped && (is_map || x.v)) {
            const u32 l1 = lw2 >> 30, at = (c.it_idx[a * CBH_BLOCK + c.tid] >> 26) + ((st >> 16) & 0x3FFFu);
            arena_put(c, at, is_map ? x : mk(c.l_tag[l1 * CBH_BLOCK + c.tid], c.l_val[l1 * CBH_BLOCK + c.tid]));
            st += 0x10000u;
          }
          c.it_state[a * CBH_BLOCK + c.tid] = st;
        } else
        if (st & ITS_RUNNING) {
          const u32 kind = st & 0xFF;
          if (x.t != CBH_T_BOOL) {
            if (TRACE && !(st & ITS_ERRSEEN)) {   // the first one is the one `acc && pred` keeps
              const u64 e = x.t == CBH_T_ERR ? x.v : (u64)CBH_ERR_NO_SUCH_OVERLOAD; IT_ERR_LO(a) = (u32)e; IT_ERR_HI(a) = (u32)(e >> 32);
            }
            if (kind == IT_EXISTS_ONE) { st |= ITS_FAIL; st &= ~ITS_RUNNING; }   // errors propagate
            else st |= ITS_ERRSEEN;                                               // may be absorbed
          } else if (kind == IT_ALL) { if (!x.v) { st |= ITS_DECIDED; st &= ~ITS_RUNNING; } }
          else if (kind == IT_EXISTS) { if (x.v) { st |= ITS_DECIDED; st &= ~ITS_RUNNING; } }
          else if (x.v) st += 0x10000u;
          c.it_state[a * CBH_BLOCK + c.tid] = st;
        }
        pc = loop_pc;
        break;
      }
      case OP_ITER_END: {
        const u32 st = c.it_state[a * CBH_BLOCK + c.tid];
        const u32 kind = st & 0xFF;
        if (result != 2) live = (st & ITS_ENTRY_LIVE) != 0;
        const Val ierr = mk(CBH_T_ERR, TRACE ? ((u64)IT_ERR_LO(a) | ((u64)IT_ERR_HI(a) << 32)) : 0ull);
        if (st & ITS_FAIL) { PUSHV(ierr); break; }
        if (kind == IT_FILTER || kind == IT_MAP || kind == IT_MAP_FILTER) {
          PUSHV(mk(CBH_T_LIST, ((u64)CBH_HEAP_LOCAL << 62) | ((u64)(c.it_idx[a * CBH_BLOCK + c.tid] >> 26) << 32) | ((st >> 16) & 0x3FFFu)));
          break;
        }
        if (kind == IT_ALL) { if (st & ITS_DECIDED) PUSHV(mk_bool(false)); else if (st & ITS_ERRSEEN) PUSHV(ierr); else PUSHV(mk_bool(true)); }
        else if (kind == IT_EXISTS) { if (st & ITS_DECIDED) PUSHV(mk_bool(true)); else if (st & ITS_ERRSEEN) PUSHV(ierr); else PUSHV(mk_bool(false)); }
        else PUSHV(mk_bool(((st >> 16) & 0x3FFFu) == 1));
        break;
      }
      case OP_TOINT: {
        Val x = TOPV(0);
        ROPE_UNSUPPORTED2(x, x);
        if (a == 1u) {   // uint(x) (cel-go common/types: int.go / double.go ConvertToType(UintType), overflow.go *ToUint64Checked)
          if (x.t == CBH_T_UINT) break;
          if (x.t == CBH_T_INT) { if ((i64)x.v < 0) SETTOP(mk_errc(CBH_ERR_UINT_OVERFLOW)); else ST(sp - 1) = CBH_T_UINT; break; }
          if (x.t == CBH_T_DOUBLE) {
            const double d = as_f64(x.v);
            if (d != d || d < 0 || d >= 18446744073709551616.0) SETTOP(mk_errc(CBH_ERR_UINT_OVERFLOW));
            else SETTOP(mk(CBH_T_UINT, (u64)d));
            break;
          }
          if (x.t == CBH_T_STRING && live) L.status |= CBH_ST_UNSUPPORTED;
          FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD);
          break;
        }
        if (x.t == CBH_T_INT) break;
        if (x.t == CBH_T_UINT) { if (x.v > (u64)INT64_MAX) SETTOP(mk_errc(CBH_ERR_INT_OVERFLOW)); else ST(sp - 1) = CBH_T_INT; break; }
        if (x.t == CBH_T_DOUBLE) {
          double d = as_f64(x.v);
          if (d != d || d >= 9223372036854775807.0 || d <= -9223372036854775808.0) SETTOP(mk_errc(CBH_ERR_INT_OVERFLOW));
          else SETTOP(mk(CBH_T_INT, (u64)(i64)d));
          break;
        }
        if (x.t == CBH_T_TIMESTAMP) { i64 ns = (i64)x.v; i64 s = ns / 1000000000LL; if (ns % 1000000000LL < 0) --s; SETTOP(mk(CBH_T_INT, (u64)s)); break; }
        if (x.t == CBH_T_DURATION) { ST(sp - 1) = CBH_T_INT; break; }
        if (x.t == CBH_T_STRING && live) L.status |= CBH_ST_UNSUPPORTED;
        FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD);
        break;
      }
      case OP_TODOUBLE: {
        Val x = TOPV(0);
        ROPE_UNSUPPORTED2(x, x);
        if (x.t == CBH_T_DOUBLE) break;
        if (x.t == CBH_T_INT) { SETTOP(mk(CBH_T_DOUBLE, f64_bits((double)(i64)x.v))); break; }
        if (x.t == CBH_T_UINT) { SETTOP(mk(CBH_T_DOUBLE, f64_bits((double)x.v))); break; }
        if (x.t == CBH_T_STRING && live) L.status |= CBH_ST_UNSUPPORTED;
        FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD);
        break;
      }
      case OP_INIPRANGE: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        ROPE_UNSUPPORTED2(x, y);
        if (x.t != CBH_T_STRING || y.t != CBH_T_STRING) { FAILTOP2(x, y, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        gbytes pi, pc2; u32 ni, nc;
        str_span(c, (u32)x.v, pi, ni); str_span(c, (u32)y.v, pc2, nc);
        const int r = ip_in_range(pi, ni, pc2, nc);
        if (r == -2 && live) L.status |= CBH_ST_UNSUPPORTED;
        if (r < 0) { SETTOP(mk_err()); break; }
        SETTOP(mk_bool(r == 1));
        break;
      }
      case OP_HASINTERSECTION: case OP_ISSUBSET: {
        Val y = TOPV(0), x = TOPV(1); --sp;
        if (x.t != CBH_T_LIST || y.t != CBH_T_LIST) { FAILTOP2(x, y, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        bool any = false, all = true;
        for (u32 i = 0; i < cont_len(x.v); ++i) {
          Val e = heap_get(c, cont_sel(x.v), cont_off(x.v) + i);
          bool in = false;
          for (u32 j = 0; j < cont_len(y.v) && !in; ++j) in = val_equal(c, L, e, heap_get(c, cont_sel(y.v), cont_off(y.v) + j));
          any |= in; all &= in;
        }
        SETTOP(mk_bool((op == OP_HASINTERSECTION) ? any : all));
        break;
      }
      case OP_LISTOP: {     // a = 0 intersect / 1 except / 2 concatenation: a new list in the lane's arena
        Val y = TOPV(0), x = TOPV(1); --sp;
        if (x.t != CBH_T_LIST || y.t != CBH_T_LIST) { FAILTOP2(x, y, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        // intersect iterates the SHORTER list (cerbos_lib.go:434-437 swaps its operands): order and duplicates follow it
        if (a == 0 && cont_len(x.v) > cont_len(y.v)) { const Val sw = x; x = y; y = sw; }
        const u32 nx = cont_len(x.v), ny = cont_len(y.v), need = a == 2 ? nx + ny : nx;
        if (need > CBH_ARENA_ENTRIES - ap || nx > CBH_ARENA_ENTRIES || ny > CBH_ARENA_ENTRIES) {
          if (live) L.status |= CBH_ST_UNSUPPORTED;
          SETTOP(mk_err());
          break;
        }
        const u32 start = ap;
        u32 n = 0;
        for (u32 i = 0; i < nx; ++i) {
          const Val e = heap_get(c, cont_sel(x.v), cont_off(x.v) + i);
          bool keep = true;
          if (a != 2) {
            bool in = false;
            for (u32 j = 0; j < ny && !in; ++j) in = val_equal(c, L, e, heap_get(c, cont_sel(y.v), cont_off(y.v) + j));
            keep = in == (a == 0);
          }
          if (keep) arena_put(c, start + n++, e);
        }
        if (a == 2) for (u32 j = 0; j < ny; ++j) arena_put(c, start + n++, heap_get(c, cont_sel(y.v), cont_off(y.v) + j));
        ap += need;
        SETTOP(mk(CBH_T_LIST, ((u64)CBH_HEAP_LOCAL << 62) | ((u64)start << 32) | n));
        break;
      }
      case OP_STRCAT: {     // the rope a ++ b
        Val y = TOPV(0), x = TOPV(1); --sp;
        if (!is_strlike(x.t) || !is_strlike(y.t)) { FAILTOP2(x, y, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        const u32 nx = x.t == CBH_T_ROPE ? rope_parts(x.v) : 1u, ny = y.t == CBH_T_ROPE ? rope_parts(y.v) : 1u;
        if (nx + ny > CBH_ARENA_ENTRIES - ap) { if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }   // the arena is full
        const u32 start = ap;
        for (u32 k = 0; k < nx; ++k) arena_put(c, ap++, x.t == CBH_T_ROPE ? heap_get(c, CBH_HEAP_LOCAL, cont_off(x.v) + k) : x);
        for (u32 k = 0; k < ny; ++k) arena_put(c, ap++, y.t == CBH_T_ROPE ? heap_get(c, CBH_HEAP_LOCAL, cont_off(y.v) + k) : y);
        SETTOP(mk_rope(start, nx + ny));
        break;
      }
      case OP_STRCASE: {    // a = 1 lowerAscii / 2 upperAscii: the same parts, read through the mapping (the later mapping wins)
        Val x = TOPV(0);
        if (!is_strlike(x.t)) { FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        const u32 nx = x.t == CBH_T_ROPE ? rope_parts(x.v) : 1u;
        if (nx > CBH_ARENA_ENTRIES - ap) { if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }
        const u32 start = ap;
        for (u32 k = 0; k < nx; ++k) {
          const Val part = x.t == CBH_T_ROPE ? heap_get(c, CBH_HEAP_LOCAL, cont_off(x.v) + k) : x;
          arena_put(c, ap++, mk(CBH_T_STRING, (part.v & 0xFFFFFFFFull) | ((u64)a << 32)));
        }
        SETTOP(mk_rope(start, nx));
        break;
      }
      case OP_LISTFN: {
        if (a == 1) {   // slice(start, end): a view of the same elements
          Val e = TOPV(0), b0 = TOPV(1), x = TOPV(2); sp -= 2;
          if (x.t != CBH_T_LIST || b0.t != CBH_T_INT || e.t != CBH_T_INT) { if (x.t == CBH_T_ERR) break; FAILTOP2(b0, e, CBH_ERR_NO_SUCH_OVERLOAD); break; }
          const i64 lo = (i64)b0.v, hi = (i64)e.v;
          if (lo < 0 || hi < 0 || lo > hi || hi > (i64)cont_len(x.v)) { SETTOP(mk_err()); break; }
          SETTOP(mk(CBH_T_LIST, ((u64)cont_sel(x.v) << 62) | ((u64)(cont_off(x.v) + (u32)lo) << 32) | (u64)(hi - lo)));
          break;
        }
        Val x = TOPV(0);
        if (a == 0 ? x.t != CBH_T_LIST : x.t != CBH_T_INT) { FAILTOP1(x, CBH_ERR_NO_SUCH_OVERLOAD); break; }
        const i64 n64 = a == 0 ? (i64)cont_len(x.v) : ((i64)x.v < 0 ? 0 : (i64)x.v);
        if (n64 > (i64)(CBH_ARENA_ENTRIES - ap)) { if (live) L.status |= CBH_ST_UNSUPPORTED; SETTOP(mk_err()); break; }
        const u32 n = (u32)n64, start = ap;
        for (u32 i = 0; i < n; ++i)
          arena_put(c, start + i, a == 0 ? heap_get(c, cont_sel(x.v), cont_off(x.v) + (n - 1 - i)) : mk(CBH_T_INT, i));
        ap += n;
        SETTOP(mk(CBH_T_LIST, ((u64)CBH_HEAP_LOCAL << 62) | ((u64)start << 32) | n));
        break;
      }
      case OP_VARSCOPE: {   // a = trace string id of the variable's name | mode << 23: TOS is the variable's inlined definition
        if (ST(sp - 1) == CBH_T_ERR) {
          if ((a >> 23) != 0 && !strict) {
            // a derived-role DEFINITION's variables go through evaluateVariables (check.go:612-633): the error is recorded and
            // the variable is null; every other params set through evaluatePrograms (:651-677), which leaves it unset
            if (live) L.status |= CBH_ST_CEL_ERROR;
            SETTOP(mk(CBH_T_NULL, 0));
          } else if (TRACE) SV(sp - 1) = (u64)CBH_ERR_UNDEFINED_FIELD | ((u64)(a & 0x7FFFFFu) << 8);   // unset: "undefined field '<name>'"
        }
        break;
      }
      case OP_OUT: {        // trace programs: a = trace string id of the rule's FQN; next words = rule word (id of the rule's
                            // evaluation key | not-met << 23), number of the part ("hole") of the output expression TOS is
        const u32 ek = uload(&code[pc]), part = uload(&code[pc + 1]); pc += 2;
        const Val x = TOPV(0);
        if (TRACE && live) {
          const u32 w1 = tctx | ((part & 63u) << 6);
          if (x.t == CBH_T_ERR) trace_log(ka->o, L.req, CBH_TR_OUTPUT_ERROR | w1, a, (u32)x.v, (u64)ek | ((x.v >> 32) << 32), tmask);
          else {
            trace_log(ka->o, L.req, CBH_TR_OUTPUT | w1, a, x.t | (ek << 8), x.v, tmask);
            // a list the program built lives in this lane's arena, which is gone when the log is read: its elements follow
            if (x.t == CBH_T_LIST && cont_sel(x.v) == CBH_HEAP_LOCAL)
              for (u32 i = 0; i < cont_len(x.v) && i < CBH_ARENA_ENTRIES; ++i) {
                const Val e = heap_get(c, CBH_HEAP_LOCAL, cont_off(x.v) + i);
                trace_log(ka->o, L.req, CBH_TR_OUTPUT_ELEMENT | w1, a, e.t | (ek << 8), e.v, (u64)i);
              }
          }
        }
        break;
      }
      case OP_UNSUPPORTED:
      default:
        if constexpr (TRACE) {   // OP_EDRVAL - runtime.effectiveDerivedRoles as a value (an output's part, a variable's definition) - lives
                                 // here, in the trace instantiation only: the decision kernels' interpreter sits a register or two under
                                 // its second wave (profiles/r04_kernel_resources_last_build.txt) and one more case label cost it that wave
          if (op == OP_EDRVAL) {
            if (L.edr_err) PUSHV((tfailed >> 56) == 0 ? mk(CBH_T_ERR, (u64)CBH_ERR_EDR_FAILED | (tfailed << 8)) : mk_err());
            else PUSHV(mk(CBH_T_EDRSET, L.edr));
            break;
          }
        }
        if (live) L.status |= CBH_ST_UNSUPPORTED;
        PUSHV(mk_err());
        break;
    }
    if (!live_in) L.status = status_in;
  }
  L.status |= CBH_ST_UNSUPPORTED;  // step budget exhausted
  return (u32)result | (L.status << 8);
}
#undef FAILTOP1
#undef FAILTOP2
#undef ROPE_UNSUPPORTED2
#undef IT_ERR_LO
#undef IT_ERR_HI

#ifndef CBH_HOSTSIM
__attribute__((noinline))
#endif
__device__ u32 run_uniform(const KernelArgs* ka, const VmLds lds, u32 req, u64 edr, bool edr_err, u32 pc, bool active) {
  return run_uniform_impl<false>(ka, lds, req, edr, edr_err, pc, active, 0u, 0ull, 0ull);
}
#ifndef CBH_HOSTSIM
__attribute__((noinline))
#endif
__device__ u32 run_uniform_trace(const KernelArgs* ka, const VmLds lds, u32 req, u64 edr, bool edr_err, u32 pc, bool active, u32 tctx,
                                 u64 tmask, u64 tfailed) {
  return run_uniform_impl<true>(ka, lds, req, edr, edr_err, pc, active, tctx, tmask, tfailed);
}
