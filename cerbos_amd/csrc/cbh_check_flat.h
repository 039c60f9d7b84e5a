// cbh_check_flat_kernel - the decision kernel for FLAT tables: resource policies only (no derived roles, role
// policies, parent roles or principal policies), every condition a fused leaf or a tree of them, every rule record
// decided by its class masks (cbh_blob.h CBH_ROW_F_*_BY_CLASS) - and for batches with at most four actions and
// four roles per request, outside strict mode.  The host picks it (cbh_pick_check_kernel); everything else runs
// on the general walk of cbh_check_wave.h.  Same contract, same outputs, bit for bit.
//
// What is different from the general walk: the loops are turned inside out.  check.go:208-442 walks
// roles -> scopes -> bindings; the role walks of one action are independent of each other until the final fold
// (first ALLOW wins, else the first DENY seeds, check.go:429-442), and a binding's condition does not depend on
// the role.  So here a wave walks scopes -> records ONCE, and every lane carries the state of all its role walks
// side by side: bit (4 r + k) of a 16-bit vector = "role r's walk for action k".  One visit of a record serves
// every role of every lane - match, condition, DENY / ALLOW bookkeeping are the same few instructions whatever
// the number of roles - instead of one visit per role.  The scope a walk was decided at is kept as its depth in
// the request's scope chain (four bit planes); the fold at the end picks, per action, the first role that
// allowed, else the first that denied.
//
// Evaluation errors stay exact: the reference never evaluates the roles after the one that allowed an action
// (check.go:433-436), so an error raised for bit (r, k) only counts if no role before r allowed k - decided in
// the fold, where that is known.  (Strict mode turns an error into an immediate DENY attributed to the rule's
// policy, check.go:353-356: order dependent, hence left to the general walk.)
#pragma once
#include "cbh_check_wave.h"

#define CBH_FLAT_MAX_DEPTH 16   /* scope chain entries a request can have here (the lowering checks the table) */
// Workgroups of four waves: the waves never talk to each other (each owns a quarter of the group's LDS), the
// size only quarters the number of workgroups the dispatcher has to start - ~1.7 us of a ~20 us launch at 1M tuples
// (tools/skeleton_bench.hip).  The host simulation runs one wave per block.
#if defined(CBH_FLAT_WAVES_OVERRIDE)
#define CBH_FLAT_WAVES CBH_FLAT_WAVES_OVERRIDE
#elif !defined(CBH_HOSTSIM)
#define CBH_FLAT_WAVES 4u
#else
#define CBH_FLAT_WAVES 1u
#endif
#define CBH_FLAT_THREADS (CBH_FLAT_WAVES * CBH_BLOCK)

// OR of a 64-bit value over the wave, through LDS: a wave's LDS operations execute in program order, so the zeroing
// store, the 64 atomic ORs and the read-back need no barrier on the device; the ballots are the rendezvous the host
// simulation's lane fibers need (and cost the device one scalar move each).
__device__ __forceinline__ u64 wave_or64(u64 v, u32 wave, u32 lane) {
  __shared__ unsigned long long acc[CBH_FLAT_WAVES];
  if (lane == 0) acc[wave] = 0;
  (void)wave_ballot(true);
  if (v) atomicOr(&acc[wave], (unsigned long long)v);
  (void)wave_ballot(true);
  const u64 r = acc[wave];
  (void)wave_ballot(true);
  return r;
}

#ifdef CBH_FLAT_NO_NT
#define CBH_FLAT_DMA_AUX 0
#else
#define CBH_FLAT_DMA_AUX 2   /* nt */
#endif
#ifndef CBH_FLAT_SIFT_MIN
#define CBH_FLAT_SIFT_MIN 12u         /* buckets with more records than this are sifted by class masks before any record is read */
#endif
#define CBH_FLAT_LDS_STRINGS 4096u   /* class tables of at most this many table strings are staged in LDS (2 bytes each) */
struct u32x4u { u32 x, y, z, w; };
__device__ __forceinline__ u32x4u load_u32x4(const CBH_G u32* p) {   // one 16-byte load; `p` is dword aligned
#ifndef CBH_HOSTSIM
  typedef u32 v4 __attribute__((ext_vector_type(4), aligned(4)));
  const v4 v = *(const CBH_G v4*)p;
  return u32x4u{v.x, v.y, v.z, v.w};
#else
  return u32x4u{p[0], p[1], p[2], p[3]};
#endif
}
// A rule record fetched by vector loads at a wave-uniform address, and its move into scalar registers.
#if !defined(CBH_HOSTSIM) && defined(CBH_FLAT_SCALAR_RECS)
struct VRec { TblRowFull r; };
__device__ __forceinline__ VRec vload_rec(const CBH_G u32* rows, u32 idx) { VRec v; v.r = uload_rec<TblRowFull>(rows, idx); return v; }
__device__ __forceinline__ TblRowFull rec_uniform(const VRec& v) { return v.r; }
__device__ __forceinline__ void flat_keep(u32 v) { asm volatile("" ::"v"(v)); }
#elif !defined(CBH_HOSTSIM)
typedef u32 u32v4 __attribute__((ext_vector_type(4)));
struct VRec { u32v4 a, b, c, d; };
__device__ __forceinline__ VRec vload_rec(const CBH_G u32* rows, u32 idx) {
  const CBH_G u32v4* p = (const CBH_G u32v4*)(rows + (size_t)idx * 16u);
  VRec r; r.a = p[0]; r.b = p[1]; r.c = p[2]; r.d = p[3];
  return r;
}
#define RFL(x) ((u32)__builtin_amdgcn_readfirstlane((int)(x)))
__device__ __forceinline__ TblRowFull rec_uniform(const VRec& v) {
  TblRowFull r;
  r.hot.flags = RFL(v.a.x); r.hot.cond = RFL(v.a.y); r.hot.drcond = RFL(v.a.z); r.hot.policy = RFL(v.a.w);
  r.hot.rm_lo = RFL(v.b.x); r.hot.rm_hi = RFL(v.b.y); r.hot.am_lo = RFL(v.b.z); r.hot.am_hi = RFL(v.b.w);
  r.leaf.w = RFL(v.c.x); r.leaf.a0 = RFL(v.c.y); r.leaf.a1 = RFL(v.c.z); r.leaf.ret = RFL(v.c.w);
  r.leaf.ctag = RFL(v.d.x); r.leaf.clo = RFL(v.d.y); r.leaf.chi = RFL(v.d.z); r.leaf.pad = RFL(v.d.w);
  return r;
}
#undef RFL
// a value loaded only for the load's side effect (a cache line on its way): keep the load, wait for it here
__device__ __forceinline__ void flat_keep(u32 v) { asm volatile("" ::"v"(v)); }
#else
struct VRec { TblRowFull r; };
static inline VRec vload_rec(const u32* rows, u32 idx) { VRec v; __builtin_memcpy(&v.r, rows + (size_t)idx * 16u, 64); return v; }
static inline TblRowFull rec_uniform(const VRec& v) { return v.r; }
static inline void flat_keep(u32) {}
#endif
// stores of results nobody in this kernel reads again: written through, so that the end of the kernel does not have to
// flush them out of the L2 (the dirty lines of a 1M-tuple batch are 12 MB)
template <typename T>
__device__ __forceinline__ void store_nt(CBH_G T* p, T v) {
#ifndef CBH_HOSTSIM
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// one record of CBH_SEC_DRX (cbh_blob.h CbhDrxField): a derived-role definition as the flat kernel reads it
struct __attribute__((aligned(64))) TblDrx { u32 rm_lo, rm_hi, flags, cond, name, p0, p1, p2; LeafRec leaf; };

// Dynamic LDS of the flat kernels, in dwords.  Per wave: the value planes [2][ncc][64] (low / high dword of every
// cached column; filled by asynchronous global->LDS copies), for the variant with the evaluator call the tag-word plane
// [ncc][64] the shared evaluator reads (cbh_check_wave.h fill_column_cache), the packed tags [ceil(ncc / 4)][64] (one byte
// per column, four columns to a dword: a tag is one ds_read_u8), the scope-chain scratch [max_depth][64].  Behind the
// waves' regions, once per workgroup: the two class tables.  Host and kernel size it with this one function.
struct FlatLds { u32 tagw_off, tags_off, chain_off, wave_dwords, class_bytes; };
static __host__ __device__ __forceinline__ FlatLds cbh_flat_lds(u32 ncc, u32 table_max_depth, u32 table_strings, bool with_call) {
  FlatLds l;
  const u32 depth = table_max_depth < CBH_FLAT_MAX_DEPTH ? table_max_depth : CBH_FLAT_MAX_DEPTH;
  l.tagw_off = 2u * ncc * CBH_BLOCK;
#ifdef CBH_FLAT_TAGW   /* lab variant: tags read from the tag-word plane in both kernels */
  with_call = true;
#endif
  l.tags_off = l.tagw_off + (with_call ? ncc * CBH_BLOCK : 0u);
  l.chain_off = l.tags_off + ((ncc + 3u) / 4u) * CBH_BLOCK;
  l.wave_dwords = l.chain_off + depth * CBH_BLOCK;
  l.class_bytes = table_strings <= CBH_FLAT_LDS_STRINGS ? ((2u * table_strings + 15u) & ~15u) : 0u;
  return l;
}
static inline size_t cbh_flat_lds_bytes(u32 ncc, u32 table_max_depth, u32 table_strings, bool with_call, u32 waves) {
  const FlatLds l = cbh_flat_lds(ncc, table_max_depth, table_strings, with_call);
  return (size_t)l.wave_dwords * 4u * waves + l.class_bytes;
}

// Request data is read once: loaded past the caches' retention (nt) so that it does not push the table out of the L2.
template <typename T>
__device__ __forceinline__ T load_nt(const CBH_G T* p) {
#if !defined(CBH_HOSTSIM) && !defined(CBH_FLAT_NO_NT)
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

// The attribute columns of this lane's request: values by asynchronous global->LDS copies, tags as bytes packed four
// columns to a dword (`tags` = this wave's packed-tag planes).
template <u32 N>
__device__ __forceinline__ void flat_load_tags(CBH_L u32* tags, u32 tid, u32 n, const BatchDev& b, u32 NR, u32 req) {   // columns 0 .. N-1, all loads in flight together
  u32 t[N];
#pragma unroll
  for (u32 k = 0; k < N; ++k) t[k] = load_nt(b.col_tag + ((size_t)(k < n ? k : 0u) * NR + req));   // (N rounds n up to a multiple of four: the spare slots re-read column 0)
#pragma unroll
  for (u32 g = 0; g < N / 4; ++g) tags[g * CBH_BLOCK + tid] = t[4 * g] | (t[4 * g + 1] << 8) | (t[4 * g + 2] << 16) | (t[4 * g + 3] << 24);
}
template <bool WITH_CALL>
__device__ __forceinline__ void flat_fill_columns(const Ctx& c, CBH_L u32* tags, const BatchDev& b, u32 NR, u32 req) {
  for (u32 k = 0; k < c.n_cached; ++k) {
    const size_t ix = (size_t)k * NR + req;
    const CBH_G u32* vsrc = (const CBH_G u32*)(b.col_val + ix);
#ifndef CBH_HOSTSIM
    __builtin_amdgcn_global_load_lds((const CBH_G void*)vsrc, (CBH_L void*)(c.cc + k * CBH_BLOCK), 4, 0, CBH_FLAT_DMA_AUX);
    __builtin_amdgcn_global_load_lds((const CBH_G void*)(vsrc + 1), (CBH_L void*)(c.cc + (c.n_cached + k) * CBH_BLOCK), 4, 0, CBH_FLAT_DMA_AUX);
#ifdef CBH_FLAT_TAGW
    if (true)
#else
    if (WITH_CALL)   // the shared evaluator's tag-word plane: the aligned dword holding this lane's tag byte
#endif
      __builtin_amdgcn_global_load_lds((const CBH_G void*)(b.col_tag + (ix & ~(size_t)3)), (CBH_L void*)(c.cc + (2 * c.n_cached + k) * CBH_BLOCK), 4, 0, CBH_FLAT_DMA_AUX);
#else
    c.cc[k * CBH_BLOCK + c.tid] = vsrc[0];
    c.cc[(c.n_cached + k) * CBH_BLOCK + c.tid] = vsrc[1];
    if (WITH_CALL) c.cc[(2 * c.n_cached + k) * CBH_BLOCK + c.tid] = (u32)b.col_tag[ix] << ((ix & 3u) * 8u);
#endif
  }
#ifdef CBH_FLAT_TAGW
  return;
#endif
  const u32 n = c.n_cached;   // wave-uniform: the arm that covers it, every load of it unconditional
  if (n > 12) flat_load_tags<16>(tags, c.tid, n, b, NR, req);
  else if (n > 8) flat_load_tags<12>(tags, c.tid, n, b, NR, req);
  else if (n > 4) flat_load_tags<8>(tags, c.tid, n, b, NR, req);
  else if (n > 0) flat_load_tags<4>(tags, c.tid, n, b, NR, req);
}
struct FlatCol { u32 t, lo, hi; };
struct FlatTags { const CBH_L u8* bytes; const CBH_L u32* tagw; u32 NR, req; };   // this lane's packed tag bytes: column k at bytes[(k / 4) * 256 + k % 4]
__device__ __forceinline__ FlatCol flat_col(const Ctx& c, const FlatTags& tg, u32 col) {   // `col` wave-uniform, < n_cached
  FlatCol v;
#ifdef CBH_FLAT_TAGW
  v.t = (tg.tagw[col * CBH_BLOCK] >> (((col * tg.NR + tg.req) & 3u) * 8u)) & 0xFFu;
#else
  v.t = tg.bytes[(col >> 2) * (CBH_BLOCK * 4u) + (col & 3u)];
#endif
  v.lo = c.cc[col * CBH_BLOCK + c.tid];
  v.hi = c.cc[(c.n_cached + col) * CBH_BLOCK + c.tid];
  return v;
}
// The classified fused leaves (celc.py _leaf_class 1, 2, 3, 4, 6) without a single divergent branch: every lane
// computes the answer, the error flag and the "needs the full evaluator" flag with compares and selects; which
// class it is is a wave-uniform switch.  Same answers as leaf_fast (cbh_check_wave.h), which stays the reference
// for the shapes not listed here.  Returns bit 0 = satisfied, bit 1 = CEL error (counts as not satisfied),
// bit 2 = undecided here (cross-type numerics, containers): needs the shared evaluator.
// `cls` = the leaf class, `op` = OP_EQ .. OP_IN, `ca` / `cb` = the column(s), `k0 k1 k2` = the constant: (tag, value) of a
// string / bool, the two halves of a double, or up to three string ids.  All wave-uniform.
__device__ __forceinline__ u32 flat_leaf_core(const Ctx& c, const FlatTags& tg, u32 cls, u32 op, u32 ca, u32 cb, u32 k0, u32 k1, u32 k2, u32 pid) {
  const bool want_eq = op == OP_EQ;
  switch (cls) {
    case 1: {   // column ==/!= string or bool constant (k0 = its tag, k1 = its value)
      const FlatCol x = flat_col(c, tg, ca);
      const bool err = x.t >= CBH_T_ABSENT;   // ABSENT (0xF0) or ERR (0xFF)
      const bool eq = x.t == k0 && x.lo == k1;   // other types are plainly unequal
      return err ? 2u : (u32)(eq == want_eq);
    }
    case 2: {   // column <op> double constant (k1, k2 = its halves)
      const FlatCol x = flat_col(c, tg, ca);
      const bool err = x.t >= CBH_T_ABSENT;
      const bool dbl = x.t == CBH_T_DOUBLE, othernum = x.t == CBH_T_INT || x.t == CBH_T_UINT;
      const double p = as_f64((u64)x.lo | ((u64)x.hi << 32)), q = as_f64((u64)k1 | ((u64)k2 << 32));
      const bool ordering = op != OP_EQ && op != OP_NE;
      const bool cmp = (op == OP_EQ) ? p == q : (op == OP_NE) ? p != q : (op == OP_LT) ? p < q : (op == OP_LE) ? p <= q
                     : (op == OP_GT) ? p > q : p >= q;   // NaN: every ordering false, != true
      const bool slow = !err && othernum;                          // int / uint against a double constant: cross-type numerics
      const bool overload = !dbl && !othernum && ordering;         // ordering a non-number against a number: no such overload (cbh_vm.h val_compare)
      const u32 r = dbl ? (u32)cmp : (u32)(op == OP_NE);           // a non-number is plainly unequal to a number
      return (err || overload) ? 2u : slow ? 4u : r;
    }
    case 3: {   // column ==/!= column
      const FlatCol x = flat_col(c, tg, ca), y = flat_col(c, tg, cb);
      const bool err = x.t >= CBH_T_ABSENT || y.t >= CBH_T_ABSENT;
      const bool same = x.t == y.t;
      const bool scalar = x.t < CBH_T_LIST || x.t == CBH_T_TIMESTAMP || x.t == CBH_T_DURATION;
      const bool xnum = x.t == CBH_T_INT || x.t == CBH_T_UINT || x.t == CBH_T_DOUBLE, ynum = y.t == CBH_T_INT || y.t == CBH_T_UINT || y.t == CBH_T_DOUBLE;
      const bool bits_eq = x.lo == y.lo && x.hi == y.hi;
      const bool dbl_eq = as_f64((u64)x.lo | ((u64)x.hi << 32)) == as_f64((u64)y.lo | ((u64)y.hi << 32));
      const bool eq = same && (x.t == CBH_T_DOUBLE ? dbl_eq : bits_eq);
      const bool slow = !err && ((same && !scalar) || (!same && xnum && ynum));   // containers / cross-type numeric equality
      return err ? 2u : slow ? 4u : (u32)(eq == want_eq);
    }
    case 4: {   // column ==/!= P.id
      const FlatCol x = flat_col(c, tg, ca);
      const bool err = x.t >= CBH_T_ABSENT;
      const bool eq = x.t == CBH_T_STRING && x.lo == pid;
      return err ? 2u : (u32)(eq == want_eq);
    }
    case 6: {   // column in [at most three string constants] (k0 k1 k2 = their ids, CBH_NONE pads)
      const FlatCol x = flat_col(c, tg, ca);
      const bool err = x.t >= CBH_T_ABSENT;
      const bool found = x.t == CBH_T_STRING && (x.lo == k0 || x.lo == k1 || x.lo == k2);
      return err ? 2u : (u32)found;
    }
    default: return 4u;
  }
}
// ... from the 8-dword fused-leaf record embedded in a rule record (celc.py _leaf_record)
__device__ __forceinline__ u32 flat_leaf(const Ctx& c, const FlatTags& tg, const LeafRec& lr, u32 pid) {
  const u32 a = lr.w >> 8;
  const u32 ka = (a >> 8) & 0xFu, op = a & 0xFFu;
  const u32 cls = lr.pad;
  return flat_leaf_core(c, tg, cls, op, (cls == 4u && ka != 3u) ? lr.a1 : lr.a0, lr.a1, lr.ctag, lr.clo, lr.chi, pid);
}

// A condition tree of classified leaves (cbh_blob.h CBH_ROW_F_TREE_EMBEDDED; `desc` = the descriptor in the record's leaf
// slot): the 4-bit ops in order, leaves from the strip - four dwords each {class | op << 4 | column a << 12 | column b << 20 |
// constant tag << 28, k0, k1, k2}, fetched four leaves at a time with one 16-dword scalar load - no tape reads and no
// divergent branch.  Same bookkeeping as eval_leaf_tree (cbh_check_wave.h): a leaf behind the deciding one of its level
// is not evaluated by the reference (check.go:697-749), so its error / "needs the full evaluator" flags do not count.
// Returns flat_leaf's bits for the tree.
#ifndef CBH_HOSTSIM
typedef u32 LeafQuad __attribute__((ext_vector_type(16)));   // (a vector, not a struct of an array: that one the compiler parks in scratch)
__device__ __forceinline__ LeafQuad load_quad(const CBH_G u32* code, u32 idx) {
  return *(const __attribute__((address_space(4))) LeafQuad*)uniform_addr((unsigned long long)(code + (size_t)idx * 16u));
}
#define QD(q, i) ((q)[i])
#else
struct LeafQuad { u32 d[16]; };
static inline LeafQuad load_quad(const u32* code, u32 idx) { LeafQuad q; __builtin_memcpy(&q, code + (size_t)idx * 16u, 64); return q; }
#define QD(q, i) ((q).d[i])
#endif
__device__ __forceinline__ u32 flat_tree(const Ctx& c, const FlatTags& tg, const LeafRec& desc, u32 pid) {
  bool live = true, last = false;
  u32 saved = 0, acc = 0, depth = 0, li = 0, err = 0, slow = 0;
  LeafQuad q = load_quad(c.t.code, desc.a1);   // strips start on a 16-dword boundary; a1 = that boundary's index
  for (u32 k = 0; k < 32; ++k) {
    const u32 g = k >> 3;   // wave-uniform; mask blends, see flat_col
    const u32 opw = (desc.w & (0u - (u32)(g == 0u))) | (desc.a0 & (0u - (u32)(g == 1u))) | (desc.ret & (0u - (u32)(g == 2u))) | (desc.ctag & (0u - (u32)(g == 3u)));
    const u32 op = (opw >> (4u * (k & 7u))) & 15u;
    if (op == 0) break;
    if (op == 1) {
      const u32 j = li & 3u;
      if (j == 0 && li != 0) q = load_quad(c.t.code, desc.a1 + (li >> 2));
      const u32 m0 = 0u - (u32)(j == 0u), m1 = 0u - (u32)(j == 1u), m2 = 0u - (u32)(j == 2u), m3 = 0u - (u32)(j == 3u);
      const u32 h = (QD(q, 0) & m0) | (QD(q, 4) & m1) | (QD(q, 8) & m2) | (QD(q, 12) & m3);
      const u32 k0 = (QD(q, 1) & m0) | (QD(q, 5) & m1) | (QD(q, 9) & m2) | (QD(q, 13) & m3);
      const u32 k1 = (QD(q, 2) & m0) | (QD(q, 6) & m1) | (QD(q, 10) & m2) | (QD(q, 14) & m3);
      const u32 k2 = (QD(q, 3) & m0) | (QD(q, 7) & m1) | (QD(q, 11) & m2) | (QD(q, 15) & m3);
      ++li;
      const u32 cls = h & 15u;
      // classes 1 / 2 carry (value lo, value hi) in k0 k1 and the tag in the header; class 6 three ids
      const u32 lv = cls == 6u ? flat_leaf_core(c, tg, cls, (h >> 4) & 0xFFu, (h >> 12) & 0xFFu, (h >> 20) & 0xFFu, k0, k1, k2, pid)
                               : flat_leaf_core(c, tg, cls, (h >> 4) & 0xFFu, (h >> 12) & 0xFFu, (h >> 20) & 0xFFu, h >> 28, k0, k1, pid);
      last = live && (lv & 1u) != 0;
      err |= live ? (lv & 2u) : 0u;
      slow |= live ? (lv & 4u) : 0u;
    } else if (op < 5) {          // TREE_BEGIN kind op - 2
      const u32 bit = 1u << depth;
      saved = live ? (saved | bit) : (saved & ~bit);
      acc = (op == 2) ? (acc | bit) : (acc & ~bit);
      ++depth;
    } else if (op < 8) {          // TREE_ACC kind op - 5: all - the first false decides; any / none - the first true
      const u32 bit = 1u << (depth - 1u);
      const bool decides = live && ((op == 5) ? !last : last);
      acc = decides ? ((op == 5) ? (acc & ~bit) : (acc | bit)) : acc;
      live = live && !decides;
    } else {                      // TREE_END kind op - 8
      --depth;
      const u32 bit = 1u << depth;
      live = (saved & bit) != 0;
      last = ((acc & bit) != 0) != (op == 10);
    }
  }
  return slow ? 4u : ((u32)last | err);
}

// The largest value of `v` (< 2^nbits) over the lanes with `in`, bit by bit from the top: one ballot per bit.
__device__ __forceinline__ u32 wave_max_bits(u32 v, bool in, u32 nbits) {
  u32 best = 0;
  for (u32 k = nbits; k-- > 0;) {   // wave-uniform trip count
    const bool bit = ((v >> k) & 1u) != 0;
    if (wave_ballot(in && bit) != 0) { in = in && bit; best |= 1u << k; }
  }
  return best;
}

template <bool WITH_CALL>
__device__ __forceinline__ void flat_body(const KernelArgs& ka_regs, Ctx& c) {
  const TableDev& t = ka_regs.t;
  const BatchDev& b = ka_regs.b;
  const OutDev& o = ka_regs.o;
  const u32 flags = ka_regs.flags;
  const u32 wave = threadIdx.x / CBH_BLOCK;   // which of the group's waves (c.tid is the lane within it)
#ifdef CBH_PROFILE_CYCLES   // profiling build only (tools/gpu_cycles_flat.py)
  const u64 cyc0 = __builtin_readcyclecounter();
  const u64 rt0 = __builtin_amdgcn_s_memrealtime();
  u32 dbg_rows = 0, dbg_rounds = 0, dbg_match = 0;
  u64 dbg_dir = 0, dbg_sift = 0, dbg_visit = 0, dbg_cond = 0, dbg_t = 0;
#define FLAT_T0() dbg_t = __builtin_readcyclecounter()
#define FLAT_ACC(acc) acc += __builtin_readcyclecounter() - dbg_t
#define FLAT_DBG(x) x
#else
#define FLAT_DBG(x)
#define FLAT_T0()
#define FLAT_ACC(acc)
#endif
  const u32 rix = b.req_lo + blockIdx.x * CBH_FLAT_THREADS + threadIdx.x;
  const bool valid = rix < b.req_hi;
  const u32 req = valid ? rix : b.req_lo;
  const u32 NR = b.n_requests;
#define RQ(f) load_nt(b.req_u32 + ((size_t)(f) * NR + req))
  const u32 pid = RQ(CBH_RQ_PRINCIPAL_ID), kind = RQ(CBH_RQ_KIND), r_scope = RQ(CBH_RQ_R_SCOPE), r_ver = RQ(CBH_RQ_R_VERSION);
  const u32 role_off = RQ(CBH_RQ_ROLE_OFF), act_off = RQ(CBH_RQ_ACT_OFF);
  const u32 role_cnt = valid ? RQ(CBH_RQ_ROLE_CNT) : 0, act_cnt = valid ? RQ(CBH_RQ_ACT_CNT) : 0;   // both <= 4 (host-checked)
#undef RQ
  const FlatLds lds = cbh_flat_lds(c.n_cached, t.max_depth, t.K, WITH_CALL);
  CBH_L u32* tag_planes = c.cc + lds.tags_off;   // (c.cc = this wave's region)
  flat_fill_columns<WITH_CALL>(c, tag_planes, b, NR, req);
  FlatTags tg; tg.bytes = (const CBH_L u8*)(tag_planes + c.tid); tg.tagw = c.cc + lds.tagw_off + c.tid; tg.NR = NR; tg.req = req;
  const u32 all = (1u << act_cnt) - 1u;
  // [depth][lane]: scope index at that depth of the lane's chain - in the dynamic LDS behind the column caches,
  // sized by the table's longest chain (a one-scope table pays 256 B per wave, not 4 KB: LDS sets the occupancy here)
  const u32 max_depth = t.max_depth < CBH_FLAT_MAX_DEPTH ? t.max_depth : CBH_FLAT_MAX_DEPTH;
  CBH_L u32* chain_si = c.cc + lds.chain_off;
  // actions and roles -> classes (CBH_SEC_ACTION_CLASS / CBH_SEC_ROLE_CLASS; 63 = a string no rule names).
  // A flat table has fewer than 32 classes per dimension and its masks mirror "any other string" (bit 63)
  // in bit 31 of the low dword: the match is a 1-bit field extract from ONE dword at a per-lane position.
  // Every load below is unconditional (an index that does not exist reads element 0 instead and is masked
  // afterwards): the eight id loads go out together, then the eight class loads - two round trips, not sixteen.
  u32 ac[4], rc[4], aid[4], rid[4];
  // The class tables (one byte per table string) are copied into LDS by the workgroup while the request loads are in
  // flight - a table of up to CBH_FLAT_LDS_STRINGS strings - so that the lookups below are LDS reads, not a third
  // dependent trip to memory.
  const bool cls_in_lds = t.K <= CBH_FLAT_LDS_STRINGS;
  CBH_L u8* cls_lds = (CBH_L u8*)((CBH_L u32*)cbh_dyn_lds + CBH_FLAT_WAVES * lds.wave_dwords);   // [action classes K][role classes K]
  if (cls_in_lds) {
    for (u32 i = threadIdx.x; i < t.K; i += CBH_FLAT_THREADS) { cls_lds[i] = t.action_class[i]; cls_lds[t.K + i] = t.role_class[i]; }
  }
  // Actions: a batch of four-action requests laid out back to back has ACT_OFF = 4 * request - read the four ids from
  // there with ONE 16-byte load that does not wait for ACT_OFF to arrive, and fall back to the dependent loads for the
  // lanes where the guess was wrong.
  const bool spec = b.n_tuples >= 4u;   // wave-uniform
  const u32 spec_ix = (4u * req + 4u <= b.n_tuples) ? 4u * req : 0u;
  u32x4u sp; sp.x = sp.y = sp.z = sp.w = 0;
  if (spec) sp = load_u32x4(b.tuple_action + spec_ix);
#pragma unroll
  for (u32 k = 0; k < 4; ++k) rid[k] = load_nt(b.roles + (k < role_cnt ? role_off + k : 0u));
  const bool spec_hit = spec && act_cnt == 4u && act_off == spec_ix;
  aid[0] = sp.x; aid[1] = sp.y; aid[2] = sp.z; aid[3] = sp.w;
  if (!spec_hit) {
#pragma unroll
    for (u32 k = 0; k < 4; ++k) aid[k] = b.tuple_action[k < act_cnt ? act_off + k : 0u];
  }
  const u32 kmax = t.K ? t.K - 1u : 0u;
  if (cls_in_lds) {
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
      const u32 ca = cls_lds[aid[k] < t.K ? aid[k] : kmax], cr = cls_lds[t.K + (rid[k] < t.K ? rid[k] : kmax)];
      ac[k] = (k < act_cnt && aid[k] < t.K && ca < 31u) ? ca : 31u;    // 31 = a string no rule names (cbh_blob.h)
      rc[k] = (k < role_cnt && rid[k] < t.K && cr < 31u) ? cr : 31u;
    }
  } else {
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
      const u32 ca = t.action_class[aid[k] < t.K ? aid[k] : kmax], cr = t.role_class[rid[k] < t.K ? rid[k] : kmax];
      ac[k] = (k < act_cnt && aid[k] < t.K && ca < 31u) ? ca : 31u;
      rc[k] = (k < role_cnt && rid[k] < t.K && cr < 31u) ? cr : 31u;
    }
  }
  u32 lane_ac = 0, lane_rc = 0;
  u32 walks = 0;   // bit 4r + k: role r exists and action k exists
#pragma unroll
  for (u32 k = 0; k < 4; ++k) {
    if (k < act_cnt) lane_ac |= 1u << ac[k];
    if (k < role_cnt) { lane_rc |= 1u << rc[k]; walks |= all << (4 * k); }
  }
  // classes present in the wave: a record none of them can match is skipped on the scalar unit
  const u64 wave_cls = wave_or64((u64)lane_ac | ((u64)lane_rc << 32), wave, c.tid);
  const u32 wave_ac = (u32)wave_cls, wave_rc = (u32)(wave_cls >> 32);
  FLAT_DBG(const u64 cyc1 = __builtin_readcyclecounter();)   // request fields, ids and classes have arrived

  const bool lenient = (flags & CBH_F_LENIENT_SCOPE_SEARCH) != 0;
  Lane L; L.req = req; L.edr = 0; L.status = 0; L.edr_err = false; L.pid = pid;

  u32 S = walks;                 // walks still going
  u32 has_allow = 0, allow = 0, deny = 0, err = 0, unsup = 0;
  u32 dp0 = 0, dp1 = 0, dp2 = 0, dp3 = 0;   // bit planes of the depth a walk was decided at
  // scope indices are < 2^scope_bits; the walk below merges lanes by scope, deepest (= largest index) first
  const u32 scope_bits = t.n_scopes > 1 ? 32u - (u32)__builtin_clz(t.n_scopes - 1u) : 0u;

  // a condition reference for the lanes with `active`: bit 0 satisfied, bit 1 CEL error, bit 3 outside the device subset.
  // `how`: 1 = `lr` is the fused leaf, 2 = `lr` describes a one-level tree of classified leaves (both inline: flat_leaf /
  // flat_tree), 0 = neither; what the inline code leaves open goes through the shared evaluator.
  auto leafish = [&](u32 ref, u32 how, const LeafRec& lr, bool active) -> u32 {
    u32 lv = 4u;
    if (how == 1u) lv = flat_leaf(c, tg, lr, pid);
    else if (how == 2u) lv = flat_tree(c, tg, lr, pid);
    const bool slow = active && lv == 4u;
    if (WITH_CALL) {
      // The classified leaves leave open only what needs memory (container equality) or cross-type numerics.  The
      // kernel variant for batches that can hold such values makes ONE real call into the shared evaluator here; the
      // call's register convention costs the whole kernel its occupancy, so batches whose attribute columns are all
      // null / bool / double / string / timestamp / duration (the host checks, cbh_engine.hip validate_batch) run
      // the variant compiled without it - there `slow` cannot be true.
      if (wave_ballot(slow) != 0) {
        const u32 r = eval_ref<false>(c.ka_mem, lds_of(c), req, 0, false, ref, slow);
        if (slow) lv = ((r & 0xFFu) == 1u ? 1u : 0u) | (((r >> 8) & CBH_ST_CEL_ERROR) ? 2u : 0u) | (((r >> 8) & CBH_ST_UNSUPPORTED) ? 8u : 0u);
      }
    } else if (slow) lv = 8u;   // unreachable by the host's check; loud (UNSUPPORTED), never a guessed effect
    return active ? lv : 0u;
  };

  // ---- the walk.  check.go:208-442 walks, per request, its scope chain from the request's scope up to the root and,
  // per scope, the bindings of (version, kind, scope).  Here every lane keeps the scope it stands at (`cur`, its own
  // depth in `mydepth`); each round takes the DEEPEST scope any lane stands at (scopes are numbered parents first, so
  // that is the largest index), and the lanes standing there with the leader's (version, kind) walk that one bucket
  // together, then step to their parent scope.  Lanes whose chains start at different scopes of one branch therefore
  // fall in with each other as soon as the deeper ones have climbed to the shallower ones' start: a wave holding the
  // five request scopes of one kind walks its three buckets once, not once per request scope.  Each lane still meets
  // its own scopes in chain order and a bucket's records in binding order.
  const u32 first = chain_first(t, r_scope, FLAG_RES, lenient);   // per lane (ruletable.go:848-882)
  u32 cur = first, mydepth = 0;
  bool exists = false;
  FLAT_DBG(const u64 cyc2 = cyc1 + (__builtin_readcyclecounter() - cyc1) * (u64)(wave_ballot(first != 0xFFFFFFFEu) != 0);)   // chain starts known
  for (;;) {
    // a lane goes on while it has walks to decide, and after that until it knows that some policy exists
    // (check.go:119-121, 168-170: "no policy at all" is an answer too)
    const bool active = cur != CBH_NONE && (S != 0 || !exists);
    if (wave_ballot(active) == 0) break;
    const u32 g_si = wave_max_bits(cur, active, scope_bits);
    const u64 here = wave_ballot(active && cur == g_si);
    const u32 lead = first_lane(here);
    const u32 g_ver = wave_readlane(r_ver, lead), g_k = wave_readlane(kind, lead);
    const bool ing = active && cur == g_si && r_ver == g_ver && kind == g_k;
    const bool go = wave_ballot(ing && S != 0) != 0;
    FLAT_DBG(++dbg_rounds;)
    uint4 bucket; bucket.x = bucket.y = bucket.z = bucket.w = 0;
    FLAT_T0();
    const bool have_bucket = udir_find(t, CBH_B_RESOURCE, g_ver, g_k, g_si, bucket);   // present for every resource policy (index.go:966-997)
    FLAT_DBG(if (have_bucket) { dbg_dir += (__builtin_readcyclecounter() - dbg_t) * (u64)(bucket.y != 0xFFFFFFFFu); })
    exists = exists || (ing && have_bucket);
    if (go) {
      if (ing && mydepth < max_depth) chain_si[mydepth * CBH_BLOCK + c.tid] = g_si;
      const u32 S_before = S;
      // one binding (check.go:295-414) for the lanes of this round
      auto visit = [&](u32 row, const TblRowFull& rf) {
        const TblRow& rw = rf.hot;
        FLAT_DBG(++dbg_rows;)
        if ((rw.rm_lo & wave_rc) == 0 || (rw.am_lo & wave_ac) == 0) return;   // no lane of the wave holds a class it names
        const u32 mact = ((rw.am_lo >> ac[0]) & 1u) | (((rw.am_lo >> ac[1]) & 1u) << 1) | (((rw.am_lo >> ac[2]) & 1u) << 2) | (((rw.am_lo >> ac[3]) & 1u) << 3);
        // one nibble per role (sign-extended 1-bit extracts), one bit per nibble for the actions; walks of other groups sit out
        const u32 mrole = ((0u - ((rw.rm_lo >> rc[0]) & 1u)) & 0xFu) | ((0u - ((rw.rm_lo >> rc[1]) & 1u)) & 0xF0u) |
                          ((0u - ((rw.rm_lo >> rc[2]) & 1u)) & 0xF00u) | ((0u - ((rw.rm_lo >> rc[3]) & 1u)) & 0xF000u);
        const u32 m = ing ? (mrole & (mact * 0x1111u) & S) : 0u;
        if (wave_ballot(m != 0) == 0) return;
        FLAT_DBG(++dbg_match; const u64 c0 = __builtin_readcyclecounter();)
        // the walks this record's effect applies to: all matched ones unless a condition says no.  The derived-role
        // condition comes first and the rule's own condition is evaluated only where that held (check.go:328-380);
        // each once per record and request, whatever the roles (check.go:316-340)
        u32 hit = m;
        if (rw.drcond != CBH_NONE) {
          const u32 how = (rw.flags & CBH_ROW_F_DRLEAF_EMBEDDED) ? 1u : (rw.flags & CBH_ROW_F_DRTREE_EMBEDDED) ? 2u : 0u;
          const LeafRec l2 = uload_rec<LeafRec>(t.rowleaf2, row);
          const u32 lv = leafish(rw.drcond, how, l2, hit != 0);
          err |= (lv & 2u) ? hit : 0u; unsup |= (lv & 8u) ? hit : 0u;
          hit = (lv & 1u) ? hit : 0u;
        }
        if (rw.cond != CBH_NONE && wave_ballot(hit != 0) != 0) {
          const u32 how = (rw.flags & CBH_ROW_F_LEAF_EMBEDDED) ? 1u : (rw.flags & CBH_ROW_F_TREE_EMBEDDED) ? 2u : 0u;
          const u32 lv = leafish(rw.cond, how, rf.leaf, hit != 0);
          err |= (lv & 2u) ? hit : 0u; unsup |= (lv & 8u) ? hit : 0u;
          hit = (lv & 1u) ? hit : 0u;
        }
        FLAT_DBG(dbg_cond += (__builtin_readcyclecounter() - c0) * (u64)(hit != 0xFFFFFFFFu || true);)
        if ((rw.flags & 3u) == CBH_EFFECT_ALLOW) has_allow |= hit;
        else if ((rw.flags & 3u) == CBH_EFFECT_DENY) { deny |= hit; S &= ~hit; }   // ends these walks (check.go:392-403)
      };
      if (have_bucket && bucket.y) {
        // Bindings in order.  A large bucket is taken 64 records at a time and first sifted by the (role classes, action
        // classes) pairs of its records (CBH_SEC_ROWMASK: lane j tests record j against the classes present in the
        // wave, one 8-byte load per lane) so that only records some lane can match are fetched at all; a small one is
        // read record by record.  Either way the next record to be visited is loaded while the current one is processed.
        // Records travel through VECTOR loads at a wave-uniform address (every lane gets the same 64 bytes) and are moved to
        // scalar registers when their turn comes: vector loads return in order, so two records stay in flight while a
        // third is processed and nothing else waits for them.  (Scalar loads share their counter with the LDS, and return
        // out of order: a record prefetched that way is waited for in full by the first LDS read of a condition.)
        const u32 end = bucket.x + bucket.y;
        if (bucket.y <= CBH_FLAT_SIFT_MIN) {   // a small bucket: record by record
          VRec q0 = vload_rec(t.rows, bucket.x), q1 = vload_rec(t.rows, bucket.x + 1u < end ? bucket.x + 1u : bucket.x);
          for (u32 row = bucket.x; row < end; ++row) {
            const VRec cur = q0;
            q0 = q1;
            q1 = vload_rec(t.rows, row + 2u < end ? row + 2u : end - 1u);
            const TblRowFull rf = rec_uniform(cur);
            FLAT_DBG(const u64 v0 = __builtin_readcyclecounter() * (u64)(rf.hot.flags != 0xFFFFFFFFu);)
            visit(row, rf);
            FLAT_DBG(dbg_visit += __builtin_readcyclecounter() - v0;)
          }
        } else {
          for (u32 base = bucket.x; base < end; base += 64u) {
            FLAT_T0();
            const u32 n_here = end - base < 64u ? end - base : 64u;
            const u32 mine = base + (c.tid < n_here ? c.tid : 0u);
            const u32 rm = t.rowmask[2u * (size_t)mine], am = t.rowmask[2u * (size_t)mine + 1u];
            const bool pass = c.tid < n_here && (rm & wave_rc) != 0 && (am & wave_ac) != 0;
            u64 vis = wave_ballot(pass);
            FLAT_DBG(dbg_sift += (__builtin_readcyclecounter() - dbg_t) * (u64)(vis != 0xFFFFFFFFFFFFFFFEull);)
            if (vis == 0) continue;
            // the records that passed: their lines are pulled towards the L2 together now (one touch per lane), the
            // loads below then find them there instead of each paying a trip to memory in turn
            const u32 touched = pass ? t.rows[(size_t)mine * 16u] : 0u;
            u32 n_left = (u32)__builtin_popcountll(vis);
            u32 ja = (u32)__builtin_ctzll(vis); vis &= vis - 1ull;
            VRec qa = vload_rec(t.rows, base + ja);
            u32 jb = ja; VRec qb = qa;
            if (vis) { jb = (u32)__builtin_ctzll(vis); vis &= vis - 1ull; qb = vload_rec(t.rows, base + jb); }
            while (n_left) {
              const VRec cur = qa;
              const u32 row = base + ja;
              qa = qb; ja = jb;
              if (vis) { jb = (u32)__builtin_ctzll(vis); vis &= vis - 1ull; qb = vload_rec(t.rows, base + jb); }
              const TblRowFull rf = rec_uniform(cur);
              FLAT_DBG(const u64 v0 = __builtin_readcyclecounter() * (u64)(rf.hot.flags != 0xFFFFFFFFu);)
              visit(row, rf);
              FLAT_DBG(dbg_visit += __builtin_readcyclecounter() - v0;)
              --n_left;
            }
            flat_keep(touched);
          }
        }
      }
      const u32 ha = ing ? (has_allow & S) : 0u;   // check.go:416-425
      const u32 sp = (uload(&t.scope_flags[g_si]) >> 2) & 3u;
      if (sp == SP_REQUIRE_CONSENT) has_allow &= ~ha;
      else if (sp == SP_OVERRIDE_PARENT) { allow |= ha; S &= ~ha; }
      const u32 newly = S_before & ~S;
      dp0 |= (mydepth & 1u) ? newly : 0u; dp1 |= (mydepth & 2u) ? newly : 0u; dp2 |= (mydepth & 4u) ? newly : 0u; dp3 |= (mydepth & 8u) ? newly : 0u;
    }
    const u32 up = uchain_next(t, uload(&t.scope_parent[g_si]), FLAG_RES);   // check.go:231
    if (ing) { cur = (mydepth + 1u < max_depth) ? up : CBH_NONE; ++mydepth; }
  }

  FLAT_DBG(const u64 cyc3 = __builtin_readcyclecounter();)   // the walk is over
  // ---- the fold (check.go:429-442), per action: the first role that allowed, else the first role that denied
  const bool decided = first == CBH_NONE || !exists;   // nothing to evaluate: "NO_MATCH" (check.go:119-121, 168-170)
  const u32 pol_none = (u32)(decided ? CBH_P_NO_MATCH : (role_cnt ? CBH_P_RESOURCE : CBH_P_EMPTY)) << 28 | ((!decided && role_cnt) ? first : 0u);
  const u32 pol_hit = ((u32)CBH_P_RESOURCE << 28) | first;
  u32 eff4 = 0, st4 = 0, pol[4], scp[4];
#pragma unroll
  for (u32 k = 0; k < 4; ++k) {
    const u32 ak = (allow >> k) & 0x1111u, dk = (deny >> k) & 0x1111u;
    const u32 win = ak ? (ak & (0u - ak)) : (dk & (0u - dk));   // lowest role bit of the deciding kind
    const u32 wb = win << k;                                     // back at its position 4r + k
    const u32 d = ((dp0 & wb) ? 1u : 0u) | ((dp1 & wb) ? 2u : 0u) | ((dp2 & wb) ? 4u : 0u) | ((dp3 & wb) ? 8u : 0u);
    pol[k] = win ? pol_hit : pol_none;
    scp[k] = CBH_NONE;
    if (win && k < act_cnt) scp[k] = chain_si[d * CBH_BLOCK + c.tid];
    eff4 |= (u32)(ak ? CBH_EFFECT_ALLOW : CBH_EFFECT_DENY) << (8 * k);   // NO_MATCH -> DENY (check.go:451-453)
    // an evaluation the reference would not have made - a role after the one that allowed - does not count
    const u32 seen = ak ? (((ak & (0u - ak)) << 1) - 1u) : 0xFFFFu;
    const u32 ek = (err >> k) & 0x1111u & seen, uk = (unsup >> k) & 0x1111u & seen;
    st4 |= (u32)(uk ? CBH_ST_UNSUPPORTED : (ek ? CBH_ST_CEL_ERROR : CBH_ST_OK)) << (8 * k);
  }

  // ---- effective derived roles (check.go:237-282): the definitions of a scope's policy are evaluated when a role
  // walk REACHES that scope - a walk the reference really makes, i.e. not one of a role after the role that
  // allowed its action.  Known now: a decided walk reached the scopes up to the one that decided it, an undecided
  // one the whole chain.  The chain is walked a second time for the definitions alone.
  u64 edr = 0;
  if ((flags & CBH_F_WANT_DERIVED_ROLES) && t.n_dr) {
    u32 legit = 0;
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
      const u32 ak = (allow >> k) & 0x1111u;
      const u32 seen = ak ? (((ak & (0u - ak)) << 1) - 1u) : 0xFFFFu;
      legit |= ((walks >> k) & 0x1111u & seen) << k;
    }
    const u32 done = allow | deny;
    u32 reach = 0;   // deepest chain position a legitimate walk reached, + 1 (0 = none)
    if (legit & ~done) reach = CBH_FLAT_MAX_DEPTH;
    else if (legit) {   // maximum over the decided walks, from the bit planes (most significant plane first)
      u32 cand = legit, d = 0;
      u32 tp = cand & dp3; if (tp) { cand = tp; d |= 8u; }
      tp = cand & dp2; if (tp) { cand = tp; d |= 4u; }
      tp = cand & dp1; if (tp) { cand = tp; d |= 2u; }
      tp = cand & dp0; if (tp) { cand = tp; d |= 1u; }
      reach = d + 1u;
    }
    bool derr = false, dr_unsup = false;
    u32 cur2 = first, d2 = 0;   // the same merged climb as the walk above, for the scopes a legitimate walk reached
    for (;;) {
      const bool active = cur2 != CBH_NONE && d2 < reach;
      if (wave_ballot(active) == 0) break;
      const u32 g_si = wave_max_bits(cur2, active, scope_bits);
      const u32 lead = first_lane(wave_ballot(active && cur2 == g_si));
      const u32 g_ver = wave_readlane(r_ver, lead), g_k = wave_readlane(kind, lead);
      const bool ing = active && cur2 == g_si && r_ver == g_ver && kind == g_k;
      uint4 bucket; bucket.x = bucket.y = bucket.z = bucket.w = 0;
      if (udir_find(t, CBH_B_RESOURCE, g_ver, g_k, g_si, bucket)) {
        for (u32 d = bucket.z; d < bucket.z + bucket.w; ++d) {
          const TblDrx dx = uload_rec<TblDrx>(t.drx, d);
          const bool applies = ing && (dx.rm_lo & lane_rc) != 0;   // parent roles x the request's roles (check.go:244)
          if (wave_ballot(applies) == 0) continue;
          u32 lv = 1u;
          if (dx.cond != CBH_NONE) lv = leafish(dx.cond, dx.flags & 3u, dx.leaf, applies);
          if (applies) { if (lv & 1u) edr |= 1ull << dx.name; derr = derr || (lv & 2u) != 0; dr_unsup = dr_unsup || (lv & 8u) != 0; }
        }
      }
      const u32 up = uchain_next(t, uload(&t.scope_parent[g_si]), FLAG_RES);
      if (ing) { cur2 = up; ++d2; }
    }
    if (derr) st4 |= 0x01010101u & ~((st4 >> 1) & 0x01010101u);   // evaluation errors are a per-request fact: every action that is not UNSUPPORTED
    if (dr_unsup) st4 = 0x02020202u;
  }

#ifdef CBH_PROFILE_CYCLES
  if (flags & CBH_F_DEBUG_CYCLES) {   // policy / scope words <- phase cycles, wall-clock (100 MHz) start / end, visit counts
    const u64 cyc4 = __builtin_readcyclecounter();
    if (flags & 0x200u) {   // second view: where the walk's cycles went
      pol[0] = (u32)(cyc3 - cyc2); pol[1] = (u32)dbg_dir; pol[2] = (u32)dbg_sift; pol[3] = (u32)dbg_visit;
      scp[0] = (u32)dbg_cond; scp[1] = dbg_match; scp[2] = dbg_rows; scp[3] = dbg_rounds;
    } else {
      pol[0] = (u32)(cyc1 - cyc0); pol[1] = (u32)(cyc2 - cyc1); pol[2] = (u32)(cyc3 - cyc2); pol[3] = (u32)(cyc4 - cyc3);
      scp[0] = (u32)rt0; scp[1] = (u32)__builtin_amdgcn_s_memrealtime(); scp[2] = dbg_rows; scp[3] = dbg_rounds;
    }
  }
#endif
  const bool packed = valid && act_cnt == 4 && (act_off & 3u) == 0;
  if (packed) {
#ifndef CBH_HOSTSIM
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#else
    struct u32x4 { u32 x, y, z, w; };
#endif
    if (o.edr) store_nt(o.edr + req, edr);
    store_nt((CBH_G u32*)(o.effect + act_off), eff4);
    if (o.status) store_nt((CBH_G u32*)(o.status + act_off), st4);
    if (o.policy) { u32x4 v; v.x = pol[0]; v.y = pol[1]; v.z = pol[2]; v.w = pol[3]; store_nt((CBH_G u32x4*)(o.policy + act_off), v); }
    if (o.scope) { u32x4 v; v.x = scp[0]; v.y = scp[1]; v.z = scp[2]; v.w = scp[3]; store_nt((CBH_G u32x4*)(o.scope + act_off), v); }
  } else if (valid) {
    if (o.edr) o.edr[req] = edr;
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
      if (k < act_cnt) {
        o.effect[act_off + k] = (u8)((eff4 >> (8 * k)) & 0xFFu);
        if (o.status) o.status[act_off + k] = (u8)((st4 >> (8 * k)) & 0xFFu);
        if (o.policy) o.policy[act_off + k] = pol[k];
        if (o.scope) o.scope[act_off + k] = scp[k];
      }
    }
  }
}

#ifndef CBH_HOSTSIM
#define CBH_FLAT_ATTRS(MINW) __launch_bounds__(CBH_FLAT_THREADS) __attribute__((amdgpu_waves_per_eu(MINW, 8)))
#else
#define CBH_FLAT_ATTRS(MINW)
#endif
// each wave of the group owns its region of the dynamic LDS (cbh_flat_lds); Ctx::cc = the start of that region
#define CBH_FLAT_CTX(a, ka, WITH_CALL)                                                                                            \
  const u32 ncc = cached_columns(&a);                                                                                             \
  Ctx c{a.t, a.b, a.now_ns, a.flags, threadIdx.x % CBH_BLOCK, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,  \
        (CBH_L u32*)cbh_dyn_lds + (threadIdx.x / CBH_BLOCK) * cbh_flat_lds(ncc, a.t.max_depth, a.t.K, WITH_CALL).wave_dwords, ncc, ka}
// batches of plain scalars (no int / uint / list / map attribute values): no call, under 96 VGPRs (two records in flight), 5 waves per SIMD
__global__ CBH_FLAT_ATTRS(5) void cbh_check_flat_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  CBH_FLAT_CTX(a, ka, false);
  flat_body<false>(a, c);
}
// any batch: the same walk with the call into the shared evaluator compiled in (4 waves per SIMD)
__global__ CBH_FLAT_ATTRS(4) void cbh_check_flat_kernel_any(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  CBH_FLAT_CTX(a, ka, true);
  flat_body<true>(a, c);
}

// Which kernel decides this batch: a flat one when table (CBH_MF_FLAT), batch shape (<= 4 actions and <= 4 roles per
// request; `plain_tags`: no attribute value is an int / uint / list / map - selects the variant without the evaluator call)
// and evaluation mode (not strict) allow it, else the general walk's instantiation for the table class.
// `threads` = the workgroup size to launch it with; dynamic LDS: the column cache of cbh_check_wave.h, or - `flat` - what
// cbh_flat_lds_bytes says for the variant (`flat_with_call`).
static inline cbh_check_kernel_fn cbh_pick_kernel(u32 table_flags, u32 n_derived_roles, bool has_globs, u32 max_actions, u32 max_roles, bool plain_tags,
                                                  u32 eval_flags, u32* threads, bool* flat, bool* flat_with_call) {
  *flat_with_call = true;
  *flat = (table_flags & CBH_MF_FLAT) && max_actions <= 4 && max_roles <= 4 && !(eval_flags & CBH_F_STRICT_EVALUATION);
  if (*flat) {
    *threads = CBH_FLAT_THREADS;
    const bool nocall = plain_tags && (table_flags & CBH_MF_FLAT_CLOSED);
    *flat_with_call = !nocall;
    return nocall ? cbh_check_flat_kernel : cbh_check_flat_kernel_any;
  }
  *threads = CBH_BLOCK;
  return cbh_pick_check_kernel(table_flags, n_derived_roles, has_globs, max_actions);
}
