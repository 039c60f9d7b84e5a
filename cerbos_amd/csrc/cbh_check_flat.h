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
// Every access is an ATOMIC one: with a plain read-back the compiler may take a lane that ORs nothing in (v == 0: a request
// without roles, a lane beyond the batch's end) to know the word already - it moved the read INTO `if (v)` and left such lanes
// with what they last saw there (the previous call's result, or the zero lane 0 stored).  Harmless while a lane only asked
// what IT can match; wrong as soon as a lane asks on behalf of the wave (one record to a lane: the staged walk, cbh_check_walk2.h).
__device__ __forceinline__ u64 wave_or64(u64 v, u32 wave, u32 lane) {
  __shared__ unsigned long long acc[CBH_FLAT_WAVES];
#ifndef CBH_HOSTSIM
  if (lane == 0) __hip_atomic_store(&acc[wave], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  (void)wave_ballot(true);
  if (v) (void)__hip_atomic_fetch_or(&acc[wave], (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  (void)wave_ballot(true);
  const u64 r = __hip_atomic_load(&acc[wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  if (lane == 0) acc[wave] = 0;
  (void)wave_ballot(true);
  if (v) atomicOr(&acc[wave], (unsigned long long)v);
  (void)wave_ballot(true);
  const u64 r = acc[wave];
#endif
  (void)wave_ballot(true);
  return r;
}

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
// ---- staged records.  A bucket's records used to arrive one scalar load at a time, each visit waiting for its own trip
// to memory (~1 900 cycles a visit on a 100-rule policy, profiles/r03_*).  Now the wave fetches up to 64 records with ONE
// round of vector loads - lane i takes record base + i into its own registers - decides from the class masks, all lanes
// at once, which of them can match anything in the wave (a ballot), and visits only those: a visit takes its record out
// of the owning lane's registers with v_readlane into scalar registers - no memory access - so the code behind it reads
// wave-uniform fields exactly as it did after the scalar load.
template <int NDW> struct StagedRec { u32 w[NDW]; };
template <int NDW>
__device__ __forceinline__ StagedRec<NDW> stage_rec(const CBH_G u32* base, u32 idx, bool wanted) {   // `idx` per lane; NDW % 4 == 0
  StagedRec<NDW> r;
#pragma unroll
  for (int k = 0; k < NDW; ++k) r.w[k] = 0;
  if (wanted) {
    const CBH_G u32* p = base + (size_t)idx * NDW;
#pragma unroll
    for (int k = 0; k < NDW; k += 4) { const u32x4u v = load_u32x4(p + k); r.w[k] = v.x; r.w[k + 1] = v.y; r.w[k + 2] = v.z; r.w[k + 3] = v.w; }
  }
  return r;
}
template <typename R, int NDW>
__device__ __forceinline__ R staged_take(const StagedRec<NDW>& s, u32 lane) {   // `lane` wave-uniform
  static_assert(sizeof(R) == NDW * 4, "record size");
  u32 w[NDW];
#pragma unroll
  for (int k = 0; k < NDW; ++k) w[k] = wave_readlane(s.w[k], lane);
  R r;
  __builtin_memcpy(&r, w, sizeof(R));
  return r;
}
__device__ __forceinline__ u32 ctz64(u64 v) { return (u32)__builtin_ctzll(v); }
// 1 for a word that is not zero, else 0 - as ONE instruction (min(v, 1)): written as a comparison the compiler makes it a compare
// and a select through the condition register, and where sixteen of them follow each other that is a third of the loop.
__device__ __forceinline__ u32 nz01(u32 v) {
#ifndef CBH_HOSTSIM
  u32 r;
  asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(v));
  return r;
#else
  return v ? 1u : 0u;
#endif
}

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

// A cached attribute column for this lane: tag and the two value dwords (cbh_check_wave.h fill_column_cache).
struct FlatCol { u32 t, lo, hi; };
__device__ __forceinline__ FlatCol flat_col(const Ctx& c, u32 col, u32 req) {   // `col` wave-uniform
  FlatCol v;
  v.t = cached_tag(c, col, req);
  v.lo = c.cc[col * CBH_BLOCK + c.tid];
  v.hi = c.cc[(c.n_cached + col) * CBH_BLOCK + c.tid];
  return v;
}
// The classified fused leaves (celc.py _leaf_class 1, 2, 3, 4, 6) without a single divergent branch: every lane
// computes the answer, the error flag and the "needs the full evaluator" flag with compares and selects; which
// class it is is a wave-uniform switch.  Same answers as leaf_fast (cbh_check_wave.h), which stays the reference
// for the shapes not listed here.  Returns bit 0 = satisfied, bit 1 = CEL error (counts as not satisfied),
// bit 2 = undecided here (mixed numeric types, containers): the caller hands that lane to eval_cond_rec.
// LISTS = false: the variant for tables closed over classes 1-4 and 6 (CBH_MF_FLAT_CLOSED), which never hold a membership leaf.
template <bool LISTS = true>
__device__ __forceinline__ u32 flat_leaf(const Ctx& c, const LeafRec& lr, u32 req, u32 pid) {
  const u32 a = lr.w >> 8;
  const u32 ka = (a >> 8) & 0xFu, op = a & 0xFFu;   // wave-uniform
  const bool want_eq = op == OP_EQ;
  switch (lr.pad) {
    case 1: {   // column ==/!= string or bool constant
      const FlatCol x = flat_col(c, lr.a0, req);
      const bool err = x.t >= CBH_T_ABSENT;   // ABSENT (0xF0) or ERR (0xFF)
      const bool eq = x.t == lr.ctag && x.lo == lr.clo;   // other types are plainly unequal
      return err ? 2u : (u32)(eq == want_eq);
    }
    case 2: {   // column <op> double constant
      const FlatCol x = flat_col(c, lr.a0, req);
      const bool err = x.t >= CBH_T_ABSENT;
      const bool dbl = x.t == CBH_T_DOUBLE, othernum = x.t == CBH_T_INT || x.t == CBH_T_UINT;
      const double p = as_f64((u64)x.lo | ((u64)x.hi << 32)), q = as_f64((u64)lr.clo | ((u64)lr.chi << 32));
      const bool ordering = op != OP_EQ && op != OP_NE;
      const bool cmp = (op == OP_EQ) ? p == q : (op == OP_NE) ? p != q : (op == OP_LT) ? p < q : (op == OP_LE) ? p <= q
                     : (op == OP_GT) ? p > q : p >= q;   // NaN: every ordering false, != true
      const bool slow = !err && othernum;                          // int / uint against a double constant: cross-type numerics
      const bool overload = !dbl && !othernum && ordering;         // ordering a non-number against a number: no such overload (cbh_vm.h val_compare)
      const u32 r = dbl ? (u32)cmp : (u32)(op == OP_NE);           // a non-number is plainly unequal to a number
      return (err || overload) ? 2u : slow ? 4u : r;
    }
    case 3: {   // column ==/!= column
      const FlatCol x = flat_col(c, lr.a0, req), y = flat_col(c, lr.a1, req);
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
    case 4: {   // column ==/!= P.id (either order)
      const FlatCol x = flat_col(c, ka == 3 ? lr.a0 : lr.a1, req);
      const bool err = x.t >= CBH_T_ABSENT;
      const bool eq = x.t == CBH_T_STRING && x.lo == pid;
      return err ? 2u : (u32)(eq == want_eq);
    }
    case 6: {   // column in [at most three string constants]
      const FlatCol x = flat_col(c, lr.a0, req);
      const bool err = x.t >= CBH_T_ABSENT;
      const bool found = x.t == CBH_T_STRING && (x.lo == lr.ctag || x.lo == lr.clo || x.lo == lr.chi);
      return err ? 2u : (u32)found;
    }
    case 7:     // string constant in a list / among a map's keys the request brings
    case 8: {   // column in such a column
      if (!LISTS) return 4u;
      const FlatCol h = flat_col(c, lr.a1, req);
      u32 nt = CBH_T_STRING, nlo = lr.clo;
      bool err = h.t >= CBH_T_ABSENT;
      if (lr.pad == 8u) {
        if (ka == 4u) nlo = pid;   // the principal's id
        else { const FlatCol x = flat_col(c, lr.a0, req); nt = x.t; nlo = x.lo; err = err || x.t >= CBH_T_ABSENT; }
      }
      const bool is_map = h.t == CBH_T_MAP, cont = h.t == CBH_T_LIST || is_map;
      const u32 sel = h.hi >> 30, off = h.hi & 0x3FFFFFFFu, len = cont ? h.lo : 0u, stride = is_map ? 2u : 1u;
      // a needle that is not a string, a container outside the batch's heap, a long one: the shared evaluator's
      const bool slow = !err && (nt != CBH_T_STRING || (cont && (sel != CBH_HEAP_BATCH || len > 64u)));
      const u32 n = (err || slow) ? 0u : len;
      bool found = false;
      for (u32 i = 0; wave_ballot(i < n) != 0; i += 4u) {   // four elements a round (their loads fly together); as many rounds as the longest container in the wave needs
        u32 tg[4]; u32 vl[4];
#pragma unroll
        for (u32 q = 0; q < 4; ++q) {
          const u32 at = off + stride * (i + q < n ? i + q : (n ? n - 1u : 0u));   // (a lane beyond its end re-reads its last element)
          tg[q] = n ? (u32)c.b.heap_tag[at] : 0u; vl[q] = n ? (u32)c.b.heap_val[at] : 0u;
        }
#pragma unroll
        for (u32 q = 0; q < 4; ++q) found = found || (i + q < n && tg[q] == CBH_T_STRING && vl[q] == nlo);
      }
      // `in` has no overload for anything but a list or a map on the right (cel-go: no such overload -> an evaluation error)
      return (err || (!slow && !cont)) ? 2u : slow ? 4u : (u32)found;
    }
    default: return 4u;
  }
}

// A condition tree of classified leaves (cbh_blob.h CBH_ROW_F_TREE_EMBEDDED; `desc` = the descriptor in the record's leaf
// slot): the 4-bit ops in order, leaves from the strip, no tape reads and no divergent branch.  Same bookkeeping as
// eval_leaf_tree (cbh_check_wave.h): a leaf behind the deciding one of its level is not evaluated by the reference
// (check.go:697-749), so its error / "needs the full evaluator" flags do not count.  Returns flat_leaf's bits for the tree.
template <bool LISTS = true>
__device__ __forceinline__ u32 flat_tree(const Ctx& c, const LeafRec& desc, u32 req, u32 pid) {
  const u32 opw[4] = {desc.w, desc.a0, desc.ret, desc.ctag};   // wave-uniform
  bool live = true, last = false;
  u32 saved = 0, acc = 0, depth = 0, leaf = desc.a1, err = 0, slow = 0;
  for (u32 k = 0; k < 32; ++k) {
    const u32 ws = k >> 3, word = ws == 0u ? opw[0] : ws == 1u ? opw[1] : ws == 2u ? opw[2] : opw[3];
    const u32 op = (word >> (4u * (k & 7u))) & 15u;
    if (op == 0) break;
    if (op == 1) {
      const LeafRec lr = uload_rec<LeafRec>(c.t.code, leaf++);
      const u32 lv = flat_leaf<LISTS>(c, lr, req, pid);
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

// ---- the mask walk's leaf results.  A fused leaf's outcome is one of four: false, true, CEL error, "needs the full evaluator"
// (flat_leaf's 0 / 1 / 2 / 4) - two bits.  A wave keeps the outcomes of its (at most 64) numbered leaves in LDS, four leaves to
// a byte, [leaf / 4][lane]: written leaf by leaf (wave-uniform number), read by each lane for the leaves ITS candidates'
// conditions name (lane_items) - a byte read at a per-lane row.
#define CBH_LV_ROWS 16u
__device__ __forceinline__ u32 lv_code(const CBH_L u8* lvtab, u32 tid, u32 leaf) {   // 0 false, 1 true, 2 error, 3 needs the evaluator
  return ((u32)lvtab[(leaf >> 2) * CBH_BLOCK + tid] >> ((leaf & 3u) * 2u)) & 3u;
}
__device__ __forceinline__ u32 lv_of_code(u32 code) { return code == 3u ? 4u : code; }   // back to flat_leaf's bits
// ---- evaluating a pool of classified leaves (cbh_blob.h CbhLeaf4) for every lane of the wave.  Four leaves - one 16-dword
// scalar load - at a time, and the four of a block are of ONE class (blob.py _layout_leaves pads a class's run), so a block's
// code is straight-line: no dispatch per leaf, the comparison of a numeric leaf picked by masks instead of branches.  The
// lane's last column stays in registers (a class's leaves are laid out by column).  The answers are flat_leaf's, as 2-bit
// outcomes: 0 false, 1 true, 2 CEL error, 3 needs the full evaluator.
struct __attribute__((aligned(64))) Leaf4x4 { u32 w[16]; };
__device__ __forceinline__ void leaf_col(const Ctx& c, u32 w, u32 req, u32& curcol, FlatCol& x) {   // `w`, `curcol` wave-uniform
  const u32 col = (w >> 16) & 0xFFu;
  if (col != curcol) { x = flat_col(c, col, req); curcol = col; }
}
__device__ __forceinline__ u32 leaf_block_codes(const Ctx& c, const Leaf4x4& b, u32 req, u32 pid, u32& curcol, FlatCol& x) {
  const u32 cls = b.w[0] & 15u;   // the block's class
  u32 acc = 0;
  if (cls == 1u) {          // column ==/!= string or bool constant
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      const u32 w = b.w[4 * q];
      leaf_col(c, w, req, curcol, x);
      const bool ne = ((w >> 4) & 0xFFu) != OP_EQ;
      const bool eq = x.t == b.w[4 * q + 1] && x.lo == b.w[4 * q + 2];
      acc |= (x.t >= CBH_T_ABSENT ? 2u : (u32)(eq != ne)) << (2u * q);
    }
  } else if (cls == 2u) {   // column <op> double constant: what the comparison accepts is three wave-uniform masks
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      const u32 w = b.w[4 * q], op = (w >> 4) & 0xFFu;
      leaf_col(c, w, req, curcol, x);
      const bool a_lt = op == OP_LT || op == OP_LE || op == OP_NE, a_eq = op == OP_EQ || op == OP_LE || op == OP_GE,
                 a_gt = op == OP_GT || op == OP_GE || op == OP_NE, a_un = op == OP_NE;   // (NaN: every ordering false, != true)
      const double p = as_f64((u64)x.lo | ((u64)x.hi << 32)), k = as_f64((u64)b.w[4 * q + 1] | ((u64)b.w[4 * q + 2] << 32));
      const bool lt = p < k, eq = p == k, gt = p > k;
      const bool cmp = (a_lt && lt) || (a_eq && eq) || (a_gt && gt) || (a_un && !(lt || eq || gt));
      const bool dbl = x.t == CBH_T_DOUBLE, othernum = x.t == CBH_T_INT || x.t == CBH_T_UINT;
      const bool ordering = op != OP_EQ && op != OP_NE;
      const u32 r = dbl ? (u32)cmp : (u32)(op == OP_NE);   // a non-number is plainly unequal to a number
      acc |= ((x.t >= CBH_T_ABSENT || (!dbl && !othernum && ordering)) ? 2u : othernum ? 3u : r) << (2u * q);
    }
  } else if (cls == 6u) {   // column in [at most three string constants]
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      const u32 w = b.w[4 * q];
      leaf_col(c, w, req, curcol, x);
      const bool found = x.t == CBH_T_STRING && (x.lo == b.w[4 * q + 1] || x.lo == b.w[4 * q + 2] || x.lo == b.w[4 * q + 3]);
      acc |= (x.t >= CBH_T_ABSENT ? 2u : (u32)found) << (2u * q);
    }
  } else if (cls == 4u) {   // column ==/!= P.id
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      const u32 w = b.w[4 * q];
      leaf_col(c, w, req, curcol, x);
      const bool ne = ((w >> 4) & 0xFFu) != OP_EQ;
      acc |= (x.t >= CBH_T_ABSENT ? 2u : (u32)((x.t == CBH_T_STRING && x.lo == pid) != ne)) << (2u * q);
    }
  } else if (cls == 3u) {   // column ==/!= column
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
      const u32 w = b.w[4 * q];
      leaf_col(c, w, req, curcol, x);
      const FlatCol y = flat_col(c, (w >> 24) & 0xFFu, req);
      const bool ne = ((w >> 4) & 0xFFu) != OP_EQ;
      const bool same = x.t == y.t;
      const bool scalar = x.t < CBH_T_LIST || x.t == CBH_T_TIMESTAMP || x.t == CBH_T_DURATION;
      const bool xnum = x.t == CBH_T_INT || x.t == CBH_T_UINT || x.t == CBH_T_DOUBLE, ynum = y.t == CBH_T_INT || y.t == CBH_T_UINT || y.t == CBH_T_DOUBLE;
      const bool bits_eq = x.lo == y.lo && x.hi == y.hi;
      const bool dbl_eq = as_f64((u64)x.lo | ((u64)x.hi << 32)) == as_f64((u64)y.lo | ((u64)y.hi << 32));
      const bool eq = same && (x.t == CBH_T_DOUBLE ? dbl_eq : bits_eq);
      const bool slow = (same && !scalar) || (!same && xnum && ynum);   // containers / cross-type numeric equality
      acc |= ((x.t >= CBH_T_ABSENT || y.t >= CBH_T_ABSENT) ? 2u : slow ? 3u : (u32)(eq != ne)) << (2u * q);
    }
  } else acc = 0xFFu;       // none of the classified shapes: the shared evaluator's
  return acc;
}
// The blocks of `todo` (bit g = leaves 4g .. 4g + 3 of the pool at `recs`; wave-uniform) into `lvtab`, front to back; the next
// block's record is in flight while this one is evaluated.
__device__ __forceinline__ void eval_leaf_blocks(const Ctx& c, const CBH_G u32* recs, u32 todo, u32 req, u32 pid, CBH_L u8* lvtab) {
  if (todo == 0) return;
  u32 curcol = CBH_NONE;
  FlatCol cx; cx.t = 0; cx.lo = 0; cx.hi = 0;
  u32 g = (u32)__builtin_ctz(todo);
  Leaf4x4 nxt = uload_rec<Leaf4x4>(recs, g);
  while (todo != 0) {
    todo &= todo - 1u;
    const Leaf4x4 b = nxt;
    const u32 gn = todo ? (u32)__builtin_ctz(todo) : g;
    nxt = uload_rec<Leaf4x4>(recs, gn);
    lvtab[g * CBH_BLOCK + c.tid] = (u8)leaf_block_codes(c, b, req, pid, curcol, cx);
    g = gn;
  }
}
// A deeper tree of classified leaves from the leaves' outcomes in `lvtab` (flat_tree's bookkeeping): `ops` / `idx` wave-uniform.
__device__ __forceinline__ u32 tree_from_codes(const u32 (&opw)[4], u64 idx, const CBH_L u8* lvtab, u32 tid) {
  bool live = true, last = false;
  u32 saved = 0, acc = 0, depth = 0, err = 0, slow = 0;
  for (u32 k = 0; k < 32; ++k) {
    const u32 ws = k >> 3, word = ws == 0u ? opw[0] : ws == 1u ? opw[1] : ws == 2u ? opw[2] : opw[3];   // (scalar selects: an index the loop computes would put the four words in scratch)
    const u32 op = (word >> (4u * (k & 7u))) & 15u;
    if (op == 0) break;
    if (op == 1) {
      const u32 code = lv_code(lvtab, tid, (u32)idx & 0xFFu); idx >>= 8;
      last = live && code == 1u;
      err |= (live && code == 2u) ? 2u : 0u;
      slow |= (live && code == 3u) ? 4u : 0u;
    } else if (op < 5) {
      const u32 bit = 1u << depth;
      saved = live ? (saved | bit) : (saved & ~bit);
      acc = (op == 2) ? (acc | bit) : (acc & ~bit);
      ++depth;
    } else if (op < 8) {
      const u32 bit = 1u << (depth - 1u);
      const bool decides = live && ((op == 5) ? !last : last);
      acc = decides ? ((op == 5) ? (acc & ~bit) : (acc | bit)) : acc;
      live = live && !decides;
    } else {
      --depth;
      const u32 bit = 1u << depth;
      live = (saved & bit) != 0;
      last = ((acc & bit) != 0) != (op == 10);
    }
  }
  return slow ? 4u : ((u32)last | err);
}
// ONE level of leaves from their 2-bit outcomes packed in `w` (leaf j of the level = bits 2j, 2j + 1; `n` leaves):
// all - the first leaf that is not satisfied decides (an error counts as not satisfied and is reported, check.go:697-749);
// any - the first satisfied leaf decides; the leaves behind the deciding one are never evaluated by the reference, so their
// errors / "needs the evaluator" outcomes do not count.  Everything may differ from lane to lane.  Returns flat_leaf's bits.
__device__ __forceinline__ u32 level_from_codes(u32 w, u32 n, bool any, bool neg) {
  const u32 F = 0x5555u & ((1u << (2u * n)) - 1u);   // bit 2j for the leaves that exist
  const u32 lo = w & F, hi = (w >> 1) & F;
  const u32 s = lo & ~hi, e = hi & ~lo, sl = lo & hi;   // satisfied / error / needs the evaluator, at bit 2j
  const u32 dec = any ? s : (~s & F);                   // the leaves that would decide; the first of them does
  const u32 first = dec & (0u - dec);
  const u32 evald = dec ? (first | (first - 1u)) : F;   // the leaves the reference evaluates
  const bool res = any ? dec != 0 : dec == 0;
  return (evald & sl) ? 4u : ((u32)(res != neg) | ((evald & e) ? 2u : 0u));
}
struct __attribute__((aligned(64))) SegHdr { u32 allow_lo, allow_hi, deny_lo, deny_hi, sc_lo, sc_hi, sd_lo, sd_hi, n_items, n_leaves, size16, n_records, row_begin, n_complex, off_rc, off_leaves; };
struct __attribute__((aligned(64))) SegItem { u32 cc_lo, cc_hi, cd_lo, cd_hi, how, ref, id, n_leaves, idx_lo, idx_hi, pad0, pad1, ops[4]; };
// per wave, in LDS: the segment's class masks u64[64], record -> item u8[2][64], item descriptors u64[64], the leaves' outcomes u8[16][64]
#define CBH_SEG_LDS_BYTES (512u + 128u + 512u + CBH_LV_ROWS * CBH_BLOCK)
__device__ __forceinline__ u64 load_u64g(const CBH_G u32* p) {   // 8-byte aligned
#ifndef CBH_HOSTSIM
  return *(const CBH_G u64*)p;
#else
  u64 v; __builtin_memcpy(&v, p, 8); return v;
#endif
}

// MODE 0: records one scalar load at a time; 1: staged 64 at a time in the lanes' registers; 2: no record is visited - the
// bucket's SEGMENTS decide by masks (cbh_blob.h CBH_SEC_SEGS).
// EP: the instantiations of cbh_check_batch_trail (AuditTrail.EffectivePolicies, check.go:302-304) - per chain position a lane keeps
// which of its walks met a binding there and the bucket's policy; the fold marks the policies the walks the reference REALLY makes
// (not those of a role behind the one that allowed) have touched.  A bucket of a flat table is one resource policy's.
// MEMO: the walk keeps the wave's last four condition outcomes (leafish_memo below) - the instantiation for tables WITH derived roles
// (cbh_check_flat_kernel_dr): the memo's registers cost the plain kernel a wave of occupancy and C2 2 % for nothing.
template <bool WITH_CALL, int MODE, bool EP = false, bool MEMO = false>
__device__ __forceinline__ void flat_body(const KernelArgs& ka_regs, Ctx& c) {
  constexpr bool STAGED = MODE == 1;
  constexpr u32 BTYPE = MODE == 2 ? (u32)CBH_B_RESSEG : (u32)CBH_B_RESOURCE;
  const TableDev& t = ka_regs.t;
  const BatchDev& b = ka_regs.b;
  const OutDev& o = ka_regs.o;
  const u32 flags = ka_regs.flags;
  const u32 wave = threadIdx.x / CBH_BLOCK;   // (NOT through readfirstlane: measured 7-19 % slower on every workload, profiles/r06_ab_prologue.txt)   // which of the group's waves (c.tid is the lane within it)
#ifdef CBH_PROFILE_CYCLES   // profiling build only (tools/gpu_cycles_flat.py)
  const u64 cyc0 = __builtin_readcyclecounter();
  const u64 rt0 = __builtin_amdgcn_s_memrealtime();
  u32 dbg_rows = 0, dbg_rounds = 0, dbg_evals = 0, dbg_visits = 0;
  u64 cyc_eval = 0, cyc_stage = 0;
#define FLAT_DBG(x) x
#else
#define FLAT_DBG(x)
#endif
  const u32 rix = b.req_lo + blockIdx.x * CBH_FLAT_THREADS + threadIdx.x;
  const bool valid = rix < b.req_hi;
  const u32 req = valid ? rix : b.req_lo;
  const u32 NR = b.n_requests;
#define RQ(f) b.req_u32[(size_t)(f) * NR + req]
  const u32 pid = RQ(CBH_RQ_PRINCIPAL_ID), kind = RQ(CBH_RQ_KIND), r_scope = RQ(CBH_RQ_R_SCOPE), r_ver = RQ(CBH_RQ_R_VERSION);
  const u32 role_off = RQ(CBH_RQ_ROLE_OFF), act_off = RQ(CBH_RQ_ACT_OFF);
  const u32 role_cnt = valid ? RQ(CBH_RQ_ROLE_CNT) : 0, act_cnt = valid ? RQ(CBH_RQ_ACT_CNT) : 0;   // both <= 4 (host-checked)
#undef RQ
  // ---- the request's loads, in TWO round trips to memory.  First trip, all issued back to back and depending on nothing: the
  // request words above, the four action ids (speculated, below), this thread's bytes of the two class tables, the tag bytes of
  // the cached columns (cc_load_tags).  Second trip, issued when the request words are in: the role ids and every column's
  // asynchronous copies into LDS (cc_fill).  The copies go out LAST - any LDS access the compiler cannot tell apart from
  // their destinations waits for all of them - behind this prologue's own LDS stores.  (They used to go out first: the class
  // tables' LDS stores then waited for the columns, the action ids were loaded behind that wait and the role ids behind another
  // - three trips, and one more per column where the tags are packed.)
  const u32 w0r = b.req_lo + blockIdx.x * CBH_FLAT_THREADS + wave * CBH_BLOCK;
  const u32 w0 = w0r < b.req_hi ? w0r : b.req_lo;   // the wave's first request (uniform): the columns' planes are addressed from it
  const u32 wd = valid ? c.tid : 0u;                // ... and this lane's distance from it
  // Actions: a batch of four-action requests laid out back to back has ACT_OFF = 4 * request - read the four ids from
  // there with ONE 16-byte load that does not wait for ACT_OFF to arrive, and fall back to the dependent loads for the
  // lanes where the guess was wrong.  (Unconditional: without four tuples the load reads request words and is not used.)
  const bool spec = b.n_tuples >= 4u;   // wave-uniform
  const u32 spec_ix = (4u * req + 4u <= b.n_tuples) ? 4u * req : 0u;
  const u32x4u sp = load_u32x4(spec ? b.tuple_action + spec_ix : b.req_u32);
  const u32 kmax = t.K ? t.K - 1u : 0u;
  const u32 cls_i = threadIdx.x < t.K ? threadIdx.x : kmax;
  const u32 cls_a0 = (t.K ? t.action_class : (const CBH_G u8*)b.req_u32)[cls_i], cls_r0 = (t.K ? t.role_class : (const CBH_G u8*)b.req_u32)[cls_i];
  const CcTags cct = cc_load_tags(c, b, NR, w0, wd);
  const u32 all = (1u << act_cnt) - 1u;
  // [depth][lane]: scope index at that depth of the lane's chain - in the dynamic LDS behind the column caches,
  // sized by the table's longest chain (a one-scope table pays 256 B per wave, not 4 KB: LDS sets the occupancy here)
  const u32 max_depth = t.max_depth < CBH_FLAT_MAX_DEPTH ? t.max_depth : CBH_FLAT_MAX_DEPTH;
  // A table of at most 256 scopes keeps them as bytes: a quarter of the footprint.
  const bool chain8 = t.n_scopes <= 256u;
  const u32 chain_dwords = chain8 ? max_depth * (CBH_BLOCK / 4u) : max_depth * CBH_BLOCK;
  CBH_L u32* chain_si = (CBH_L u32*)cbh_dyn_lds + CBH_FLAT_WAVES * CBH_CC_DWORDS(c.n_cached, (c.flags & CBH_FI_PACKED_TAGS) != 0) + wave * chain_dwords;
  CBH_L u8* chain_si8 = (CBH_L u8*)chain_si;
  // actions and roles -> classes (CBH_SEC_ACTION_CLASS / CBH_SEC_ROLE_CLASS; 63 = a string no rule names).
  // A flat table has fewer than 32 classes per dimension and its masks mirror "any other string" (bit 63)
  // in bit 31 of the low dword: the match is a 1-bit field extract from ONE dword at a per-lane position.
  // Every load below is unconditional (an index that does not exist reads element 0 instead and is masked
  // afterwards): the eight id loads go out together, then the eight class loads - two round trips, not sixteen.
  // The class tables (one byte per table string) are copied into LDS by the workgroup while the request loads are in
  // flight - a table of up to CBH_FLAT_LDS_STRINGS strings - so that the lookups below are LDS reads, not a third
  // dependent trip to memory.
  const bool cls_in_lds = t.K <= CBH_FLAT_LDS_STRINGS;
  CBH_L u8* cls_lds = (CBH_L u8*)((CBH_L u32*)cbh_dyn_lds + CBH_FLAT_WAVES * (CBH_CC_DWORDS(c.n_cached, (c.flags & CBH_FI_PACKED_TAGS) != 0) + chain_dwords));   // [action classes K][role classes K]
  // second trip: the role ids (and, where the speculation missed, the action ids)
  u32 ac[4], rc[4], aid[4], rid[4];
#pragma unroll
  for (u32 k = 0; k < 4; ++k) rid[k] = b.roles[k < role_cnt ? role_off + k : 0u];
  const bool spec_hit = spec && act_cnt == 4u && act_off == spec_ix;
  aid[0] = sp.x; aid[1] = sp.y; aid[2] = sp.z; aid[3] = sp.w;
  if (!spec_hit) {
#pragma unroll
    for (u32 k = 0; k < 4; ++k) aid[k] = b.tuple_action[k < act_cnt ? act_off + k : 0u];
  }
  if (cls_in_lds) {
    if (threadIdx.x < t.K) { cls_lds[threadIdx.x] = (u8)cls_a0; cls_lds[t.K + threadIdx.x] = (u8)cls_r0; }
    for (u32 i = threadIdx.x + CBH_FLAT_THREADS; i < t.K; i += CBH_FLAT_THREADS) { cls_lds[i] = t.action_class[i]; cls_lds[t.K + i] = t.role_class[i]; }
  }
  // MODE 2: the class masks of the segment being decided, per wave (behind the class tables: cbh_flat_class_bytes is a multiple of 16)
  CBH_L u64* segm = (CBH_L u64*)(cls_lds + (cls_in_lds ? ((2u * t.K + 15u) & ~15u) : 0u)) + wave * (CBH_SEG_LDS_BYTES / 8u);
  CBH_L u8* seg_rec = (CBH_L u8*)(segm + 64);      // [2][64]: record -> item of its condition / of its derived-role condition
  CBH_L u64* seg_desc = segm + 64 + 16;            // [64] CbhSegDesc
  CBH_L u8* lvtab = (CBH_L u8*)(segm + 64 + 16 + 64);   // [CBH_LV_ROWS][64]
  // EP: [depth][touched walks | policy][lane], behind everything else (cbh_flat_trail_bytes)
  CBH_L u32* ep_lds = (CBH_L u32*)(cls_lds + (cls_in_lds ? ((2u * t.K + 15u) & ~15u) : 0u) + (MODE == 2 ? CBH_FLAT_WAVES * CBH_SEG_LDS_BYTES : 0u)) + wave * max_depth * 2u * CBH_BLOCK;
  const bool want_ep = EP && (flags & CBH_F_WANT_EFFECTIVE_POLICIES) != 0 && o.eff_pol != nullptr;
  if (EP) { for (u32 d = 0; d < 2u * max_depth; ++d) ep_lds[d * CBH_BLOCK + c.tid] = 0; }
  cc_fill(c, b, NR, w0, wd, cct);   // (behind every LDS store of this prologue)
  // ... and the chain's first scope (ruletable.go:848-882; per lane, reads the scope tables): its load goes out with the second trip
  const bool lenient = (flags & CBH_F_LENIENT_SCOPE_SEARCH) != 0;
  const u32 first = chain_first(t, r_scope, FLAG_RES, lenient);
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
    FLAT_DBG(const u64 e0 = __builtin_readcyclecounter(); ++dbg_evals;)
    if (how == 1u) lv = flat_leaf<WITH_CALL>(c, lr, req, pid);
    else if (how == 2u) lv = flat_tree<WITH_CALL>(c, lr, req, pid);
    FLAT_DBG(cyc_eval += (__builtin_readcyclecounter() - e0) * (u64)(wave_ballot(lv != 77u) != 0);)
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

  // The same with a MEMO of the wave's last outcomes (MEMO instantiations, without the evaluator call: there a classified condition's
  // outcome is computed branch-free for EVERY lane, whoever asked).  A table's derived-role conditions are a handful of programs
  // that every rule naming the role and every scope's definitions share (the lowering interns programs): C3 evaluates its four
  // about a dozen times per wave - the walk per record, the second climb per scope.  Four entries: the reference (uniform) and
  // the lanes' 4-bit outcomes, packed in one register.  `leaf` is called only on a miss (for a record's derived-role condition
  // it is a scalar load of its own).
  u32 memo_r0 = CBH_NONE, memo_r1 = CBH_NONE, memo_r2 = CBH_NONE, memo_r3 = CBH_NONE, memo_n = 0;   // wave-uniform
  u32 memo_lv = 0;
  auto leafish_memo = [&](u32 ref, u32 how, auto&& leaf, bool active) -> u32 {
    if constexpr (WITH_CALL || !MEMO) return leafish(ref, how, leaf(), active);
    else {
      const u32 j = ref == memo_r0 ? 0u : ref == memo_r1 ? 1u : ref == memo_r2 ? 2u : ref == memo_r3 ? 3u : 4u;
      if (j < 4u) return active ? ((memo_lv >> (4u * j)) & 15u) : 0u;
      const u32 lv = leafish(ref, how, leaf(), true);   // every lane's outcome (a lane beyond the batch's end shadows a request)
      if (memo_n < 4u) {
        memo_r0 = memo_n == 0u ? ref : memo_r0; memo_r1 = memo_n == 1u ? ref : memo_r1; memo_r2 = memo_n == 2u ? ref : memo_r2; memo_r3 = memo_n == 3u ? ref : memo_r3;
        memo_lv |= lv << (4u * memo_n);
        ++memo_n;
      }
      return active ? lv : 0u;
    }
  };

  // ---- the walk.  check.go:208-442 walks, per request, its scope chain from the request's scope up to the root and,
  // per scope, the bindings of (version, kind, scope).  Here every lane keeps the scope it stands at (`cur`, its own
  // depth in `mydepth`); each round takes the DEEPEST scope any lane stands at (scopes are numbered parents first, so
  // that is the largest index), and the lanes standing there with the leader's (version, kind) walk that one bucket
  // together, then step to their parent scope.  Lanes whose chains start at different scopes of one branch therefore
  // fall in with each other as soon as the deeper ones have climbed to the shallower ones' start: a wave holding the
  // five request scopes of one kind walks its three buckets once, not once per request scope.  Each lane still meets
  // its own scopes in chain order and a bucket's records in binding order.
  u32 cur = first, mydepth = 0;
  bool exists = false;
  // MODE 2.  Pooled table: the leaves are numbered table-wide and evaluated ONCE per wave, when the first lane meets a
  // candidate with a condition; else every segment brings its own.
  const bool pooled = MODE == 2 && (t.seg_info & CBH_MSEG_POOLED) != 0;
  u32 blocks_done = 0;   // wave-uniform: the blocks of four leaves whose outcomes lvtab holds
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
    const bool have_bucket = udir_find(t, BTYPE, g_ver, g_k, g_si, bucket);   // present for every resource policy (index.go:966-997)
    exists = exists || (ing && have_bucket);
    if (go) {
      if (ing && mydepth < max_depth) { if (chain8) chain_si8[mydepth * CBH_BLOCK + c.tid] = (u8)g_si; else chain_si[mydepth * CBH_BLOCK + c.tid] = g_si; }
      const u32 S_before = S;
      u32 touch = 0, g_pol = 0;   // EP: the walks that met a binding of this bucket, the bucket's policy
      if constexpr (MODE == 2) {
        // ---- the mask walk.  Index.Query answers a request with an AND of per-dimension bitmaps (index/index.go:270-305);
        // so does this: per segment of <= 64 records, a lane's candidates = (OR of the masks of its action classes) &
        // (OR of its role classes'); the fused leaves of the table's conditions are evaluated once per wave, for every lane
        // (eval_leaf_pool); each lane then decides the conditions of ITS candidates from the leaves' outcomes (lane_items);
        // and a walk's outcome in the segment is mask algebra: hits = candidates of (role, action) & satisfied records, the
        // first DENY among them ends the walk (check.go:392-403), an ALLOW before it counts, an evaluation error before it counts.
        // A segment's header (one scalar load) and its tables (three vector loads, one element per lane: 64 class masks, 2 x 64
        // record -> item bytes as 32 dwords, the items' descriptors - all at fixed places of the block, none waits for the
        // header) are fetched one segment AHEAD: the loads of segment s + 1 fly while segment s is decided.
        u32 blk16 = bucket.x;
        const bool any_seg = have_bucket && bucket.y != 0;
        SegHdr hd_nx = uload_rec<SegHdr>(t.segs, any_seg ? blk16 : 0u);
        const CBH_G u32* bn = t.segs + (size_t)(any_seg ? blk16 : 0u) * 16u;
        u64 m_nx = load_u64g(bn + 16u + 2u * c.tid), d_nx = load_u64g(bn + CBH_SEG_FIXED_DWORDS + 2u * c.tid);
        u32 r_nx = bn[144u + (c.tid & 31u)];
        for (u32 sgi = 0; any_seg && sgi < bucket.y && wave_ballot(ing && S != 0) != 0; ++sgi) {
          const CBH_G u32* blk = t.segs + (size_t)blk16 * 16u;
          const SegHdr hd = hd_nx;
          blk16 += hd.size16;
          if (!pooled) blocks_done = 0;   // the segment's own leaves
          segm[c.tid] = m_nx;
          if (c.tid < 32u) ((CBH_L u32*)seg_rec)[c.tid] = r_nx;
          seg_desc[c.tid] = d_nx;
          {   // the next segment's (unconditional - behind the last segment the bucket's first block is read again and not used: a
              // load inside a conditional block is waited for at the block's end, which made this fetch-ahead a wait per segment)
            const u32 nb16 = (sgi + 1u < bucket.y) ? blk16 : bucket.x;
            hd_nx = uload_rec<SegHdr>(t.segs, nb16);
            bn = t.segs + (size_t)nb16 * 16u;
            m_nx = load_u64g(bn + 16u + 2u * c.tid); d_nx = load_u64g(bn + CBH_SEG_FIXED_DWORDS + 2u * c.tid);
            r_nx = bn[144u + (c.tid & 31u)];
          }
          (void)wave_ballot(true);
          u64 A = 0, R = 0;
#pragma unroll
          for (u32 k = 0; k < 4; ++k) { if (k < act_cnt) A |= segm[ac[k]]; if (k < role_cnt) R |= segm[32u + rc[k]]; }
          const u64 cand_any = (ing && S != 0) ? (A & R) : 0ull;
          if (EP) {
            if (sgi == 0) g_pol = uload(&t.rows[(size_t)hd.row_begin * 16u + 3u]);   // TblRow.policy of the bucket's first record
            if (cand_any != 0) {
#pragma unroll
              for (u32 k = 0; k < 4; ++k)
#pragma unroll
                for (u32 r = 0; r < 4; ++r)
                  if (((S >> (4u * r + k)) & 1u) && (segm[ac[k]] & segm[32u + rc[r]]) != 0) touch |= 1u << (4u * r + k);
            }
          }
          FLAT_DBG(dbg_rows += hd.n_records;)
          if (wave_ballot(cand_any != 0) != 0) {
            u64 csat = ~0ull, dsat = ~0ull, cerr = 0, derr = 0, cuns = 0, duns = 0;
            const u64 simple_c = (u64)hd.sc_lo | ((u64)hd.sc_hi << 32), simple_d = (u64)hd.sd_lo | ((u64)hd.sd_hi << 32);
            FLAT_DBG(const u64 i0 = __builtin_readcyclecounter();)
            const bool any_cond = hd.n_complex != 0 || wave_ballot((cand_any & (simple_c | simple_d)) != 0) != 0;   // some candidate has a condition
            if (any_cond) {
              // the leaves' outcomes, for every lane of the wave: every block lvtab does not hold yet, in one go.  (Evaluating
              // only the blocks the wave's role classes can reach - about half of them on T - was measured SLOWER, 25.5 against
              // 29.7 G decisions/s sustained: the short batches of blocks expose the scalar loads the long run hides.)
              const u32 nblk = ((pooled ? (t.seg_info & 0xFFu) : hd.n_leaves) + 3u) >> 2;
              const u32 todo = ~blocks_done & ((1u << nblk) - 1u);
              if (todo) {
                FLAT_DBG(const u64 l0 = __builtin_readcyclecounter();)
                eval_leaf_blocks(c, pooled ? t.leafpool : blk + hd.off_leaves, todo, req, pid, lvtab);
                blocks_done |= todo;
                (void)wave_ballot(true);
                FLAT_DBG(cyc_stage += (__builtin_readcyclecounter() - l0) * (u64)(wave_ballot(req != 0xdeadbeefu) != 0); dbg_visits += (u32)__builtin_popcount(todo);)
              }
            }
            // the conditions the wave evaluates as one (deeper trees, more than four leaves, programs): rare
            for (u32 ci = 0; ci < hd.n_complex; ++ci) {
              const SegItem it = uload_rec<SegItem>(blk + (hd.off_rc >> 16), ci);
              const u64 cc = (u64)it.cc_lo | ((u64)it.cc_hi << 32), cd = (u64)it.cd_lo | ((u64)it.cd_hi << 32);
              const bool need = ((cc | cd) & cand_any) != 0;
              if (wave_ballot(need) == 0) continue;
              FLAT_DBG(++dbg_evals;)
              const u64 idx = (u64)it.idx_lo | ((u64)it.idx_hi << 32);
              u32 lv = 4u;
              if ((it.how & 3u) == 1u) {
                u32 w = 0;
                for (u32 j = 0; j < it.n_leaves; ++j) w |= lv_code(lvtab, c.tid, (u32)(idx >> (8u * j)) & 0xFFu) << (2u * j);
                lv = level_from_codes(w, it.n_leaves, (it.how & 256u) != 0, (it.how & 512u) != 0);
              } else if ((it.how & 3u) == 2u) lv = tree_from_codes(it.ops, idx, lvtab, c.tid);
              const bool slow = need && lv == 4u;
              if (WITH_CALL) {
                if (wave_ballot(slow) != 0) {
                  const u32 r = eval_ref<false>(c.ka_mem, lds_of(c), req, 0, false, it.ref, slow);
                  if (slow) lv = ((r & 0xFFu) == 1u ? 1u : 0u) | (((r >> 8) & CBH_ST_CEL_ERROR) ? 2u : 0u) | (((r >> 8) & CBH_ST_UNSUPPORTED) ? 8u : 0u);
                }
              } else if (slow) lv = 8u;   // unreachable by the host's check; loud (UNSUPPORTED), never a guessed effect
              csat &= (lv & 1u) ? ~0ull : ~cc; dsat &= (lv & 1u) ? ~0ull : ~cd;
              cerr |= (lv & 2u) ? cc : 0ull; derr |= (lv & 2u) ? cd : 0ull; cuns |= (lv & 8u) ? cc : 0ull; duns |= (lv & 8u) ? cd : 0ull;
            }
            // every lane by itself: its candidates' conditions that are ONE level of at most four leaves (nearly all are), from
            // the leaves' outcomes - straight-line code, as many rounds as the lane with the most such candidates has
            u64 cslow = 0, dslow = 0;
            u32 lane_slow = 0;   // the variant without the call: some candidate's condition needs the evaluator (cannot be: the host checks)
            auto lane_items = [&](u64 m, const CBH_L u8* rec, u64& sat, u64& errm, u64& slowm) {
              while (wave_ballot(m != 0) != 0) {
                const bool act = m != 0;
                const u32 i = act ? ctz64(m) : 0u;
                m &= m - 1ull;
                const u32 itx = rec[i];
                const u64 d = seg_desc[itx < CBH_SEG_RECORDS ? itx : 0u];
                const u32 lw = (u32)d, fl = (u32)(d >> 32);
                u32 w = 0;
#pragma unroll
                for (u32 j = 0; j < 4; ++j) {   // leaf number l: its outcome is bits 2 (l & 3) .. of lvtab[l >> 2][lane]
                  const u32 row = (lw >> (8u * j + 2u)) & 0x3Fu, sh = ((lw >> (8u * j)) & 3u) * 2u;
                  w |= (((u32)lvtab[row * CBH_BLOCK + c.tid] >> sh) & 3u) << (2u * j);
                }
                const u32 lv = level_from_codes(w, fl & 7u, (fl & 8u) != 0, (fl & 16u) != 0);
                const u64 bit = act ? (1ull << i) : 0ull;
                sat &= (lv & 1u) ? ~0ull : ~bit;
                errm |= (lv & 2u) ? bit : 0ull;
                if (WITH_CALL) slowm |= (lv & 4u) ? bit : 0ull; else lane_slow |= act ? (lv & 4u) : 0u;
                FLAT_DBG(++dbg_evals;)
              }
            };
            // (Deciding each record of the wave's UNION of candidates once, on wave-uniform descriptors, when that union is small -
            // a wave's requests share route and mostly roles - was built and measured 3 % slower on T, 29.8 against 30.6 G
            // decisions/s: fewer vector instructions, but the scalar chain per record is longer than the per-lane round.)
            if (simple_c) lane_items(cand_any & simple_c, seg_rec, csat, cerr, cslow);
            if (simple_d) lane_items(cand_any & simple_d, seg_rec + CBH_SEG_RECORDS, dsat, derr, dslow);
            // a leaf that needs the full evaluator (int / uint / container values): the variant with the call runs the item's
            // program for the lanes whose candidate it is, one record at a time; the other variant cannot meet one (the host checks)
            if (wave_ballot((cslow | dslow) != 0) != 0) {
              if (WITH_CALL) {
                for (;;) {
                  const u64 who = wave_ballot((cslow | dslow) != 0);
                  if (who == 0) break;
                  const u32 ld = first_lane(who);
                  const u64 lc = wave_readlane64(cslow, ld), ldd = wave_readlane64(dslow, ld);
                  const bool is_d = lc == 0;
                  const u32 i = ctz64(is_d ? ldd : lc);
                  const u64 bit = 1ull << i;
                  const u32 itx = uniform((u32)seg_rec[(is_d ? CBH_SEG_RECORDS : 0u) + i]);
                  const u32 ref = uload(blk + (hd.off_rc & 0xFFFFu) + itx);
                  const bool mine = ((is_d ? dslow : cslow) & bit) != 0;
                  const u32 r = eval_ref<false>(c.ka_mem, lds_of(c), req, 0, false, ref, mine);
                  if (mine) {
                    const bool ok = (r & 0xFFu) == 1u, e = ((r >> 8) & CBH_ST_CEL_ERROR) != 0, u = ((r >> 8) & CBH_ST_UNSUPPORTED) != 0;
                    if (is_d) { dslow &= ~bit; dsat = ok ? (dsat | bit) : dsat; derr |= e ? bit : 0ull; duns |= u ? bit : 0ull; }
                    else { cslow &= ~bit; csat = ok ? (csat | bit) : csat; cerr |= e ? bit : 0ull; cuns |= u ? bit : 0ull; }
                  }
                }
              }
            }
            if (!WITH_CALL && lane_slow) { cuns |= cand_any; duns |= cand_any; }   // loud (UNSUPPORTED on every candidate), never a guessed effect
            FLAT_DBG(cyc_eval += (__builtin_readcyclecounter() - i0) * (u64)(wave_ballot(csat != 77u) != 0);)
            // the derived-role condition comes first; the rule's own condition counts only where that held (check.go:328-380)
            const u64 satrec = dsat & csat, errrec = derr | (dsat & cerr), unsrec = duns | (dsat & cuns);
            const u64 allow_m = (u64)hd.allow_lo | ((u64)hd.allow_hi << 32), deny_m = (u64)hd.deny_lo | ((u64)hd.deny_hi << 32);
            const bool anybad = wave_ballot(((errrec | unsrec) & cand_any) != 0) != 0;
            // ---- the walks' outcomes in this segment.  Within a segment the sixteen walks (role r, action k) of a lane are
            // independent of each other: a walk reads only its own bit of S.  What a walk needs is two facts - did it meet a
            // satisfied ALLOW, did it meet a satisfied DENY (which ends it, check.go:392-403; an ALLOW of a walk a DENY ends is
            // never read again) - i.e. whether two 64-bit ANDs are zero.  The action side of both ANDs is formed once per action
            // (records of the lane's k-th action whose conditions hold, split by effect), a walk then costs two AND / AND-OR pairs
            // and two min / shift-or pairs that push its two facts into two 16-bit words - bit 4r + k, roles and actions taken
            // from the top so that the first fact pushed ends at bit 15 - and the bookkeeping is done once for all sixteen.
            const u32 live16 = ing ? S : 0u;   // this lane's walks still going at the segment's start
            if (anybad) {
              // An evaluation error (or a condition outside the device subset) among the wave's candidates - a few lanes of a few
              // waves: the walk's error bits are kept exactly, per walk, and only for the walks in which some lane has such a
              // candidate (two cheap tests first: no lane's r-th role reaches one of its bad candidates; nor through its k-th action).
              // An error counts for a walk when the reference evaluated that record: up to and including the walk's first
              // satisfied DENY (a tree can be satisfied AND have absorbed an error).
              const u64 badc = cand_any & (errrec | unsrec);   // this lane's candidates that raised one: none, for nearly every lane
#pragma unroll
              for (u32 r = 0; r < 4; ++r) {
                const u64 rmk = segm[32u + rc[r]];
                const bool rlive = ((live16 >> (4u * r)) & 0xFu) != 0;
                if (wave_ballot(rlive && (rmk & badc) != 0) == 0) continue;
#pragma unroll
                for (u32 k = 0; k < 4; ++k) {
                  const u32 bit = 1u << (4u * r + k);
                  const u64 cand = (live16 & bit) ? (rmk & segm[ac[k]]) : 0ull;
                  if (wave_ballot((cand & badc) != 0) == 0) continue;
                  const u64 dh = cand & satrec & deny_m;
                  const u64 below = dh ? (((dh & (0ull - dh)) << 1) - 1ull) : ~0ull;
                  err |= (cand & errrec & below) ? bit : 0u;
                  unsup |= (cand & unsrec & below) ? bit : 0u;
                }
              }
            }
            u32 a_lo[4], a_hi[4], d_lo[4], d_hi[4];
#pragma unroll
            for (u32 k = 0; k < 4; ++k) {
              const u64 am = segm[ac[k]] & satrec;
              a_lo[k] = (u32)(am & allow_m); a_hi[k] = (u32)((am & allow_m) >> 32);
              d_lo[k] = (u32)(am & deny_m); d_hi[k] = (u32)((am & deny_m) >> 32);
            }
            u32 A16 = 0, D16 = 0;
#pragma unroll
            for (u32 rr = 0; rr < 4; ++rr) {
              const u32 r = 3u - rr;
              if (wave_ballot(((live16 >> (4u * r)) & 0xFu) != 0) == 0) { A16 <<= 4; D16 <<= 4; continue; }   // no lane has a live walk of its r-th role
              const u64 rmk = segm[32u + rc[r]];
              const u32 r_lo = (u32)rmk, r_hi = (u32)(rmk >> 32);
#pragma unroll
              for (u32 kk = 0; kk < 4; ++kk) {
                const u32 k = 3u - kk;
                const u32 ta = (r_hi & a_hi[k]) | (r_lo & a_lo[k]), td = (r_hi & d_hi[k]) | (r_lo & d_lo[k]);
                A16 = (A16 << 1) | nz01(ta);
                D16 = (D16 << 1) | nz01(td);
              }
            }
            A16 &= live16; D16 &= live16;
            has_allow |= A16;
            deny |= D16; S &= ~D16;
          }
          (void)wave_ballot(true);   // (the next segment's tables overwrite these)
        }
      } else if constexpr (STAGED) {
        if (have_bucket && bucket.y) {
          const u32 end = bucket.x + bucket.y;
          for (u32 base = bucket.x; base < end && wave_ballot(ing && S != 0) != 0; base += CBH_BLOCK) {   // bindings in order (check.go:295-414)
            const u32 n = end - base < CBH_BLOCK ? end - base : CBH_BLOCK;
            const bool mine = c.tid < n;
            FLAT_DBG(const u64 g0 = __builtin_readcyclecounter();)
            const StagedRec<16> sr = stage_rec<16>(t.rows, base + c.tid, mine);   // hot half + leaf slot (TblRowFull)
            const StagedRec<8> s2 = stage_rec<8>(t.rowleaf2, base + c.tid, mine && sr.w[2] != CBH_NONE);   // the derived-role condition's leaf
            // records some class present in the wave can match (TblRow: rm_lo = word 4, am_lo = word 6)
            u64 cand = wave_ballot(mine && (sr.w[4] & wave_rc) != 0 && (sr.w[6] & wave_ac) != 0);
            FLAT_DBG(dbg_rows += n; cyc_stage += (__builtin_readcyclecounter() - g0) * (u64)(cand != 0xdeadbeefull);)
            while (cand != 0) {
              const u32 i = ctz64(cand);
              cand &= cand - 1ull;
              const u32 row = base + i;
              const TblRowFull rf = staged_take<TblRowFull>(sr, i);
              const TblRow& rw = rf.hot;
              const u32 mact = ((rw.am_lo >> ac[0]) & 1u) | (((rw.am_lo >> ac[1]) & 1u) << 1) | (((rw.am_lo >> ac[2]) & 1u) << 2) | (((rw.am_lo >> ac[3]) & 1u) << 3);
              // one nibble per role (sign-extended 1-bit extracts), one bit per nibble for the actions; walks of other groups sit out
              const u32 mrole = ((0u - ((rw.rm_lo >> rc[0]) & 1u)) & 0xFu) | ((0u - ((rw.rm_lo >> rc[1]) & 1u)) & 0xF0u) |
                                ((0u - ((rw.rm_lo >> rc[2]) & 1u)) & 0xF00u) | ((0u - ((rw.rm_lo >> rc[3]) & 1u)) & 0xF000u);
              const u32 m = ing ? (mrole & (mact * 0x1111u) & S) : 0u;
              if (wave_ballot(m != 0) == 0) continue;
              if (EP) touch |= m;
              FLAT_DBG(++dbg_visits;)
              // the walks this record's effect applies to: all matched ones unless a condition says no.  The derived-role
              // condition comes first and the rule's own condition is evaluated only where that held (check.go:328-380);
              // each once per record and request, whatever the roles (check.go:316-340)
              u32 hit = m;
              if (rw.drcond != CBH_NONE) {
                const u32 how = (rw.flags & CBH_ROW_F_DRLEAF_EMBEDDED) ? 1u : (rw.flags & CBH_ROW_F_DRTREE_EMBEDDED) ? 2u : 0u;
                const u32 lv = leafish_memo(rw.drcond, how, [&] { return staged_take<LeafRec>(s2, i); }, hit != 0);
                err |= (lv & 2u) ? hit : 0u; unsup |= (lv & 8u) ? hit : 0u;
                hit = (lv & 1u) ? hit : 0u;
              }
              if (rw.cond != CBH_NONE && wave_ballot(hit != 0) != 0) {
                const u32 how = (rw.flags & CBH_ROW_F_LEAF_EMBEDDED) ? 1u : (rw.flags & CBH_ROW_F_TREE_EMBEDDED) ? 2u : 0u;
                const u32 lv = leafish(rw.cond, how, rf.leaf, hit != 0);
                err |= (lv & 2u) ? hit : 0u; unsup |= (lv & 8u) ? hit : 0u;
                hit = (lv & 1u) ? hit : 0u;
              }
              if ((rw.flags & 3u) == CBH_EFFECT_ALLOW) has_allow |= hit;
              else if ((rw.flags & 3u) == CBH_EFFECT_DENY) { deny |= hit; S &= ~hit; }   // ends these walks (check.go:392-403)
              (void)row;
            }
          }
        }
      } else {
        if (have_bucket && bucket.y) {
          const u32 last = bucket.x + bucket.y - 1u;
          TblRowFull nxt = uload_rec<TblRowFull>(t.rows, bucket.x);
          for (u32 row = bucket.x; row <= last; ++row) {   // bindings in order (check.go:295-414)
            const TblRowFull rf = nxt;   // hot half + leaf slot: one scalar load, issued one record ahead
            nxt = uload_rec<TblRowFull>(t.rows, row < last ? row + 1u : last);
            const TblRow& rw = rf.hot;
            FLAT_DBG(++dbg_rows;)
            if ((rw.rm_lo & wave_rc) == 0 || (rw.am_lo & wave_ac) == 0) continue;
            const u32 mact = ((rw.am_lo >> ac[0]) & 1u) | (((rw.am_lo >> ac[1]) & 1u) << 1) | (((rw.am_lo >> ac[2]) & 1u) << 2) | (((rw.am_lo >> ac[3]) & 1u) << 3);
            // one nibble per role (sign-extended 1-bit extracts), one bit per nibble for the actions; walks of other groups sit out
            const u32 mrole = ((0u - ((rw.rm_lo >> rc[0]) & 1u)) & 0xFu) | ((0u - ((rw.rm_lo >> rc[1]) & 1u)) & 0xF0u) |
                              ((0u - ((rw.rm_lo >> rc[2]) & 1u)) & 0xF00u) | ((0u - ((rw.rm_lo >> rc[3]) & 1u)) & 0xF000u);
            const u32 m = ing ? (mrole & (mact * 0x1111u) & S) : 0u;
            if (wave_ballot(m != 0) == 0) continue;
            if (EP) touch |= m;
            // the walks this record's effect applies to: all matched ones unless a condition says no.  The derived-role
            // condition comes first and the rule's own condition is evaluated only where that held (check.go:328-380);
            // each once per record and request, whatever the roles (check.go:316-340)
            u32 hit = m;
            if (rw.drcond != CBH_NONE) {
              const u32 how = (rw.flags & CBH_ROW_F_DRLEAF_EMBEDDED) ? 1u : (rw.flags & CBH_ROW_F_DRTREE_EMBEDDED) ? 2u : 0u;
              const u32 lv = leafish_memo(rw.drcond, how, [&] { return uload_rec<LeafRec>(t.rowleaf2, row); }, hit != 0);
              err |= (lv & 2u) ? hit : 0u; unsup |= (lv & 8u) ? hit : 0u;
              hit = (lv & 1u) ? hit : 0u;
            }
            if (rw.cond != CBH_NONE && wave_ballot(hit != 0) != 0) {
              const u32 how = (rw.flags & CBH_ROW_F_LEAF_EMBEDDED) ? 1u : (rw.flags & CBH_ROW_F_TREE_EMBEDDED) ? 2u : 0u;
              const u32 lv = leafish(rw.cond, how, rf.leaf, hit != 0);
              err |= (lv & 2u) ? hit : 0u; unsup |= (lv & 8u) ? hit : 0u;
              hit = (lv & 1u) ? hit : 0u;
            }
            if ((rw.flags & 3u) == CBH_EFFECT_ALLOW) has_allow |= hit;
            else if ((rw.flags & 3u) == CBH_EFFECT_DENY) { deny |= hit; S &= ~hit; }   // ends these walks (check.go:392-403)
          }
        }
      }
      if (EP) {
        if (MODE != 2 && have_bucket && bucket.y) g_pol = uload(&t.rows[(size_t)bucket.x * 16u + 3u]);   // TblRow.policy: a bucket = one policy
        if (ing && mydepth < max_depth) { ep_lds[(2u * mydepth) * CBH_BLOCK + c.tid] = touch; ep_lds[(2u * mydepth + 1u) * CBH_BLOCK + c.tid] = g_pol; }
      }
      const u32 ha = ing ? (has_allow & S) : 0u;   // check.go:416-425
      const u32 sp = (uload(&t.scope_flags[g_si]) >> 2) & 3u;
      if (sp == SP_REQUIRE_CONSENT) has_allow &= ~ha;
      else if (sp == SP_OVERRIDE_PARENT) { allow |= ha; S &= ~ha; }
      if (max_depth > 1u) {   // (a table of one scope: every walk is decided at depth 0 - the planes stay empty, the fold skips them)
        const u32 newly = S_before & ~S;
        dp0 |= (mydepth & 1u) ? newly : 0u; dp1 |= (mydepth & 2u) ? newly : 0u; dp2 |= (mydepth & 4u) ? newly : 0u; dp3 |= (mydepth & 8u) ? newly : 0u;
      }
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
  const u32 st_ok = (t.flags & CBH_MF_TRACE_ALL) ? CBH_ST_WANTS_TRACE : CBH_ST_OK;   // outputs: this kernel cannot tell which inputs have any (cerbos_hip.h)
#pragma unroll
  for (u32 k = 0; k < 4; ++k) {
    const u32 ak = (allow >> k) & 0x1111u, dk = (deny >> k) & 0x1111u;
    const u32 win = ak ? (ak & (0u - ak)) : (dk & (0u - dk));   // lowest role bit of the deciding kind
    const u32 wb = win << k;                                     // back at its position 4r + k
    u32 d = 0;
    if (max_depth > 1u) d = ((dp0 & wb) ? 1u : 0u) | ((dp1 & wb) ? 2u : 0u) | ((dp2 & wb) ? 4u : 0u) | ((dp3 & wb) ? 8u : 0u);
    pol[k] = win ? pol_hit : pol_none;
    scp[k] = CBH_NONE;
    if (win && k < act_cnt) scp[k] = chain8 ? (u32)chain_si8[d * CBH_BLOCK + c.tid] : chain_si[d * CBH_BLOCK + c.tid];
    eff4 |= (u32)(ak ? CBH_EFFECT_ALLOW : CBH_EFFECT_DENY) << (8 * k);   // NO_MATCH -> DENY (check.go:451-453)
    // an evaluation the reference would not have made - a role after the one that allowed - does not count
    const u32 seen = ak ? (((ak & (0u - ak)) << 1) - 1u) : 0xFFFFu;
    const u32 ek = (err >> k) & 0x1111u & seen, uk = (unsup >> k) & 0x1111u & seen;
    st4 |= (u32)(uk ? CBH_ST_UNSUPPORTED : (ek ? CBH_ST_CEL_ERROR : st_ok)) << (8 * k);
  }

  if (EP && want_ep && valid) {   // ---- effective policies: what the walks the reference really makes have touched
    u32 legit = 0;
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
      const u32 ak = (allow >> k) & 0x1111u;
      const u32 seen = ak ? (((ak & (0u - ak)) << 1) - 1u) : 0xFFFFu;
      legit |= ((walks >> k) & 0x1111u & seen) << k;
    }
    for (u32 d = 0; d < max_depth; ++d)
      if (ep_lds[(2u * d) * CBH_BLOCK + c.tid] & legit) ep_mark(o, b, req, ep_lds[(2u * d + 1u) * CBH_BLOCK + c.tid]);
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
      if (udir_find(t, BTYPE, g_ver, g_k, g_si, bucket)) {
        for (u32 d = bucket.z; d < bucket.z + bucket.w; ++d) {
          const TblDrx dx = uload_rec<TblDrx>(t.drx, d);
          const bool applies = ing && (dx.rm_lo & lane_rc) != 0;   // parent roles x the request's roles (check.go:244)
          if (wave_ballot(applies) == 0) continue;
          u32 lv = 1u;
          if (dx.cond != CBH_NONE) lv = leafish_memo(dx.cond, dx.flags & 3u, [&] { return dx.leaf; }, applies);
          if (applies) { if (lv & 1u) edr |= 1ull << dx.name; derr = derr || (lv & 2u) != 0; dr_unsup = dr_unsup || (lv & 8u) != 0; }
        }
      }
      const u32 up = uchain_next(t, uload(&t.scope_parent[g_si]), FLAG_RES);
      if (ing) { cur2 = up; ++d2; }
    }
    if (derr) {   // evaluation errors are a per-request fact: every action that is not UNSUPPORTED
      const u32 un = (st4 >> 1) & ~st4 & 0x01010101u;   // bytes that read 2
      st4 = (un << 1) | (~un & 0x01010101u);
    }
    if (dr_unsup) st4 = 0x02020202u;
  }

#ifdef CBH_PROFILE_CYCLES
  if (flags & CBH_F_DEBUG_CYCLES) {   // policy / scope words <- phase cycles, wall-clock (100 MHz) start / end, visit counts
    const u64 cyc4 = __builtin_readcyclecounter();
    pol[0] = (u32)(cyc1 - cyc0); pol[1] = (u32)cyc_eval; pol[2] = (u32)(cyc3 - cyc2); pol[3] = (u32)cyc_stage; (void)cyc4;
    scp[0] = (u32)rt0; scp[1] = (u32)__builtin_amdgcn_s_memrealtime(); scp[2] = dbg_rows | (dbg_evals << 16); scp[3] = dbg_rounds | (dbg_visits << 16);
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
// each wave of the group owns its slice of the column cache (CBH_CC_DWORDS)
#define CBH_FLAT_CTX(a, ka)                                                                                                       \
  const u32 ncc = cached_columns(&a);                                                                                             \
  Ctx c{a.t, a.b, a.now_ns, a.flags, threadIdx.x % CBH_BLOCK, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,  \
        (CBH_L u32*)cbh_dyn_lds + (threadIdx.x / CBH_BLOCK) * CBH_CC_DWORDS(ncc, (a.flags & CBH_FI_PACKED_TAGS) != 0), ncc, ka}
// batches of plain scalars (no int / uint / list / map attribute values): no call, ~64 VGPRs, 7-8 waves per SIMD
__global__ CBH_FLAT_ATTRS(7) void cbh_check_flat_kernel(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  CBH_FLAT_CTX(a, ka);
  flat_body<false, 0>(a, c);
}
// ... and for a table with derived roles: the same with the memo of condition outcomes (C3 +4.5 %, profiles/r06_ab_condition_memo.txt)
__global__ CBH_FLAT_ATTRS(7) void cbh_check_flat_kernel_dr(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  CBH_FLAT_CTX(a, ka);
  flat_body<false, 0, false, true>(a, c);
}
// any batch: the same walk with the call into the shared evaluator compiled in (4 waves per SIMD)
__global__ CBH_FLAT_ATTRS(4) void cbh_check_flat_kernel_any(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  CBH_FLAT_CTX(a, ka);
  flat_body<true, 0>(a, c);
}
// tables with long buckets (more than CBH_FLAT_STAGE_MIN rule records in one policy): the records arrive 64 at a time in
// the lanes' registers (stage_rec) instead of one scalar load per visit - 16 more VGPRs, 5 waves per SIMD
__global__ CBH_FLAT_ATTRS(5) void cbh_check_flat_kernel_staged(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  CBH_FLAT_CTX(a, ka);
  flat_body<false, 1>(a, c);
}
__global__ CBH_FLAT_ATTRS(4) void cbh_check_flat_kernel_any_staged(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  CBH_FLAT_CTX(a, ka);
  flat_body<true, 1>(a, c);
}
// tables with segments (cbh_blob.h CBH_SEC_SEGS) and long buckets: the mask walk - no record is visited
__global__ CBH_FLAT_ATTRS(4) void cbh_check_flat_kernel_masks(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  CBH_FLAT_CTX(a, ka);
  flat_body<false, 2>(a, c);
}
__global__ CBH_FLAT_ATTRS(3) void cbh_check_flat_kernel_any_masks(const KernelArgs a, const KernelArgs* __restrict__ ka) {
  CBH_FLAT_CTX(a, ka);
  flat_body<true, 2>(a, c);
}
// cbh_check_batch_trail on a flat table: the same six walks with the effective policies kept (flat_body EP)
#define CBH_FLAT_TRAIL_KERNEL(NAME, MINW, CALL, MODE)                                                                          \
  __global__ CBH_FLAT_ATTRS(MINW) void NAME(const KernelArgs a, const KernelArgs* __restrict__ ka) { CBH_FLAT_CTX(a, ka); flat_body<CALL, MODE, true>(a, c); }
CBH_FLAT_TRAIL_KERNEL(cbh_check_flat_trail_kernel, 6, false, 0)
CBH_FLAT_TRAIL_KERNEL(cbh_check_flat_trail_kernel_any, 4, true, 0)
CBH_FLAT_TRAIL_KERNEL(cbh_check_flat_trail_kernel_staged, 5, false, 1)
CBH_FLAT_TRAIL_KERNEL(cbh_check_flat_trail_kernel_any_staged, 3, true, 1)
CBH_FLAT_TRAIL_KERNEL(cbh_check_flat_trail_kernel_masks, 3, false, 2)
CBH_FLAT_TRAIL_KERNEL(cbh_check_flat_trail_kernel_any_masks, 3, true, 2)
#define CBH_FLAT_STAGE_MIN 32u
// the mask walk decides a table that has segments and long buckets (CBH_FLAT_MASKS=0: never, =1: whatever the buckets' length - tests, A/B)
static inline bool cbh_flat_use_masks(const void* segs, u32 max_bucket) {
  static const char* e = getenv("CBH_FLAT_MASKS");
  static const bool force_staged = getenv("CBH_FORCE_STAGED") != nullptr;
  if ((e && *e == '0') || force_staged) return false;
  return segs != nullptr && ((e && *e == '1') || max_bucket > CBH_FLAT_STAGE_MIN);
}

// Which kernel decides this batch: a flat one when table (CBH_MF_FLAT), batch shape (<= 4 actions and <= 4 roles per
// request; `plain_tags`: no attribute value is an int / uint / list / map - selects the variant without the evaluator call)
// and evaluation mode (not strict) allow it, else the general walk's instantiation for the table class.
// `threads` = the workgroup size to launch it with; dynamic LDS per wave = the column cache, plus - `flat` - the
// scope-chain scratch (cbh_flat_lds_bytes).
// dynamic LDS of one wave of the flat kernel: column cache + [max_depth][64] scope indices (bytes for a table of <= 256 scopes)
static inline size_t cbh_flat_chain_bytes(u32 table_max_depth, u32 table_scopes) {
  return (size_t)(table_max_depth < CBH_FLAT_MAX_DEPTH ? table_max_depth : CBH_FLAT_MAX_DEPTH) * CBH_BLOCK * (table_scopes <= 256u ? 1 : 4);
}
// ... and, once per workgroup, the two class tables (one byte per table string each) when they are staged in LDS
static inline size_t cbh_flat_class_bytes(u32 table_strings) {
  return table_strings <= CBH_FLAT_LDS_STRINGS ? (((size_t)2 * table_strings + 15) & ~(size_t)15) : 0;
}
// ... and, for the mask walk, a segment's class masks per wave
static inline size_t cbh_flat_mask_bytes(u32 threads) { return (size_t)(threads / CBH_BLOCK) * CBH_SEG_LDS_BYTES; }
static inline bool cbh_is_mask_kernel(cbh_check_kernel_fn fn) {
  return fn == cbh_check_flat_kernel_masks || fn == cbh_check_flat_kernel_any_masks || fn == cbh_check_flat_trail_kernel_masks || fn == cbh_check_flat_trail_kernel_any_masks;
}
// the instantiation of a flat kernel that also keeps the effective policies (cbh_check_batch_trail), and its extra LDS
static inline cbh_check_kernel_fn cbh_flat_trail_variant(cbh_check_kernel_fn fn) {
  return (fn == cbh_check_flat_kernel || fn == cbh_check_flat_kernel_dr) ? cbh_check_flat_trail_kernel : fn == cbh_check_flat_kernel_any ? cbh_check_flat_trail_kernel_any
       : fn == cbh_check_flat_kernel_staged ? cbh_check_flat_trail_kernel_staged : fn == cbh_check_flat_kernel_any_staged ? cbh_check_flat_trail_kernel_any_staged
       : fn == cbh_check_flat_kernel_masks ? cbh_check_flat_trail_kernel_masks : fn == cbh_check_flat_kernel_any_masks ? cbh_check_flat_trail_kernel_any_masks : fn;
}
static inline bool cbh_is_flat_trail_kernel(cbh_check_kernel_fn fn) {
  return fn == cbh_check_flat_trail_kernel || fn == cbh_check_flat_trail_kernel_any || fn == cbh_check_flat_trail_kernel_staged ||
         fn == cbh_check_flat_trail_kernel_any_staged || fn == cbh_check_flat_trail_kernel_masks || fn == cbh_check_flat_trail_kernel_any_masks;
}
static inline size_t cbh_flat_trail_bytes(u32 threads, u32 table_max_depth) {
  const u32 depth = table_max_depth < CBH_FLAT_MAX_DEPTH ? table_max_depth : CBH_FLAT_MAX_DEPTH;
  return (size_t)(threads / CBH_BLOCK) * depth * 2u * CBH_BLOCK * 4u;
}
static inline cbh_check_kernel_fn cbh_pick_kernel(u32 table_flags, u32 n_derived_roles, bool has_globs, u32 max_actions, u32 max_roles, bool plain_tags,
                                                  u32 eval_flags, u32 max_bucket, u32* threads, bool* flat, bool masks = false) {
  *flat = (table_flags & CBH_MF_FLAT) && max_actions <= 4 && max_roles <= 4 && !(eval_flags & CBH_F_STRICT_EVALUATION);
  if (*flat) {
    *threads = CBH_FLAT_THREADS;
    const bool staged = max_bucket > CBH_FLAT_STAGE_MIN;
    if (masks) return (plain_tags && (table_flags & CBH_MF_FLAT_CLOSED)) ? cbh_check_flat_kernel_masks : cbh_check_flat_kernel_any_masks;
    if (plain_tags && (table_flags & CBH_MF_FLAT_CLOSED)) return staged ? cbh_check_flat_kernel_staged : n_derived_roles ? cbh_check_flat_kernel_dr : cbh_check_flat_kernel;
    return staged ? cbh_check_flat_kernel_any_staged : cbh_check_flat_kernel_any;
  }
  *threads = CBH_BLOCK;
  return cbh_pick_check_kernel(table_flags, n_derived_roles, has_globs, max_actions);
}
