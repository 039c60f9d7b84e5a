// Table image (blob) parsing shared by the product library and the test-only host simulation.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "cbh_vm.h"

#ifndef NFA_MAXW
#define NFA_MAXW 8
#endif

// Fills `d` / `meta` from a table image.  `base` is the address the image lives at (device or host),
// `host_copy` a readable copy of it.  Returns nullptr on success or a static error string.
#if !defined(__HIP_DEVICE_COMPILE__)   // host code: in the device pass the struct members are address-space qualified
static inline const char* cbh_parse_image(TableDev& d, std::vector<uint32_t>& meta_out, const uint8_t* base, const uint8_t* host_copy, size_t len) {
  if (len < sizeof(CbhBlobHeader)) return ("blob too small");
  const CbhBlobHeader* h = reinterpret_cast<const CbhBlobHeader*>(host_copy);
  if (h->magic != CBH_BLOB_MAGIC) return ("bad blob magic");
  if (h->version != CBH_BLOB_VERSION) return ("blob version mismatch: re-lower the rule table");
  if (h->total_len != len) return ("blob length mismatch");
  const CbhBlobSection* secs = reinterpret_cast<const CbhBlobSection*>(host_copy + sizeof(CbhBlobHeader));
    auto find = [&](uint32_t id) -> const CbhBlobSection* {
    for (uint32_t i = 0; i < h->n_sections; ++i) if (secs[i].id == id) return &secs[i];
    return nullptr;
  };
  auto dptr = [&](uint32_t id) -> const uint8_t* {
    const CbhBlobSection* s = find(id);
    return s ? base + s->offset : nullptr;
  };
  const CbhBlobSection* ms = find(CBH_SEC_META);
  if (!ms || ms->nbytes < CBH_META_N * 4) return ("blob has no META section");
  for (uint32_t i = 0; i < h->n_sections; ++i)
    if (secs[i].offset + secs[i].nbytes > len) return ("blob section out of range");
  meta_out.assign(reinterpret_cast<const uint32_t*>(host_copy + ms->offset),
                 reinterpret_cast<const uint32_t*>(host_copy + ms->offset) + CBH_META_N);
  const uint32_t* m = meta_out.data();
  if (m[CBH_M_MAX_STACK] > CBH_STACK_DEPTH) return ("a CEL program needs a deeper operand stack than the device provides");
  if (m[CBH_M_MAX_LOCALS] > CBH_MAX_LOCALS) return ("a CEL program needs more comprehension locals than the device provides");
  d.str_off = (const u32*)dptr(CBH_SEC_STR_OFF); d.str_bytes = dptr(CBH_SEC_STR_BYTES);
  d.scope_parent = (const u32*)dptr(CBH_SEC_SCOPE_PARENT); d.scope_flags = (const u32*)dptr(CBH_SEC_SCOPE_FLAGS);
  d.hash = (const CbhHashSlot*)dptr(CBH_SEC_HASH); d.hash_mask = m[CBH_M_HASH_MASK];
  d.rows = (const u32*)dptr(CBH_SEC_ROWS); d.n_rows = m[CBH_M_NROWS];
  d.rowleaf2 = (const u32*)dptr(CBH_SEC_ROWLEAF2); d.drx = (const u32*)dptr(CBH_SEC_DRX);
  if ((m[CBH_M_NROWS] && !d.rowleaf2) || (m[CBH_M_NDR] && !d.drx)) return ("blob is missing the flat-kernel sections");
  d.regex = (const u32*)dptr(CBH_SEC_REGEX);
  if (!d.regex) return ("blob is missing the regex section");
  // the trace pass's sections travel together or not at all (a table lowered without them serves decisions only)
  d.trace_rows = (const u32*)dptr(CBH_SEC_TRACE_ROWS); d.trace_dr = (const u32*)dptr(CBH_SEC_TRACE_DR);
  d.trace_rp = (const u32*)dptr(CBH_SEC_TRACE_RP); d.trace_pool = (const u32*)dptr(CBH_SEC_TRACE_POOL);
  if (d.trace_pool) {
    const CbhBlobSection *tr = find(CBH_SEC_TRACE_ROWS), *td = find(CBH_SEC_TRACE_DR), *tp = find(CBH_SEC_TRACE_RP);
    if (!tr || !td || !tp || tr->nbytes < (uint64_t)m[CBH_M_NROWS] * 32 || td->nbytes < (uint64_t)m[CBH_M_NDR] * 16 ||
        tp->nbytes < (uint64_t)m[CBH_M_NRPROWS] * 32)
      return ("blob trace sections are incomplete");
  }
  d.rowpat = (const u32*)dptr(CBH_SEC_ROWPAT);
  d.rowx = (const u32*)dptr(CBH_SEC_ROWX); d.rpx = (const u32*)dptr(CBH_SEC_RPX);
  {
    const CbhBlobSection *rx = find(CBH_SEC_ROWX), *px = find(CBH_SEC_RPX);
    if (!rx || !px || rx->nbytes < (uint64_t)m[CBH_M_NROWS] * 32 || px->nbytes < (uint64_t)m[CBH_M_NRPROWS] * 64) return ("blob is missing the walk sections");
  }
  d.gslots_generic = m[CBH_M_GSLOTS_GENERIC]; d.gslots_all = m[CBH_M_GSLOTS_ALL];
  d.inline_cols = m[CBH_M_INLINE_COLS]; d.sens_cols = m[CBH_M_SENS_COLS]; d.q_sites = m[CBH_M_Q_SITES];
  d.seg_info = m[CBH_M_SEGS]; d.segs = (const u32*)dptr(CBH_SEC_SEGS); d.leafpool = (const u32*)dptr(CBH_SEC_LEAFPOOL);
  if ((d.seg_info & CBH_MSEG_PRESENT) && (!d.segs || !d.leafpool || !(m[CBH_M_FLAGS] & CBH_MF_FLAT))) return ("blob segment sections are inconsistent");
  if ((d.seg_info & 0xFFu) > CBH_SEG_RECORDS) return ("blob leaf pool too large");
  if (!(d.seg_info & CBH_MSEG_PRESENT)) { d.segs = nullptr; d.leafpool = nullptr; }
  d.str_wflags = dptr(CBH_SEC_STR_WFLAGS);
  if (!d.str_wflags) return ("blob is missing the string flag section");
  if (d.inline_cols > CBH_CACHE_COLS || d.inline_cols > m[CBH_M_NCOLUMNS]) return ("blob inline column count out of range");
  if (d.gslots_generic > d.gslots_all || d.gslots_all > CBH_W2_MAX_GSLOTS) return ("blob evaluation-site slots out of range");
  if (m[CBH_M_NROWS] && !d.rowpat) return ("blob is missing the row pattern section");
  d.rprows = (const u32*)dptr(CBH_SEC_RPROWS); d.n_rprows = m[CBH_M_NRPROWS];
  d.pool = (const u32*)dptr(CBH_SEC_U32POOL);
  d.dr = (const u32*)dptr(CBH_SEC_DR); d.n_dr = m[CBH_M_NDR];
  d.code = (const u32*)dptr(CBH_SEC_CODE);
  d.const_tag = dptr(CBH_SEC_CONST_TAG); d.const_val = (const u64*)dptr(CBH_SEC_CONST_VAL);
  d.theap_tag = dptr(CBH_SEC_THEAP_TAG); d.theap_val = (const u64*)dptr(CBH_SEC_THEAP_VAL);
  d.const_rec = (const u32*)dptr(CBH_SEC_CONST_REC); d.theap_rec = (const u32*)dptr(CBH_SEC_THEAP_REC);
  if (!d.const_rec || !d.theap_rec) return ("blob is missing the constant record sections");
  d.role_class = dptr(CBH_SEC_ROLE_CLASS);
  if (!d.role_class) return ("blob is missing the role class section");
  d.action_class = dptr(CBH_SEC_ACTION_CLASS);
  if (!d.action_class) return ("blob is missing the action class section");
  d.gbits = (const u64*)dptr(CBH_SEC_GBITS); d.K = m[CBH_M_NSTRINGS];
  d.nfa[0] = (const u64*)dptr(CBH_SEC_NFA_ACTION); d.nfa[1] = (const u64*)dptr(CBH_SEC_NFA_ROLE); d.nfa[2] = (const u64*)dptr(CBH_SEC_NFA_KIND);
  d.nfa_words[0] = m[CBH_M_NFA_WORDS_ACTION]; d.nfa_words[1] = m[CBH_M_NFA_WORDS_ROLE]; d.nfa_words[2] = m[CBH_M_NFA_WORDS_KIND];
  for (int i = 0; i < 3; ++i) if (d.nfa_words[i] > NFA_MAXW) return ("glob NFA wider than the device supports");
  d.flags = m[CBH_M_FLAGS];
  {   // longest scope chain: sizes the flat kernel's per-wave chain scratch (cbh_check_flat.h)
    const CbhBlobSection* ps = find(CBH_SEC_SCOPE_PARENT);
    const uint32_t ns = m[CBH_M_NSCOPES];
    if (!ps || ps->nbytes < (uint64_t)ns * 4) return ("blob scope section too small");
    const uint32_t* par = reinterpret_cast<const uint32_t*>(host_copy + ps->offset);
    uint32_t deepest = 1;
    for (uint32_t s = 0; s < ns; ++s) {
      uint32_t n = 1, at = par[s];
      while (at != CBH_NONE && at < ns && n <= ns) { ++n; at = par[at]; }
      if (n > ns) return ("scope parents form a cycle");
      if (n > deepest) deepest = n;
    }
    d.max_depth = deepest;
    d.n_scopes = ns;
    for (uint32_t s = 0; s < ns; ++s)   // the flat kernel's merge order relies on it (cbh_check_flat.h)
      if (par[s] != CBH_NONE && par[s] >= s) return ("scopes are not numbered parents first");
  }
  if (!d.hash || !d.code || !d.str_off || !d.scope_flags) return ("blob is missing required sections");
  {   // the longest resource-policy bucket: tables with long ones walk them 64 records at a time (cbh_check_flat.h stage_rec)
    const CbhBlobSection* hs = find(CBH_SEC_HASH);
    if (!hs || hs->nbytes < ((uint64_t)d.hash_mask + 1) * sizeof(CbhHashSlot)) return ("blob directory too small");
    const CbhHashSlot* slots = reinterpret_cast<const CbhHashSlot*>(host_copy + hs->offset);
    d.max_bucket = 0;
    for (uint64_t i = 0; i <= d.hash_mask; ++i)
      if (slots[i].k0 == CBH_B_RESOURCE && slots[i].v1 > d.max_bucket) d.max_bucket = slots[i].v1;
    // the segments the mask walk follows (cbh_check_flat.h): every bucket's chain of blocks stays inside the section, item and
    // leaf numbers inside what the kernel keeps one bit each for
    if (d.seg_info & CBH_MSEG_PRESENT) {
      const CbhBlobSection* ss = find(CBH_SEC_SEGS);
      const CbhBlobSection* ls = find(CBH_SEC_LEAFPOOL);
      if (ss->nbytes < 64ull * CBH_SEG_TAIL_PAD16) return ("blob segment section too small");
      const uint64_t seg16 = ss->nbytes / 64 - CBH_SEG_TAIL_PAD16, pool_n = d.seg_info & 0xFFu;   // (the tail: what the unconditional descriptor loads of the last block may touch)
      if (ls->nbytes < ((pool_n + 3) / 4) * 64) return ("blob leaf pool too small");
      const bool pooled = (d.seg_info & CBH_MSEG_POOLED) != 0;
      const uint32_t* sw = reinterpret_cast<const uint32_t*>(host_copy + ss->offset);
      for (uint64_t i = 0; i <= d.hash_mask; ++i) {
        if (slots[i].k0 != CBH_B_RESSEG) continue;
        uint64_t at = slots[i].v0;
        for (uint32_t sgi = 0; sgi < slots[i].v1; ++sgi) {
          if (at + CBH_SEG_FIXED_DWORDS / 16 > seg16) return ("blob segment out of range");
          const CbhSegHdr* hd = reinterpret_cast<const CbhSegHdr*>(sw + at * 16);
          const uint64_t size = (uint64_t)hd->size16 * 16, off_refs = hd->off_refs_complex & 0xFFFFu, off_cx = hd->off_refs_complex >> 16;
          if (hd->n_items > CBH_SEG_RECORDS || hd->n_leaves > CBH_SEG_RECORDS || hd->n_records > CBH_SEG_RECORDS || hd->n_complex > CBH_SEG_COMPLEX ||
              at + hd->size16 > seg16 || (pooled && hd->n_leaves) || CBH_SEG_FIXED_DWORDS + 2ull * hd->n_items > size ||
              off_refs + hd->n_items > size || (off_cx & 15u) || off_cx + 16ull * hd->n_complex > size || (hd->off_leaves & 15u) ||
              hd->off_leaves + 16ull * ((hd->n_leaves + 3) / 4) > size)
            return ("blob segment header out of range");
          const uint32_t n_leaf = pooled ? (uint32_t)pool_n : hd->n_leaves;
          const uint8_t* rec = reinterpret_cast<const uint8_t*>(sw + at * 16 + 144);
          const uint64_t simple_c = hd->simple_c_lo | ((uint64_t)hd->simple_c_hi << 32), simple_d = hd->simple_d_lo | ((uint64_t)hd->simple_d_hi << 32);
          const CbhSegDesc* ds = reinterpret_cast<const CbhSegDesc*>(sw + at * 16 + CBH_SEG_FIXED_DWORDS);
          for (uint32_t k = 0; k < CBH_SEG_RECORDS; ++k)
            for (int which = 0; which < 2; ++which) {
              if (!(((which ? simple_d : simple_c) >> k) & 1)) continue;
              const uint32_t itx = rec[which * 64 + k];
              if (itx >= hd->n_items) return ("blob segment record names no item");
              const uint32_t n = ds[itx].flags & 7u;
              if (n == 0 || n > 4 || (ds[itx].flags & ~0x1Fu)) return ("blob segment item descriptor out of range");
              for (uint32_t j = 0; j < n; ++j) if (ds[itx].leaf[j] >= n_leaf) return ("blob segment leaf number out of range");
            }
          const CbhSegItem* it = reinterpret_cast<const CbhSegItem*>(sw + at * 16 + off_cx);
          for (uint32_t k = 0; k < hd->n_complex; ++k) {
            if ((it[k].how & 3u) > 2 || (it[k].how & ~0x303u) || it[k].n_leaves > 8 || it[k].id >= hd->n_items) return ("blob segment item out of range");
            for (uint32_t j = 0; j < ((it[k].how & 3u) ? it[k].n_leaves : 0u); ++j)
              if (((j < 4 ? it[k].leaf_idx[0] >> (8 * j) : it[k].leaf_idx[1] >> (8 * (j - 4))) & 0xFFu) >= n_leaf) return ("blob segment leaf number out of range");
          }
          at += hd->size16;
        }
      }
    }
  }
  return nullptr;
}
#endif  // !__HIP_DEVICE_COMPILE__

