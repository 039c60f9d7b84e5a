// Host side of the device flattener (cbh_wire.h), shared by the library (cbh_engine.hip) and the test-only host simulation:
// what is built once per table from the image - the hash index of the table's strings, the scope index per string, the
// attribute paths of the columns - and the sizing rules of a call.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "cbh_wire.h"

struct WireIndexHost {
  std::vector<u64> tix; u32 tix_mask = 0;
  std::vector<u32> scope_of_sid;
  std::vector<WireCol> cols; std::vector<u8> col_keys;
  // the device assembler (cbh_wire_out_*): policy keys (CBH_P_TABLE ids) then derived-role names, and where the scopes' string ids sit in the image
  std::vector<u32> name_off; std::vector<u8> name_bytes; u32 n_policies = 0, n_dr = 0; u64 scope_sid_offset = 0; u32 n_scopes = 0;
  const char* why_not = nullptr;   // the table's inputs cannot be flattened on the device (every batch goes through libcerbos_ingest.so)
};

// `image` = a readable copy of the table image (already validated by cbh_parse_image).  Returns nullptr or an error text.
static inline const char* cbh_wire_index_build(WireIndexHost& w, const uint8_t* image, size_t len, const std::vector<uint32_t>& meta) {
  const CbhBlobHeader* h = reinterpret_cast<const CbhBlobHeader*>(image);
  const CbhBlobSection* secs = reinterpret_cast<const CbhBlobSection*>(image + sizeof(CbhBlobHeader));
  auto find = [&](uint32_t id) -> const CbhBlobSection* {
    for (uint32_t i = 0; i < h->n_sections; ++i)
      if (secs[i].id == id) return (secs[i].offset <= len && secs[i].nbytes <= len - secs[i].offset) ? &secs[i] : nullptr;
    return nullptr;
  };
  const CbhBlobSection *so = find(CBH_SEC_STR_OFF), *sb = find(CBH_SEC_STR_BYTES), *ss = find(CBH_SEC_SCOPE_SID), *sc = find(CBH_SEC_COLUMN_PATHS);
  if (!so || !sb || !ss || !sc) return "image is missing a section the device flattener needs";
  const u32 K = meta[CBH_M_NSTRINGS], ns = meta[CBH_M_NSCOPES], ncol = meta[CBH_M_NCOLUMNS];
  if (((u64)K + 1) * 4 > so->nbytes || (u64)ns * 4 > ss->nbytes) return "image string / scope sections too short";
  const u32* off = reinterpret_cast<const u32*>(image + so->offset);
  const u8* bytes = image + sb->offset;
  if (off[K] > sb->nbytes) return "image string bytes too short";
  size_t cap = 64;
  while (cap < 2 * (size_t)K) cap <<= 1;
  w.tix.assign(2 * cap, 0); w.tix_mask = (u32)cap - 1;    // slot i = tix[2 i], tix[2 i + 1] (cbh_wire.h w_table_sid)
  for (u32 i = 0; i < K; ++i) {
    if (off[i + 1] < off[i] || off[i + 1] > off[K]) return "image string offsets are not monotonic";
    const u32 hsh = cbh_wire_hash(bytes + off[i], off[i + 1] - off[i]);
    u32 at = hsh & w.tix_mask;
    while (w.tix[2 * (size_t)at]) at = (at + 1) & w.tix_mask;
    w.tix[2 * (size_t)at] = ((u64)hsh << 32) | (u64)(i + 1);
    w.tix[2 * (size_t)at + 1] = ((u64)off[i] << 32) | (u64)(off[i + 1] - off[i]);
  }
  // the device compares strings eight bytes at a time: the pool must be readable CBH_WIRE_SLACK bytes beyond its last string
  // (blob.py always writes sections behind the pool; an image that ends with it is still a valid table: only the device road of the
  // wire format is closed for it - every batch goes through libcerbos_ingest.so - the table itself loads and decides.)
  const bool pool_at_end = (u64)sb->offset + off[K] + CBH_WIRE_SLACK > len;
  w.scope_of_sid.assign(K ? K : 1, CBH_NONE);
  const u32* ssid = reinterpret_cast<const u32*>(image + ss->offset);
  for (u32 i = 0; i < ns; ++i) { if (ssid[i] >= K) return "image scope string id out of range"; w.scope_of_sid[ssid[i]] = i; }
  const u8* p = image + sc->offset; const u8* e = p + sc->nbytes;
  w.cols.clear(); w.col_keys.clear(); w.why_not = nullptr;
  if (pool_at_end) w.why_not = "the image ends within the device flattener's read-ahead of the string pool";
  for (u32 c = 0; c < ncol; ++c) {
    if (e - p < 2) return "image column path section truncated";
    WireCol col; memset(&col, 0, sizeof(col));
    col.root = p[0]; const u32 nk = p[1]; p += 2;
    if (col.root > 4) return "image column root out of range";
    if (nk > CBH_WIRE_MAX_KEYS) w.why_not = "a column path is deeper than the device flattener follows";
    col.nk = nk > CBH_WIRE_MAX_KEYS ? CBH_WIRE_MAX_KEYS : nk;
    for (u32 k = 0; k < nk; ++k) {
      if (e - p < 2) return "image column path section truncated";
      const u32 l = p[0] | (p[1] << 8); p += 2;
      if ((u32)(e - p) < l) return "image column path section truncated";
      if (k < CBH_WIRE_MAX_KEYS) { col.key_off[k] = (u32)w.col_keys.size(); col.key_len[k] = l; w.col_keys.insert(w.col_keys.end(), p, p + l); }
      p += l;
    }
    col.key_hash0 = nk ? cbh_wire_hash(w.col_keys.data() + col.key_off[0], col.key_len[0]) : 0u;
    w.cols.push_back(col);
  }
  w.scope_sid_offset = ss->offset; w.n_scopes = ns;
  w.name_off.assign(1, 0); w.name_bytes.clear(); w.n_policies = w.n_dr = 0;
  if (const CbhBlobSection* sn = find(CBH_SEC_HOST_NAMES)) {
    const u8* q = image + sn->offset; const u8* qe = q + sn->nbytes;
    for (u32* cnt_out : {&w.n_policies, &w.n_dr}) {
      if (qe - q < 4) return "image host name section truncated";
      u32 cnt; memcpy(&cnt, q, 4); q += 4;
      for (u32 k = 0; k < cnt; ++k) {
        if (qe - q < 2) return "image host name section truncated";
        const u32 l = q[0] | (q[1] << 8); q += 2;
        if ((u32)(qe - q) < l) return "image host name section truncated";
        w.name_bytes.insert(w.name_bytes.end(), q, q + l); q += l;
        w.name_off.push_back((u32)w.name_bytes.size());
      }
      *cnt_out = cnt;
    }
  } else return "image is missing the host name section";
  w.name_bytes.insert(w.name_bytes.end(), CBH_WIRE_SLACK, 0);   // (read eight bytes at a time)
  w.col_keys.insert(w.col_keys.end(), CBH_WIRE_SLACK, 0);
  if (w.cols.empty()) { WireCol z; memset(&z, 0, sizeof(z)); w.cols.push_back(z); }
  return nullptr;
}

// slots of the batch-local dictionary for n messages (a power of two; the caller quadruples it when a fill reports CBH_WF_DICT_FULL)
static inline u32 cbh_wire_dict_slots(u32 n) { u64 c = 4096; while (c < 8ull * n && c < (1ull << 30)) c <<= 1; return (u32)c; }
// first guess of the heap entries a call needs (the fill counts what it wanted: one re-run at the exact size if this was short)
static inline u32 cbh_wire_heap_guess(u64 message_bytes) { const u64 g = message_bytes / 8 + 4096; return g > 0x3FFFFFFFull ? 0x3FFFFFFFu : (u32)g; }
static inline void cbh_wire_stats_init(WireStats& s) { memset(&s, 0, sizeof(s)); s.wide_lo = CBH_NONE; s.first_bad = CBH_NONE; }
