// cbh_ingest.cpp - libcerbos_ingest.so: serialized enginev1.CheckInput -> cbh_batch (include/cerbos_ingest.h).
//
// Host-only C++ (no HIP, no protobuf runtime): the wire format is walked in place, attribute columns are
// looked up directly in the serialized google.protobuf.Struct maps, and only the values a column selects
// are materialised (scalars inline, lists / maps as tapes in the batch heap).  The interning order, heap
// order and routing sort are those of cerbos_amd/flatten.py - tests/test_ingest.py compares every array.
//
// Reference behaviour restated (never copied):
//   internal/ruletable/check.go:536-554     checkInputToRequest (which fields a condition can see)
//   internal/ruletable/check.go:101-117     effective scope / policy version of principal and resource
//   internal/evaluator/evaluator.go:108-122 defaults (EvalParams.DefaultScope / DefaultPolicyVersion)
//   internal/namer/namer.go:213-218         SanitizedResource; :77-87 ScopeParents; :276-278 ScopeValue
//   api/public/cerbos/engine/v1/engine.proto:130-200 field numbers of CheckInput / Resource / Principal / AuxData
#include <algorithm>
#include <charconv>
#include <cmath>
#include <functional>
#include <map>
#include <set>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/cerbos_ingest.h"
#include "cbh_blob.h"

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }

using u8 = uint8_t; using u32 = uint32_t; using u64 = uint64_t;

// ---- protobuf wire walking -------------------------------------------------------------------------
struct Span { const u8* p = nullptr; const u8* e = nullptr; bool empty() const { return p >= e; } };
struct Field { u32 num = 0; u32 wt = 0; u64 v = 0; Span s{}; };   // v: varint / fixed value, s: length-delimited payload

inline bool varint(Span& s, u64& out) {
  if (s.p < s.e && !(*s.p & 0x80)) { out = *s.p++; return true; }   // one byte: every tag and almost every length
  u64 r = 0;
  for (int sh = 0; sh < 64 && s.p < s.e; sh += 7) {
    u8 b = *s.p++;
    r |= (u64)(b & 0x7F) << sh;
    if (!(b & 0x80)) { out = r; return true; }
  }
  return false;
}

// Next field of a message; false at the end or on malformed input (`bad` set).
inline bool next(Span& s, Field& f, bool& bad) {
  if (s.empty()) return false;
  u64 key;
  if (!varint(s, key)) { bad = true; return false; }
  f.num = (u32)(key >> 3); f.wt = (u32)(key & 7);
  if (f.wt == 2) {   // length-delimited first: strings and sub-messages are most of a CheckInput
    u64 n;
    if (!varint(s, n) || n > (u64)(s.e - s.p)) { bad = true; return false; }
    f.s = Span{s.p, s.p + n}; s.p += n; return true;
  }
  switch (f.wt) {
    case 0: if (!varint(s, f.v)) { bad = true; return false; } return true;
    case 1: if (s.e - s.p < 8) { bad = true; return false; } memcpy(&f.v, s.p, 8); s.p += 8; return true;
    case 5: if (s.e - s.p < 4) { bad = true; return false; } { u32 w; memcpy(&w, s.p, 4); f.v = w; } s.p += 4; return true;
    case 2: {
      u64 n;
      if (!varint(s, n) || n > (u64)(s.e - s.p)) { bad = true; return false; }
      f.s = Span{s.p, s.p + n}; s.p += n; return true;
    }
    default: bad = true; return false;
  }
}

std::string_view sv(Span s) { return std::string_view((const char*)s.p, (size_t)(s.e - s.p)); }

// map<string, google.protobuf.Value> entry: key = 1, value = 2
struct Entry { Span key{nullptr, nullptr}; Span val{nullptr, nullptr}; };
inline bool entry(Span e, Entry& out, bool& bad) {
  // the shape every encoder writes - key (field 1) then value (field 2), both with one-byte lengths, nothing else - is read
  // without the field loop; anything else takes the general walk below
  {
    const u8* p = e.p; const size_t n = (size_t)(e.e - e.p);
    if (n >= 4 && p[0] == 0x0A && !(p[1] & 0x80)) {
      const size_t kl = p[1];
      if (2 + kl + 2 <= n && p[2 + kl] == 0x12 && !(p[3 + kl] & 0x80) && 4 + kl + (size_t)p[3 + kl] == n) {
        out.key = Span{p + 2, p + 2 + kl};
        out.val = Span{p + 4 + kl, e.e};
        return true;
      }
    }
  }
  Field f;
  while (next(e, f, bad)) {
    if (f.num == 1 && f.wt == 2) out.key = f.s;
    else if (f.num == 2 && f.wt == 2) out.val = f.s;
  }
  return !bad;
}

// Looks `key` up in the map field `fnum` of message `msg` (last entry wins, as protobuf maps decode).
bool map_get(Span msg, u32 fnum, std::string_view key, Span& val, bool& bad) {
  Field f; bool found = false;
  while (next(msg, f, bad)) {
    if (f.num != fnum || f.wt != 2) continue;
    Entry en;
    if (!entry(f.s, en, bad)) return false;
    if (sv(en.key) == key) { val = en.val; found = true; }
  }
  return found;
}

// google.protobuf.Value oneof: null 1, number 2 (double), string 3, bool 4, struct 5, list 6.  The last
// field present wins; an empty message is null.
struct Val { u32 kind = 1; u64 v = 0; Span s{nullptr, nullptr}; };
inline bool value(Span m, Val& out, bool& bad) {
  // a field counts only with the wire type its declaration has (null / bool: varint, number: fixed64,
  // string / struct / list: length-delimited); anything else is an unknown field, skipped as protobuf does
  static const u8 want_wt[7] = {0xFF, 0, 1, 2, 0, 2, 2};
  // one field filling the whole message - what an encoder writes for a scalar - without the field loop
  {
    const u8* p = m.p; const size_t n = (size_t)(m.e - m.p);
    if (n == 9 && p[0] == 0x11) { out.kind = 2; memcpy(&out.v, p + 1, 8); out.s = Span{}; return true; }                      // number_value
    if (n >= 2 && p[0] == 0x1A && !(p[1] & 0x80) && 2 + (size_t)p[1] == n) { out.kind = 3; out.v = 0; out.s = Span{p + 2, m.e}; return true; }   // string_value
    if (n == 2 && p[0] == 0x20 && !(p[1] & 0x80)) { out.kind = 4; out.v = p[1]; out.s = Span{}; return true; }              // bool_value
  }
  Field f;
  while (next(m, f, bad)) {
    if (f.num >= 1 && f.num <= 6 && f.wt == want_wt[f.num]) {
      out.kind = f.num; out.v = f.wt == 2 ? 0 : f.v; out.s = f.wt == 2 ? f.s : Span{};
    }
  }
  return !bad;
}

// ---- string dictionaries ---------------------------------------------------------------------------------
inline u64 hash_bytes(std::string_view s) {
  const char* p = s.data(); size_t n = s.size();
  u64 h = 0x9E3779B97F4A7C15ull ^ (n * 0xff51afd7ed558ccdull);
  while (n >= 8) { u64 w; memcpy(&w, p, 8); h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 29; p += 8; n -= 8; }
  if (n) { u64 w = 0; memcpy(&w, p, n); h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 29; }
  return h ^ (h >> 32);
}

// Open-addressing set of string ids; the strings themselves live wherever `at(id)` finds them.
struct StrIndex {
  struct Slot { u32 h; u32 id1; };   // id1 = id + 1, 0 = empty
  std::vector<Slot> slots;
  u32 used = 0;
  StrIndex() : slots(64, Slot{0, 0}) {}
  template <class At> bool find(std::string_view s, u64 h, At at, u32& id) const {
    const u32 mask = (u32)slots.size() - 1, h32 = (u32)h;
    for (u32 i = h32 & mask;; i = (i + 1) & mask) {
      const Slot& sl = slots[i];
      if (!sl.id1) return false;
      if (sl.h == h32) {
        const std::string_view c = at(sl.id1 - 1);
        if (c.size() == s.size() && (s.empty() || std::memcmp(c.data(), s.data(), s.size()) == 0)) { id = sl.id1 - 1; return true; }
      }
    }
  }
  void insert(u64 h, u32 id) {   // the caller knows the string is absent
    if ((used + 1) * 2 > slots.size()) {
      std::vector<Slot> old(slots.size() * 2, Slot{0, 0});
      old.swap(slots);
      for (const Slot& sl : old) if (sl.id1) place(sl);
    }
    place(Slot{(u32)h, id + 1});
    ++used;
  }
  void reserve(size_t n) {
    size_t want = 64;
    while (want < 4 * n) want *= 2;
    if (want > slots.size() && !used) slots.assign(want, Slot{0, 0});
  }
  void place(Slot s) {
    const u32 mask = (u32)slots.size() - 1;
    u32 i = s.h & mask;
    while (slots[i].id1) i = (i + 1) & mask;
    slots[i] = s;
  }
};

struct Column { u32 root; std::vector<std::string> keys; };
enum { SPAN_REQUEST_ID, SPAN_P_ID, SPAN_P_VERSION, SPAN_R_KIND, SPAN_R_VERSION, SPAN_R_ID, IN_SPAN_N };

// A CEL value as the trace pass's consumer handles it (cbi_trace_pb): it keeps its CEL type for format().
struct TVal {
  enum Kind { Null, Bool, Int, Double, String, List, Map } k = Null;
  bool b = false; long long i = 0; double d = 0; std::string s;
  std::vector<TVal> items;   // List: the elements; Map: key, value, key, value ...
  bool local = false;        // a list the program built in its lane's arena: i elements, logged as CBH_TR_OUTPUT_ELEMENT records
};
// Template of an output expression (celc.py _output_template; cbh_blob.h CBH_SEC_TRACE_HOST)
struct TNode { u8 kind = 0; u32 hole = 0; TVal cst; std::string fmt; std::vector<TNode> kids; };

}  // namespace

struct cbi_table {
  std::vector<u8> blob;
  u32 K = 0;
  const u32* str_off = nullptr;      // table string pool (views into blob)
  const char* str_bytes = nullptr;
  StrIndex ids;                      // table string -> id
  std::unordered_map<std::string, u32> scope_index;    // scope -> index
  std::vector<Column> columns;
  std::vector<std::string> policy_keys, dr_names, scopes;   // response assembly
  // the trace pass (cbi_trace_pb): strings its records refer to, output templates by rule word, the table's constant heap
  std::vector<std::string> trace_strings;
  std::unordered_map<u32, std::pair<TNode, u32>> trace_templates;
  const u8* theap_tag = nullptr; const u64* theap_val = nullptr; u32 theap_len = 0;
  bool has_trace = false, trace_all = false;
  std::string_view at(u32 i) const { return std::string_view(str_bytes + str_off[i], str_off[i + 1] - str_off[i]); }
};

struct cbi_outputs {
  std::vector<u8> bytes, flags;
  std::vector<u64> offsets;
};

struct cbi_batch {
  cbh_batch view{};
  std::vector<u32> req, roles, tuple_req, tuple_action, str_off, req_input;
  std::vector<u8> col_tag, heap_tag, str_bytes, str_flags;
  std::vector<u64> col_val, heap_val, tuple_perm;
  std::vector<u64> str_hash;   // hash_bytes of each batch-local string (merge of slices)
  // Where the strings the response needs sit in each CheckInput (offset, length relative to the message; CheckInput source only):
  // per input IN_SPAN_N pairs - request id, principal id / version, resource kind / version / id - and per input-order tuple
  // its action.  cbi_assemble_pb then reads them instead of walking every message a second time.
  std::vector<u32> in_span, act_span;
};

namespace {

enum { T_NULL = 0, T_BOOL = 1, T_DOUBLE = 4, T_STRING = 5, T_LIST = 6, T_MAP = 7, T_ABSENT = 0xF0, T_ERR = 0xFF };
enum { RQ_PRINCIPAL_ID, RQ_P_SCOPE, RQ_P_VERSION, RQ_KIND, RQ_R_SCOPE, RQ_R_VERSION, RQ_ROLE_OFF, RQ_ROLE_CNT, RQ_ACT_OFF, RQ_ACT_CNT,
       RQ_S_RESOURCE_ID, RQ_S_KIND, RQ_S_P_SCOPE, RQ_S_R_SCOPE, RQ_S_P_VERSION, RQ_S_R_VERSION, RQ_N };
static_assert(RQ_ACT_OFF == CBH_RQ_ACT_OFF && RQ_S_R_VERSION == CBH_RQ_S_R_VERSION && RQ_N == CBH_RQ_NFIELDS, "field order of include/cerbos_hip.h");
constexpr u32 SCOPE_EXACT = 0x80000000u;
constexpr u32 MAX_ACTIONS = 64;
constexpr u32 SF_ACTION = 1, SF_ROLE = 2, SF_KIND = 4;

// namer.go:276-278
std::string_view scope_value(std::string_view s) { return (!s.empty() && s[0] == '.') ? s.substr(1) : s; }

bool name_char(char c) {
  return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_' || c == '@' || c == '.' || c == '-' || c == '/';
}
bool alpha(char c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }

// namer.go:213-218: a name of the pre-0.30 form (segments "[A-Za-z][0-9A-Za-z_@.\-/]*" joined by ':') has
// every run of characters outside [0-9A-Za-z_.] replaced by one '_'; any other name is kept as it is.
std::string_view sanitize(std::string_view v, std::string& out) {
  bool plain = true;
  for (char c : v) if (!((c >= '0' && c <= '9') || alpha(c) || c == '_' || c == '.')) { plain = false; break; }
  if (plain) return v;   // nothing to rewrite whichever form it is
  bool old_form = !v.empty();
  bool seg_start = true;
  for (char c : v) {
    if (seg_start) { if (!alpha(c)) { old_form = false; break; } seg_start = false; }
    else if (c == ':') seg_start = true;
    else if (!name_char(c)) { old_form = false; break; }
  }
  if (seg_start) old_form = false;
  if (!old_form) return v;
  out.clear(); bool in_run = false;
  for (char c : v) {
    bool ok = (c >= '0' && c <= '9') || alpha(c) || c == '_' || c == '.';
    if (ok) { out.push_back(c); in_run = false; }
    else if (!in_run) { out.push_back('_'); in_run = true; }
  }
  return out;
}

struct Interner {
  const cbi_table* t;
  cbi_batch* b;
  StrIndex local;
  // A small direct-mapped cache in front of the two dictionaries: the vocabularies that repeat from message to message - actions,
  // roles, enum-like attribute values - are answered by one probe and one short compare.  Entries point into the caller's message
  // bytes (valid for the whole call); `flags` = the CBH_SF_* bits already recorded for the string.
  struct CacheEnt { const char* p = nullptr; u32 len = 0, id = 0, flags = 0; u64 key = 0; };
  CacheEnt cache[256];
  u32 sid(std::string_view s, u32 flag = 0) {
    u64 w = 0;
    if (!s.empty()) memcpy(&w, s.data(), s.size() < 8 ? s.size() : 8);
    const u64 key = (w ^ ((u64)s.size() << 56)) * 0x9E3779B97F4A7C15ull;
    CacheEnt& e = cache[key >> 56];
    if (e.key == key && e.len == s.size() && e.p && (e.flags & flag) == flag && (s.size() <= 8 || memcmp(e.p, s.data(), s.size()) == 0)) return e.id;
    const u32 id = sid_slow(s, flag);
    if (e.key == key && e.len == s.size() && e.id == id) e.flags |= flag; else e = CacheEnt{s.data(), (u32)s.size(), id, flag, key};
    return id;
  }
  u32 sid_slow(std::string_view s, u32 flag) {
    const u64 h = hash_bytes(s);
    u32 id;
    if (t->ids.find(s, h, [this](u32 i) { return t->at(i); }, id)) return id;
    auto at = [this](u32 i) {
      return std::string_view((const char*)b->str_bytes.data() + b->str_off[i], b->str_off[i + 1] - b->str_off[i]);
    };
    if (local.find(s, h, at, id)) {
      if (flag) b->str_flags[id] |= (u8)flag;
    } else {
      id = (u32)b->str_flags.size();
      b->str_bytes.insert(b->str_bytes.end(), s.begin(), s.end());
      b->str_off.push_back((u32)b->str_bytes.size());
      b->str_flags.push_back((u8)flag);
      b->str_hash.push_back(h);
      local.insert(h, id);
    }
    return t->K + id;
  }
  // Same, through a one-entry memo: request fields that rarely change from one message to the next
  // (versions, scopes, kind) skip the hash.  The views stay valid for the whole call.
  struct Memo { std::string_view s; u32 id = 0; bool set = false; };
  Memo memo[8];
  u32 sid_memo(u32 slot, std::string_view s, u32 flag = 0) {
    Memo& m = memo[slot];
    if (m.set && m.s == s) return m.id;
    m.s = s; m.id = sid(s, flag); m.set = true;
    return m.id;
  }
};

struct TV { u8 tag; u64 val; };

struct Encoder {
  Interner& in;
  cbi_batch* b;
  bool bad = false;
  int depth = 0;

  u64 container(u32 off, size_t n) { return ((u64)1 << 62) | ((u64)off << 32) | (u64)n; }

  // entries of a map<string, Value> field `fnum` of `msg`, children first, then the (key, value) pairs
  TV enc_map(Span msg, u32 fnum) {
    std::vector<TV> ents;
    Field f;
    while (next(msg, f, bad)) {
      if (f.num != fnum || f.wt != 2) continue;
      Entry en;
      if (!entry(f.s, en, bad)) break;
      ents.push_back(TV{(u8)T_STRING, in.sid(sv(en.key))});
      ents.push_back(enc(en.val));
    }
    u32 off = (u32)b->heap_tag.size();
    for (const TV& x : ents) { b->heap_tag.push_back(x.tag); b->heap_val.push_back(x.val); }
    return TV{(u8)T_MAP, container(off, ents.size() / 2)};
  }

  TV enc(Span m) {
    Val v;
    if (!value(m, v, bad)) return TV{(u8)T_NULL, 0};
    switch (v.kind) {
      case 1: return TV{(u8)T_NULL, 0};
      case 2: return TV{(u8)T_DOUBLE, v.v};                 // structpb: every number is a double
      case 3: return TV{(u8)T_STRING, in.sid(sv(v.s))};
      case 4: return TV{(u8)T_BOOL, v.v ? 1u : 0u};
      case 5: {
        if (++depth > 100) { bad = true; return TV{(u8)T_NULL, 0}; }
        TV r = enc_map(v.s, 1);
        --depth;
        return r;
      }
      default: {  // 6: ListValue.values = 1
        if (++depth > 100) { bad = true; return TV{(u8)T_NULL, 0}; }
        std::vector<TV> vals;
        Span l = v.s; Field f;
        while (next(l, f, bad)) if (f.num == 1 && f.wt == 2) vals.push_back(enc(f.s));
        --depth;
        u32 off = (u32)b->heap_tag.size();
        for (const TV& x : vals) { b->heap_tag.push_back(x.tag); b->heap_val.push_back(x.val); }
        return TV{(u8)T_LIST, container(off, vals.size())};
      }
    }
  }
};

struct ScopeCache {   // scope -> scope word, per call (the table stays immutable and shareable between threads)
  StrIndex index;
  std::vector<std::string> keys;
  std::vector<u32> words;
};

u32 scope_word(const cbi_table* t, ScopeCache& sc, std::string_view scope) {
  const u64 h = hash_bytes(scope);
  u32 slot;
  if (sc.index.find(scope, h, [&sc](u32 i) { return std::string_view(sc.keys[i]); }, slot)) return sc.words[slot];
  std::string key(scope);
  u32 w = 0;
  auto it = t->scope_index.find(key);
  if (it != t->scope_index.end()) w = it->second | SCOPE_EXACT;
  else {
    // namer.go:77-87: "a.b.c" -> "a.b", "a", ""
    for (size_t i = scope.size(); i-- > 0;) {
      if (scope[i] == '.' || i == 0) {
        auto p = t->scope_index.find(std::string(scope.substr(0, i)));
        if (p != t->scope_index.end()) { w = p->second; break; }
      }
    }
  }
  sc.index.insert(h, (u32)sc.keys.size());
  sc.keys.push_back(std::move(key));
  sc.words.push_back(w);
  return w;
}

struct Msg {   // the pieces of one CheckInput
  Span principal{nullptr, nullptr}, resource{nullptr, nullptr}, aux{nullptr, nullptr};
};

struct Party { std::string_view id, version, scope, kind; };

}  // namespace

extern "C" {

const char* cbi_last_error(void) { return g_err.c_str(); }

int cbi_table_open(const void* blob, size_t len, cbi_table** out) {
  if (!blob || !out) return fail("cbi_table_open: null argument");
  if (len < sizeof(CbhBlobHeader)) return fail("blob too small");
  auto* t = new cbi_table();
  t->blob.assign((const u8*)blob, (const u8*)blob + len);
  const u8* base = t->blob.data();
  auto* h = (const CbhBlobHeader*)base;
  auto bail = [&](const char* m) { delete t; return fail(m); };
  if (h->magic != CBH_BLOB_MAGIC) return bail("bad blob magic");
  if (h->version != CBH_BLOB_VERSION) return bail("blob version mismatch: re-lower the rule table");
  if (h->total_len != len || sizeof(CbhBlobHeader) + (u64)h->n_sections * sizeof(CbhBlobSection) > len) return bail("blob length mismatch");
  auto* secs = (const CbhBlobSection*)(base + sizeof(CbhBlobHeader));
  auto find = [&](u32 id) -> const CbhBlobSection* {
    for (u32 i = 0; i < h->n_sections; ++i)
      if (secs[i].id == id)   // sections are 64-byte aligned and lie inside the image (no wrap-around)
        return (secs[i].offset % 8 == 0 && secs[i].offset <= len && secs[i].nbytes <= len - secs[i].offset) ? &secs[i] : nullptr;
    return nullptr;
  };
  const CbhBlobSection *so = find(CBH_SEC_STR_OFF), *sb = find(CBH_SEC_STR_BYTES), *ss = find(CBH_SEC_SCOPE_SID), *sc = find(CBH_SEC_COLUMN_PATHS),
                       *sm = find(CBH_SEC_META);
  if (!so || !sb || !ss || !sc || !sm) return bail("blob is missing a section the ingest needs");
  if (sm->nbytes < (u64)CBH_META_N * 4) return bail("META section too short");
  const u32* meta = (const u32*)(base + sm->offset);
  t->K = meta[CBH_M_NSTRINGS];
  if (((u64)t->K + 1) * 4 > so->nbytes) return bail("string offset section too short");
  const u32* off = (const u32*)(base + so->offset);
  const char* bytes = (const char*)(base + sb->offset);
  if (off[t->K] > sb->nbytes) return bail("string byte section too short");
  t->str_off = off; t->str_bytes = bytes;
  for (u32 i = 0; i < t->K; ++i) {
    if (off[i + 1] < off[i] || off[i + 1] > off[t->K]) { delete t; return fail("string offsets are not monotonic"); }
    t->ids.insert(hash_bytes(t->at(i)), i);
  }
  u32 ns = meta[CBH_M_NSCOPES];
  if ((u64)ns * 4 > ss->nbytes) return bail("scope section too short");
  const u32* ssid = (const u32*)(base + ss->offset);
  for (u32 i = 0; i < ns; ++i) {
    if (ssid[i] >= t->K) return bail("scope string id out of range");
    t->scope_index.emplace(std::string(bytes + off[ssid[i]], off[ssid[i] + 1] - off[ssid[i]]), i);
  }
  t->scopes.resize(ns);
  for (u32 i = 0; i < ns; ++i) t->scopes[i] = std::string(t->at(ssid[i]));
  if (const CbhBlobSection* sn = find(CBH_SEC_HOST_NAMES)) {
    const u8* q = base + sn->offset; const u8* qe = q + sn->nbytes;
    for (std::vector<std::string>* dst : {&t->policy_keys, &t->dr_names}) {
      if (qe - q < 4) return bail("host name section truncated");
      u32 cnt; memcpy(&cnt, q, 4); q += 4;
      for (u32 k = 0; k < cnt; ++k) {
        if (qe - q < 2) return bail("host name section truncated");
        u32 l = q[0] | (q[1] << 8); q += 2;
        if ((u32)(qe - q) < l) return bail("host name section truncated");
        dst->emplace_back((const char*)q, l); q += l;
      }
    }
  } else return bail("blob is missing the host name section");
  if (const CbhBlobSection* st = find(CBH_SEC_TRACE_HOST)) {
    const u8* q = base + st->offset; const u8* qe = q + st->nbytes;
    bool ok = true;
    auto rd32 = [&](u32& v) { if (qe - q < 4) { ok = false; v = 0; return; } memcpy(&v, q, 4); q += 4; };
    auto rdstr = [&](std::string& out) { u32 l; rd32(l); if (!ok || (u32)(qe - q) < l) { ok = false; return; } out.assign((const char*)q, l); q += l; };
    u32 ns2; rd32(ns2);
    for (u32 k = 0; ok && k < ns2; ++k) { std::string x; rdstr(x); t->trace_strings.push_back(std::move(x)); }
    std::function<void(TNode&, int)> rdnode = [&](TNode& nd, int depth) {
      if (!ok || depth > 64 || qe - q < 1) { ok = false; return; }
      nd.kind = *q++;
      if (nd.kind == 0) rd32(nd.hole);
      else if (nd.kind == 1) {
        if (qe - q < 1) { ok = false; return; }
        const u8 ct = *q++;
        if (ct == 0) nd.cst.k = TVal::Null;
        else if (ct == 1) { if (qe - q < 1) { ok = false; return; } nd.cst.k = TVal::Bool; nd.cst.b = *q++ != 0; }
        else if (ct == 2) { if (qe - q < 8) { ok = false; return; } nd.cst.k = TVal::Int; memcpy(&nd.cst.i, q, 8); q += 8; }
        else if (ct == 3) { if (qe - q < 8) { ok = false; return; } nd.cst.k = TVal::Double; memcpy(&nd.cst.d, q, 8); q += 8; }
        else if (ct == 4) { nd.cst.k = TVal::String; rdstr(nd.cst.s); }
        else ok = false;
      } else if (nd.kind == 2 || nd.kind == 3 || nd.kind == 4) {
        if (nd.kind == 4) rdstr(nd.fmt);
        u32 cnt; rd32(cnt);
        if (nd.kind == 3) cnt *= 2;
        if (!ok || cnt > (u32)(qe - q)) { ok = false; return; }
        nd.kids.resize(cnt);
        for (u32 k = 0; ok && k < cnt; ++k) rdnode(nd.kids[k], depth + 1);
      } else ok = false;
    };
    u32 nt; rd32(nt);
    for (u32 k = 0; ok && k < nt; ++k) {
      u32 word, holes; rd32(word); rd32(holes);
      TNode nd; rdnode(nd, 0);
      if (ok) t->trace_templates.emplace(word, std::make_pair(std::move(nd), holes));
    }
    if (!ok) return bail("trace host section truncated");
    const CbhBlobSection *ht = find(CBH_SEC_THEAP_TAG), *hv = find(CBH_SEC_THEAP_VAL);
    if (ht && hv && hv->nbytes / 8 >= ht->nbytes) { t->theap_tag = base + ht->offset; t->theap_val = (const u64*)(base + hv->offset); t->theap_len = (u32)ht->nbytes; }
    t->has_trace = true;
    // variables are evaluated whether a condition reads them or not, outputs on every visit of their rule: such a table's
    // inputs are all traced, any other table's only where a decision kernel marked CBH_ST_CEL_ERROR
    const CbhBlobSection* tp = find(CBH_SEC_TRACE_POOL);
    t->trace_all = (tp && tp->count > 0) || !t->trace_templates.empty();
  }
  const u8* p = base + sc->offset; const u8* e = p + sc->nbytes;
  u32 ncol = meta[CBH_M_NCOLUMNS];
  for (u32 c = 0; c < ncol; ++c) {
    if (e - p < 2) return bail("column path section truncated");
    Column col; col.root = p[0]; u32 nk = p[1]; p += 2;
    if (col.root > 4) return bail("bad column root");
    for (u32 k = 0; k < nk; ++k) {
      if (e - p < 2) return bail("column path section truncated");
      u32 l = p[0] | (p[1] << 8); p += 2;
      if ((u32)(e - p) < l) return bail("column path section truncated");
      col.keys.emplace_back((const char*)p, l); p += l;
    }
    t->columns.push_back(std::move(col));
  }
  *out = t;
  return 0;
}

void cbi_table_close(cbi_table* t) { delete t; }
uint32_t cbi_table_trace_scope(const cbi_table* t) { return !t || !t->has_trace ? 0u : t->trace_all ? 2u : 1u; }

}  // extern "C"

// Where the inputs come from: serialized CheckInputs (bytes + offsets), or the resource entries of ONE serialized
// CheckResourcesRequest, which share its principal and the caller's (verified) auxiliary data - the CheckInputs
// svc.CheckResources would build from it (cerbos_svc.go:274-287), without building them.
struct Source {
  const uint8_t* bytes = nullptr; const uint64_t* offsets = nullptr;
  bool request = false; Span principal{nullptr, nullptr}, aux{nullptr, nullptr};
  Span globals{nullptr, nullptr};   // the call's globals as a serialized google.protobuf.Struct (columns of root 4), or empty
  const Span* entries = nullptr;   // CheckResourcesRequest.ResourceEntry messages: actions = 1, resource = 2
};

// Flattens inputs [first, first + n) of `src` into `b` in input order: no routing sort, no view; req_input holds
// slice-local indices.
static int flatten_slice(const cbi_table* t, const Source& src, uint32_t first, uint32_t n, std::string_view dver,
                         std::string_view dscope, cbi_batch* b, std::string& err) {
  const uint8_t* bytes = src.bytes;
  const uint64_t* offsets = src.request ? nullptr : src.offsets + first;
  auto bail = [&](const std::string& m) { err = m; return -1; };
  const u32 ncol = (u32)t->columns.size();

  // pass 1: count device requests (a CheckInput with > 64 actions becomes several) and tuples
  u64 nreq = 0, ntup = 0;
  const u32 action_field = src.request ? 1u : 4u;
  for (u32 i = 0; i < n; ++i) {
    if (!src.request && offsets[i + 1] < offsets[i]) return bail("offsets must not decrease");
    Span m = src.request ? src.entries[first + i] : Span{bytes + offsets[i], bytes + offsets[i + 1]};
    Field f; bool bad = false; size_t na = 0;
    while (next(m, f, bad)) na += (f.num == action_field && f.wt == 2);
    if (bad) return bail("malformed CheckInput at index " + std::to_string(i));
    nreq += na ? (na + MAX_ACTIONS - 1) / MAX_ACTIONS : 1;
    ntup += na;
  }
  if (nreq > 0xFFFFFFFFull || ntup > 0xFFFFFFFFull) return bail("batch too large");
  const u32 R = (u32)nreq;
  b->req.assign((size_t)RQ_N * R, 0);
  b->col_tag.assign((size_t)ncol * R, (u8)T_ABSENT);
  b->col_val.assign((size_t)ncol * R, 0);
  b->req_input.reserve(R);
  b->tuple_req.reserve(ntup); b->tuple_action.reserve(ntup);
  b->str_off.reserve((size_t)n * 2 + 64); b->str_flags.reserve((size_t)n * 2 + 64); b->str_hash.reserve((size_t)n * 2 + 64);
  b->roles.reserve((size_t)n * 3);
  if (!src.request) { b->in_span.resize((size_t)n * 2 * IN_SPAN_N); b->act_span.resize((size_t)ntup * 2); }
  size_t act_at = 0;   // next slot of act_span
  b->str_off.push_back(0);
  b->str_bytes.reserve((size_t)n * 24 + (1 << 12));
  Interner in{t, b, {}, {}};
  in.local.reserve(n);
  Encoder encd{in, b};
  ScopeCache scopes;
  Interner::Memo scope_memos[2];
  auto RQ = [&](u32 f, u32 r) -> u32& { return b->req[(size_t)f * R + r]; };

  // per message scratch, reused
  std::vector<std::string_view> actions, roles;
  std::vector<std::pair<std::string_view, u32>> prev_roles, prev_actions;
  std::vector<Entry> attrs[5];   // entries of Principal.attr / Resource.attr / AuxData.jwt, wire order; [4]: the call's globals (once)
  std::string kind_buf;
  bool need_root[5] = {false, false, false, false, false};
  for (const Column& c : t->columns) need_root[c.root] = true;
  if (need_root[4]) {   // EvalParams.Globals (evaluator.go:52-57): Struct.fields = 1
    Span s = src.globals; Field f; bool bad = false;
    while (next(s, f, bad)) if (f.num == 1 && f.wt == 2) { Entry en; if (entry(f.s, en, bad)) attrs[4].push_back(en); }
    if (bad) return bail("malformed globals");
  }

  u32 r = 0;
  for (u32 i = 0; i < n; ++i) {
    Msg m;
    std::string_view request_id;
    actions.clear(); roles.clear();
    for (int a = 0; a < 3; ++a) attrs[a].clear();
    bool bad = false;
    if (src.request) {
      m.principal = src.principal; m.aux = src.aux;
      Span s = src.entries[first + i]; Field f;
      while (next(s, f, bad)) { if (f.wt != 2) continue; if (f.num == 1) actions.push_back(sv(f.s)); else if (f.num == 2) m.resource = f.s; }
    } else {
      Span s{bytes + offsets[i], bytes + offsets[i + 1]}; Field f;
      while (next(s, f, bad)) { if (f.wt != 2) continue;
        if (f.num == 2) m.resource = f.s; else if (f.num == 3) m.principal = f.s; else if (f.num == 4) actions.push_back(sv(f.s)); else if (f.num == 5) m.aux = f.s;
        else if (f.num == 1) request_id = sv(f.s); }
    }
    // Principal: id 1, policy_version 2, roles 3, attr 4, scope 5;  Resource: kind 1, policy_version 2, id 3, attr 4, scope 5
    Party P, Rs;
    { Span s = m.principal; Field f;
      while (next(s, f, bad)) { if (f.wt != 2) continue;
        if (f.num == 1) P.id = sv(f.s); else if (f.num == 2) P.version = sv(f.s); else if (f.num == 3) roles.push_back(sv(f.s));
        else if (f.num == 5) P.scope = sv(f.s);
        else if (f.num == 4 && need_root[0]) { Entry en; if (entry(f.s, en, bad)) attrs[0].push_back(en); } } }
    { Span s = m.resource; Field f;
      while (next(s, f, bad)) { if (f.wt != 2) continue;
        if (f.num == 1) Rs.kind = sv(f.s); else if (f.num == 2) Rs.version = sv(f.s); else if (f.num == 3) Rs.id = sv(f.s);
        else if (f.num == 5) Rs.scope = sv(f.s);
        else if (f.num == 4 && need_root[1]) { Entry en; if (entry(f.s, en, bad)) attrs[1].push_back(en); } } }
    if (need_root[2]) { Span s = m.aux; Field f;
      while (next(s, f, bad)) if (f.num == 1 && f.wt == 2) { Entry en; if (entry(f.s, en, bad)) attrs[2].push_back(en); } }
    if (bad) return bail("malformed CheckInput at index " + std::to_string(i));
    if (!src.request) {
      const char* base0 = (const char*)(bytes + offsets[i]);
      u32* sp = b->in_span.data() + (size_t)i * 2 * IN_SPAN_N;
      auto span = [&](u32 which, std::string_view v) { sp[2 * which] = v.empty() ? 0u : (u32)(v.data() - base0); sp[2 * which + 1] = (u32)v.size(); };
      span(SPAN_REQUEST_ID, request_id); span(SPAN_P_ID, P.id); span(SPAN_P_VERSION, P.version); span(SPAN_R_KIND, Rs.kind);
      span(SPAN_R_VERSION, Rs.version); span(SPAN_R_ID, Rs.id);
      u32* ap = b->act_span.data() + act_at;
      for (std::string_view a : actions) { *ap++ = a.empty() ? 0u : (u32)(a.data() - base0); *ap++ = (u32)a.size(); }
      act_at += 2 * actions.size();
    }
    const size_t na = actions.size();
    const size_t nchunks = na ? (na + MAX_ACTIONS - 1) / MAX_ACTIONS : 1;
    for (size_t ch = 0; ch < nchunks; ++ch, ++r) {
      b->req_input.push_back(i);
      std::string_view p_scope = scope_value(P.scope.empty() ? dscope : P.scope);
      std::string_view r_scope = scope_value(Rs.scope.empty() ? dscope : Rs.scope);
      std::string_view p_ver = P.version.empty() ? dver : P.version;
      std::string_view r_ver = Rs.version.empty() ? dver : Rs.version;
      RQ(RQ_PRINCIPAL_ID, r) = in.sid(P.id);
      auto scope_memo = [&](u32 slot, std::string_view sc) {   // scopes rarely change from one message to the next
        Interner::Memo& mk = scope_memos[slot];
        if (!(mk.set && mk.s == sc)) { mk.s = sc; mk.id = scope_word(t, scopes, sc); mk.set = true; }
        return mk.id;
      };
      RQ(RQ_P_SCOPE, r) = scope_memo(0, p_scope);
      RQ(RQ_P_VERSION, r) = in.sid_memo(0, p_ver);
      { Interner::Memo& mk = in.memo[1];   // keyed by the raw kind, holds the id of the sanitised one
        if (!(mk.set && mk.s == Rs.kind)) { mk.s = Rs.kind; const std::string_view sk = sanitize(Rs.kind, kind_buf);
          // a rewritten kind lives in the reused scratch string, not in the message bytes: it must never enter the cache
          // in front of the dictionaries (whose entries point at what they were made from)
          mk.id = sk.data() == Rs.kind.data() ? in.sid(sk, SF_KIND) : in.sid_slow(sk, SF_KIND); mk.set = true; }
        RQ(RQ_KIND, r) = mk.id; }
      RQ(RQ_R_SCOPE, r) = scope_memo(1, r_scope);
      RQ(RQ_R_VERSION, r) = in.sid_memo(2, r_ver);
      RQ(RQ_ROLE_OFF, r) = (u32)b->roles.size();
      RQ(RQ_ROLE_CNT, r) = (u32)roles.size();
      // role and action lists repeat from one message to the next: try the previous message's string at the same
      // position before hashing (the id already carries the role / action flag from when it was interned that way)
      for (size_t x = 0; x < roles.size(); ++x) {
        if (x >= prev_roles.size()) prev_roles.emplace_back(std::string_view(), 0u);
        auto& pr = prev_roles[x];
        if (!(pr.first.data() && pr.first == roles[x])) { pr.first = roles[x]; pr.second = in.sid(roles[x], SF_ROLE); }
        b->roles.push_back(pr.second);
      }
      RQ(RQ_S_RESOURCE_ID, r) = in.sid(Rs.id);
      RQ(RQ_S_KIND, r) = in.sid_memo(3, Rs.kind);
      RQ(RQ_S_P_SCOPE, r) = in.sid_memo(4, scope_value(P.scope));
      RQ(RQ_S_R_SCOPE, r) = in.sid_memo(5, scope_value(Rs.scope));
      RQ(RQ_S_P_VERSION, r) = in.sid_memo(6, P.version);
      RQ(RQ_S_R_VERSION, r) = in.sid_memo(7, Rs.version);
      for (u32 c = 0; c < ncol; ++c) {
        const Column& col = t->columns[c];
        u8 tag = 0; u64 val = 0; bool done = false;
        const size_t nk = col.keys.size();
        Span cur{nullptr, nullptr};
        size_t k = 0;   // keys consumed so far; `cur` is a google.protobuf.Value once k > 0
        if (col.root == 3) {
          // AuxData.jwts (field 2): map<string, JWT>, JWT.claims (field 1): map<string, Value>.  The request view
          // is name -> {"claims": {...}} (check.go:536-554), so the 2nd key of a path must be "claims".
          auto jwt_as_map = [&](Span jwt) {   // {"claims": {...}}: the key is interned before the claims, children first
            const u64 key = in.sid("claims");
            const TV inner = encd.enc_map(jwt, 1);
            const u32 off = (u32)b->heap_tag.size();
            b->heap_tag.push_back((u8)T_STRING); b->heap_val.push_back(key);
            b->heap_tag.push_back(inner.tag); b->heap_val.push_back(inner.val);
            return TV{(u8)T_MAP, encd.container(off, 1)};
          };
          Span jwt{nullptr, nullptr};
          if (nk == 0) {
            std::vector<TV> ents;
            Span s = m.aux; Field f;
            while (next(s, f, encd.bad)) {
              if (f.num != 2 || f.wt != 2) continue;
              Entry en;
              if (!entry(f.s, en, encd.bad)) break;
              ents.push_back(TV{(u8)T_STRING, in.sid(sv(en.key))});
              ents.push_back(jwt_as_map(en.val));
            }
            const u32 off = (u32)b->heap_tag.size();
            for (const TV& x : ents) { b->heap_tag.push_back(x.tag); b->heap_val.push_back(x.val); }
            tag = (u8)T_MAP; val = encd.container(off, ents.size() / 2); done = true;
          } else if (!map_get(m.aux, 2, col.keys[0], jwt, encd.bad)) { tag = nk == 1 ? (u8)T_ABSENT : (u8)T_ERR; done = true; }
          else if (nk == 1) { TV tv = jwt_as_map(jwt); tag = tv.tag; val = tv.val; done = true; }
          else if (col.keys[1] != "claims") { tag = nk == 2 ? (u8)T_ABSENT : (u8)T_ERR; done = true; }
          else if (nk == 2) { TV tv = encd.enc_map(jwt, 1); tag = tv.tag; val = tv.val; done = true; }
          else if (!map_get(jwt, 1, col.keys[2], cur, encd.bad)) { tag = nk == 3 ? (u8)T_ABSENT : (u8)T_ERR; done = true; }
          else k = 3;
        } else if (nk == 0) {
          // the whole root map: Principal.attr (4) / Resource.attr (4) / AuxData.jwt (1)
          TV tv = encd.enc_map(col.root == 0 ? m.principal : col.root == 1 ? m.resource : col.root == 4 ? src.globals : m.aux, (col.root == 2 || col.root == 4) ? 1 : 4);
          tag = tv.tag; val = tv.val; done = true;
        } else {
          bool found = false;
          for (const Entry& en : attrs[col.root]) if (sv(en.key) == col.keys[0]) { cur = en.val; found = true; }   // last entry wins
          if (!found) { tag = nk == 1 ? (u8)T_ABSENT : (u8)T_ERR; done = true; }
          k = 1;
        }
        for (; k < nk && !done; ++k) {
          Val v;
          if (!value(cur, v, encd.bad) || v.kind != 5) { tag = (u8)T_ERR; done = true; break; }
          if (!map_get(v.s, 1, col.keys[k], cur, encd.bad)) { tag = (k == nk - 1) ? (u8)T_ABSENT : (u8)T_ERR; done = true; }
        }
        if (!done) { TV tv = encd.enc(cur); tag = tv.tag; val = tv.val; }
        b->col_tag[(size_t)c * R + r] = tag;
        b->col_val[(size_t)c * R + r] = val;
      }
      if (encd.bad) return bail("malformed attribute value at index " + std::to_string(i));
      RQ(RQ_ACT_OFF, r) = (u32)b->tuple_action.size();
      size_t a0 = ch * MAX_ACTIONS, a1 = std::min(na, a0 + MAX_ACTIONS);
      RQ(RQ_ACT_CNT, r) = (u32)(a1 - a0);
      for (size_t a = a0; a < a1; ++a) {
        if (a >= prev_actions.size()) prev_actions.emplace_back(std::string_view(), 0u);
        auto& pa = prev_actions[a];
        if (!(pa.first.data() && pa.first == actions[a])) { pa.first = actions[a]; pa.second = in.sid(actions[a], SF_ACTION); }
        b->tuple_req.push_back(r); b->tuple_action.push_back(pa.second);
      }
    }
  }
  return 0;
}

// Routing sort (flatten.py sort_batch_by_route): kind, resource version, resource scope, role count,
// order-sensitive signature of the role list; stable.  Fills tuple_perm.
static void sort_batch(cbi_batch* b, u32 ncol, bool sort, int n_threads) {
  const u32 R = (u32)b->req_input.size();
  auto RQ = [&](u32 f, u32 r) -> u32& { return b->req[(size_t)f * R + r]; };
  const u32 T = (u32)b->tuple_action.size();
  b->tuple_perm.resize(T);
  std::iota(b->tuple_perm.begin(), b->tuple_perm.end(), (u64)0);
  if (!sort || R < 2) return;
  // Requests fall into few routing groups (distinct kind / version / scope / role list): find the groups with a
  // hash table, order the groups, then place the requests with one stable counting pass - O(R + G log G)
  // instead of a comparison sort of all requests.
  struct Key { u32 kind, ver, scope, cnt; u64 sig; };
  auto key_eq = [](const Key& x, const Key& y) { return x.kind == y.kind && x.ver == y.ver && x.scope == y.scope && x.cnt == y.cnt && x.sig == y.sig; };
  auto key_less = [](const Key& x, const Key& y) {
    if (x.kind != y.kind) return x.kind < y.kind;
    if (x.ver != y.ver) return x.ver < y.ver;
    if (x.scope != y.scope) return x.scope < y.scope;
    if (x.cnt != y.cnt) return x.cnt < y.cnt;
    return x.sig < y.sig;
  };
  std::vector<Key> groups;
  std::vector<u32> group_of(R);
  {
    size_t cap = 256;
    std::vector<u32> slots(cap, 0);   // group index + 1
    auto hash = [](const Key& k) {
      u64 h = (u64)k.kind * 0x9E3779B97F4A7C15ull ^ (u64)k.ver * 0xC2B2AE3D27D4EB4Full ^ (u64)k.scope * 0x165667B19E3779F9ull ^ k.sig ^ ((u64)k.cnt << 40);
      h ^= h >> 29; h *= 0xff51afd7ed558ccdull; h ^= h >> 32;
      return (size_t)h;
    };
    for (u32 q = 0; q < R; ++q) {
      const u32 off = RQ(RQ_ROLE_OFF, q), cnt = RQ(RQ_ROLE_CNT, q);
      u64 sg = 0;
      for (u32 k = 0; k < cnt; ++k) sg += ((u64)b->roles[off + k] + 1) * ((u64)k * 0x9E3779B97F4A7C15ull + 0xC2B2AE3D27D4EB4Full);
      const Key key{RQ(RQ_KIND, q), RQ(RQ_R_VERSION, q), RQ(RQ_R_SCOPE, q), cnt, sg};
      if ((groups.size() + 1) * 2 > cap) {   // grow and re-place
        cap *= 4;
        slots.assign(cap, 0);
        for (u32 g = 0; g < groups.size(); ++g) { size_t i = hash(groups[g]) & (cap - 1); while (slots[i]) i = (i + 1) & (cap - 1); slots[i] = g + 1; }
      }
      size_t i = hash(key) & (cap - 1);
      while (slots[i] && !key_eq(groups[slots[i] - 1], key)) i = (i + 1) & (cap - 1);
      if (!slots[i]) { groups.push_back(key); slots[i] = (u32)groups.size(); }
      group_of[q] = slots[i] - 1;
    }
  }
  const u32 G = (u32)groups.size();
  std::vector<u32> by_rank(G);
  std::iota(by_rank.begin(), by_rank.end(), 0u);
  std::sort(by_rank.begin(), by_rank.end(), [&](u32 x, u32 y) { return key_less(groups[x], groups[y]); });
  std::vector<u32> rank(G), start(G + 1, 0);
  for (u32 r = 0; r < G; ++r) rank[by_rank[r]] = r;
  for (u32 q = 0; q < R; ++q) ++start[rank[group_of[q]] + 1];
  for (u32 r = 0; r < G; ++r) start[r + 1] += start[r];
  std::vector<u32> order(R);
  bool identity = true;
  for (u32 q = 0; q < R; ++q) { const u32 pos = start[rank[group_of[q]]]++; order[pos] = q; identity = identity && pos == q; }
  if (identity) return;
  // gather array by array (sequential writes, reads confined to one array at a time); the arrays are independent
  // jobs and go round-robin to the threads of a parallel call
  std::vector<u32> req2((size_t)RQ_N * R), ta(T), tr(T), ri(R), new_off(R + 1, 0);
  std::vector<u8> ct((size_t)ncol * R);
  std::vector<u64> cv((size_t)ncol * R), tp(T);
  const u32* ord = order.data();
  for (u32 q = 0; q < R; ++q) new_off[q + 1] = new_off[q] + RQ(RQ_ACT_CNT, ord[q]);
  const u32 n_jobs = RQ_N + ncol + 1;
  auto job = [&](u32 j) {
    if (j < RQ_N) {
      const u32* src = &b->req[(size_t)j * R]; u32* dst = &req2[(size_t)j * R];
      if (j == RQ_ACT_OFF) { for (u32 q = 0; q < R; ++q) dst[q] = new_off[q]; }
      else for (u32 q = 0; q < R; ++q) dst[q] = src[ord[q]];
    } else if (j < RQ_N + ncol) {
      const u32 c = j - RQ_N;
      const u8* st = &b->col_tag[(size_t)c * R]; u8* dt = &ct[(size_t)c * R];
      const u64* sv_ = &b->col_val[(size_t)c * R]; u64* dv = &cv[(size_t)c * R];
      for (u32 q = 0; q < R; ++q) { dt[q] = st[ord[q]]; dv[q] = sv_[ord[q]]; }
    } else {
      for (u32 q = 0; q < R; ++q) {
        const u32 o = ord[q], s0 = RQ(RQ_ACT_OFF, o), cn = RQ(RQ_ACT_CNT, o);
        u32 pos = new_off[q];
        for (u32 k = 0; k < cn; ++k, ++pos) { ta[pos] = b->tuple_action[s0 + k]; tr[pos] = q; tp[pos] = s0 + k; }
        ri[q] = b->req_input[o];
      }
    }
  };
  const u32 W = n_threads > 1 ? std::min<u32>((u32)n_threads, n_jobs) : 1u;
  if (W == 1) { for (u32 j = 0; j < n_jobs; ++j) job(j); }
  else {
    std::vector<std::thread> th;
    for (u32 w = 0; w < W; ++w) th.emplace_back([&, w]() { for (u32 j = w; j < n_jobs; j += W) job(j); });
    for (auto& x : th) x.join();
  }
  b->req.swap(req2); b->col_tag.swap(ct); b->col_val.swap(cv);
  b->tuple_action.swap(ta); b->tuple_req.swap(tr); b->tuple_perm.swap(tp); b->req_input.swap(ri);
}

static void finish_view(cbi_batch* b, u32 ncol) {
  // never hand out null pointers for empty arrays
  auto nz32 = [](std::vector<u32>& v) { if (v.empty()) v.reserve(1); return v.data(); };
  cbh_batch& v = b->view;
  v.n_requests = (u32)b->req_input.size(); v.n_tuples = (u32)b->tuple_action.size(); v.n_roles = (u32)b->roles.size(); v.n_columns = ncol;
  v.n_strings = (u32)b->str_flags.size(); v.heap_len = (u32)b->heap_tag.size(); v.str_bytes_len = b->str_bytes.size();
  b->heap_tag.reserve(1); b->heap_val.reserve(1); b->str_bytes.reserve(1); b->str_flags.reserve(1);
  b->col_tag.reserve(1); b->col_val.reserve(1); b->tuple_perm.reserve(1); b->req_input.reserve(1);
  v.req_u32 = nz32(b->req); v.roles = nz32(b->roles); v.tuple_req = nz32(b->tuple_req); v.tuple_action = nz32(b->tuple_action);
  v.col_tag = b->col_tag.data(); v.col_val = b->col_val.data(); v.heap_tag = b->heap_tag.data(); v.heap_val = b->heap_val.data();
  v.str_off = b->str_off.data(); v.str_bytes = b->str_bytes.data(); v.str_flags = b->str_flags.data();
}

// Concatenates slice batches (slice k = inputs [base[k], base[k+1])) into one.  A batch-local string keeps
// the id it would have had in a single pass: the merged dictionary takes slice 0's strings, then the
// strings slice 1 saw first, ... - which is first-appearance order over the whole input.  Heap tapes
// and role lists concatenate; only offsets and batch-local ids are rebased.
static void merge_slices(const cbi_table* t, std::vector<cbi_batch*>& parts, const std::vector<u32>& base, cbi_batch* out, int n_threads) {
  const u32 ncol = (u32)t->columns.size(), K = t->K, P = (u32)parts.size();
  std::vector<u32> rb(P + 1, 0), tb(P + 1, 0), hb(P + 1, 0), lb(P + 1, 0);   // request / tuple / heap / role bases
  for (u32 k = 0; k < P; ++k) {
    rb[k + 1] = rb[k] + (u32)parts[k]->req_input.size(); tb[k + 1] = tb[k] + (u32)parts[k]->tuple_action.size();
    hb[k + 1] = hb[k] + (u32)parts[k]->heap_tag.size(); lb[k + 1] = lb[k] + (u32)parts[k]->roles.size();
  }
  const u32 R = rb[P], T = tb[P];
  // Merged dictionary.  The id of a string is its rank in first-appearance order over (slice, position).
  // Finding each string's first appearance is the expensive part (hashing, comparing) and splits by hash:
  // partition p looks only at strings with hash % B == p, so partitions never share a string.  What is left
  // for the serial pass is handing out ids in order and copying the bytes.
  std::vector<std::vector<u32>> remap(P);
  std::vector<std::vector<u64>> owner(P);           // (slice << 32 | position) of the first appearance
  std::vector<std::vector<u8>> oflags(P);           // flags OR-ed over all appearances, kept at the owner
  for (u32 k = 0; k < P; ++k) { const size_t ns = parts[k]->str_flags.size(); remap[k].resize(ns); owner[k].resize(ns); oflags[k] = parts[k]->str_flags; }
  auto str_of = [&](u64 ref) {
    const cbi_batch* p = parts[ref >> 32]; const u32 j = (u32)ref;
    return std::string_view((const char*)p->str_bytes.data() + p->str_off[j], p->str_off[j + 1] - p->str_off[j]);
  };
  const u32 B = n_threads > 1 ? (u32)n_threads : 1u;
  auto dedupe = [&](u32 part) {
    size_t mine = 0;
    for (u32 k = 0; k < P; ++k) for (u64 h : parts[k]->str_hash) mine += ((h >> 32) % B) == part;
    size_t cap = 64; while (cap < 2 * mine) cap <<= 1;
    struct Slot { u32 h; u32 used; u64 ref; };
    std::vector<Slot> slots(cap, Slot{0, 0, 0});
    for (u32 k = 0; k < P; ++k) {
      const std::vector<u64>& hs = parts[k]->str_hash;
      for (u32 j = 0; j < hs.size(); ++j) {
        const u64 h = hs[j];
        if (((h >> 32) % B) != part) continue;
        const u64 ref = ((u64)k << 32) | j;
        size_t i = (size_t)h & (cap - 1);
        for (;; i = (i + 1) & (cap - 1)) {
          Slot& sl = slots[i];
          if (!sl.used) { sl = Slot{(u32)h, 1, ref}; owner[k][j] = ref; break; }
          if (sl.h == (u32)h && str_of(sl.ref) == str_of(ref)) {
            owner[k][j] = sl.ref;
            oflags[sl.ref >> 32][(u32)sl.ref] |= parts[k]->str_flags[j];
            break;
          }
        }
      }
    }
  };
  if (B == 1) dedupe(0);
  else {
    std::vector<std::thread> th;
    for (u32 part = 0; part < B; ++part) th.emplace_back(dedupe, part);
    for (auto& x : th) x.join();
  }
  out->str_off.assign(1, 0);
  { size_t bytes = 0, cnt = 0; for (cbi_batch* p : parts) { bytes += p->str_bytes.size(); cnt += p->str_flags.size(); }
    out->str_bytes.reserve(bytes); out->str_off.reserve(cnt + 1); out->str_flags.reserve(cnt); out->str_hash.reserve(cnt); }
  for (u32 k = 0; k < P; ++k) {
    const cbi_batch* p = parts[k];
    for (u32 j = 0; j < p->str_flags.size(); ++j) {
      const u64 ow = owner[k][j];
      if (ow == (((u64)k << 32) | j)) {
        remap[k][j] = (u32)out->str_flags.size();
        out->str_bytes.insert(out->str_bytes.end(), p->str_bytes.begin() + p->str_off[j], p->str_bytes.begin() + p->str_off[j + 1]);
        out->str_off.push_back((u32)out->str_bytes.size());
        out->str_flags.push_back(oflags[k][j]);
        out->str_hash.push_back(p->str_hash[j]);
      } else {
        remap[k][j] = remap[ow >> 32][(u32)ow];   // the owner comes earlier in (slice, position) order
      }
    }
  }
  for (u32 k = 0; k < P; ++k) {   // slices hold consecutive inputs: their spans follow each other
    out->in_span.insert(out->in_span.end(), parts[k]->in_span.begin(), parts[k]->in_span.end());
    out->act_span.insert(out->act_span.end(), parts[k]->act_span.begin(), parts[k]->act_span.end());
  }
  out->req.assign((size_t)RQ_N * R, 0);
  out->col_tag.assign((size_t)ncol * R, 0); out->col_val.assign((size_t)ncol * R, 0);
  out->roles.resize(lb[P]); out->tuple_req.resize(T); out->tuple_action.resize(T);
  out->heap_tag.resize(hb[P]); out->heap_val.resize(hb[P]); out->req_input.resize(R);
  static const u32 STRING_FIELDS[] = {RQ_PRINCIPAL_ID, RQ_P_VERSION, RQ_KIND, RQ_R_VERSION, RQ_S_RESOURCE_ID, RQ_S_KIND,
                                      RQ_S_P_SCOPE, RQ_S_R_SCOPE, RQ_S_P_VERSION, RQ_S_R_VERSION};
  auto place = [&](u32 k) {   // slices write disjoint ranges: one thread each
    const cbi_batch* p = parts[k];
    const std::vector<u32>& mp = remap[k];
    const u32 r0 = rb[k], nr = rb[k + 1] - rb[k], h0 = hb[k];
    auto sid = [&](u32 id) { return id >= K ? K + mp[id - K] : id; };
    auto val = [&](u8 tag, u64 v) -> u64 {
      if (tag == T_STRING) return sid((u32)v);
      if (tag == T_LIST || tag == T_MAP) return v + ((u64)h0 << 32);   // (HEAP_BATCH << 62) | off << 32 | len
      return v;
    };
    for (u32 f = 0; f < RQ_N; ++f) std::memcpy(&out->req[(size_t)f * R + r0], &p->req[(size_t)f * nr], (size_t)nr * 4);
    for (u32 f : STRING_FIELDS) for (u32 r = 0; r < nr; ++r) { u32& x = out->req[(size_t)f * R + r0 + r]; x = sid(x); }
    for (u32 r = 0; r < nr; ++r) {
      out->req[(size_t)RQ_ROLE_OFF * R + r0 + r] += lb[k]; out->req[(size_t)RQ_ACT_OFF * R + r0 + r] += tb[k];
      out->req_input[r0 + r] = p->req_input[r] + base[k];
    }
    for (size_t i = 0; i < p->roles.size(); ++i) out->roles[lb[k] + i] = sid(p->roles[i]);
    for (size_t i = 0; i < p->tuple_action.size(); ++i) { out->tuple_action[tb[k] + i] = sid(p->tuple_action[i]); out->tuple_req[tb[k] + i] = p->tuple_req[i] + r0; }
    for (u32 c = 0; c < ncol; ++c)
      for (u32 r = 0; r < nr; ++r) {
        const u8 tag = p->col_tag[(size_t)c * nr + r];
        out->col_tag[(size_t)c * R + r0 + r] = tag;
        out->col_val[(size_t)c * R + r0 + r] = val(tag, p->col_val[(size_t)c * nr + r]);
      }
    for (size_t i = 0; i < p->heap_tag.size(); ++i) { out->heap_tag[h0 + i] = p->heap_tag[i]; out->heap_val[h0 + i] = val(p->heap_tag[i], p->heap_val[i]); }
  };
  if (n_threads <= 1) { for (u32 k = 0; k < P; ++k) place(k); }
  else {
    std::vector<std::thread> th;
    for (u32 k = 0; k < P; ++k) th.emplace_back(place, k);
    for (auto& x : th) x.join();
  }
}

static int flatten_source(const cbi_table* t, const Source& src, uint32_t n, const char* default_version, const char* default_scope,
                          int sort, int n_threads, cbi_batch** out) {
  const std::string_view dver = default_version ? default_version : "default";
  const std::string_view dscope = default_scope ? default_scope : "";
  const u32 ncol = (u32)t->columns.size();
  u32 P = n_threads > 1 ? (u32)n_threads : 1u;
  if (P > 64) P = 64;
  if (n < 1024u * P) P = n / 1024u ? n / 1024u : 1u;   // not worth a thread below ~1k messages
  auto b = new cbi_batch();
  std::string err;
  if (P == 1) {
    if (flatten_slice(t, src, 0, n, dver, dscope, b, err) != 0) { delete b; return fail(err); }
  } else {
    std::vector<u32> base(P + 1);
    for (u32 k = 0; k <= P; ++k) base[k] = (u32)((u64)n * k / P);
    std::vector<cbi_batch*> parts(P);
    std::vector<std::string> errs(P);
    std::vector<int> rcs(P, 0);
    for (u32 k = 0; k < P; ++k) parts[k] = new cbi_batch();
    {
      std::vector<std::thread> th;
      for (u32 k = 0; k < P; ++k)
        th.emplace_back([&, k]() { rcs[k] = flatten_slice(t, src, base[k], base[k + 1] - base[k], dver, dscope, parts[k], errs[k]); });
      for (auto& x : th) x.join();
    }
    int bad = -1;
    for (u32 k = 0; k < P && bad < 0; ++k) if (rcs[k] != 0) bad = (int)k;
    if (bad < 0) merge_slices(t, parts, base, b, (int)P);
    for (cbi_batch* p : parts) delete p;
    if (bad >= 0) { delete b; return fail(errs[bad] + " (slice starting at message " + std::to_string(base[bad]) + ")"); }
  }
  if (b->heap_tag.size() >= ((size_t)1 << 30)) { delete b; return fail("batch too large: nested attribute values exceed the heap's 30-bit offsets"); }
  sort_batch(b, ncol, sort != 0, (int)P);
  finish_view(b, ncol);
  *out = b;
  return 0;
}

// The resource entries (field 4) of a CheckResourcesRequest, its principal (3), request id (1), include_meta (2).
// Scalars: the last occurrence wins.  The principal and an entry's resource are MESSAGE fields: protobuf merges their occurrences
// (what proto.Unmarshal did to the request the server validated and logged), and parsing the concatenation of their bytes is that
// merge - so a request that is not canonical in this way is read through a canonical copy (`owned`: the occurrences back to back),
// as the device road's split does (cbh_wire_req.h).
struct RequestParts {
  std::vector<Span> entries; Span principal{nullptr, nullptr}; std::string_view request_id; bool include_meta = false;
  std::vector<std::vector<u8>> owned;   // (an inner vector's bytes stay where they are when the outer one grows)
};
static void append_varint(std::vector<u8>& o, u64 v) { while (v >= 0x80) { o.push_back((u8)(v | 0x80)); v >>= 7; } o.push_back((u8)v); }
static bool split_request(const uint8_t* request, uint64_t len, RequestParts& rp) {
  Span s{request, request + len}; Field f; bool bad = false;
  u32 n_principal = 0;
  while (next(s, f, bad)) {
    if (f.num == 2 && f.wt == 0) rp.include_meta = f.v != 0;
    if (f.wt != 2) continue;
    if (f.num == 1) rp.request_id = sv(f.s); else if (f.num == 3) { rp.principal = f.s; ++n_principal; } else if (f.num == 4) rp.entries.push_back(f.s);
  }
  if (bad) return false;
  if (n_principal > 1) {
    std::vector<u8> all;
    s = Span{request, request + len};
    while (next(s, f, bad)) if (f.wt == 2 && f.num == 3) all.insert(all.end(), f.s.p, f.s.e);
    rp.owned.push_back(std::move(all));
    rp.principal = Span{rp.owned.back().data(), rp.owned.back().data() + rp.owned.back().size()};
  }
  for (Span& e : rp.entries) {
    u32 n_res = 0;
    { Span t = e; while (next(t, f, bad)) n_res += (f.wt == 2 && f.num == 2); }
    if (bad) return false;
    if (n_res <= 1) continue;
    std::vector<u8> canon, res;   // the entry's other fields as they are, then ONE resource made of all its occurrences
    Span t = e;
    for (;;) {
      const u8* start = t.p;
      if (!next(t, f, bad)) break;
      if (f.wt == 2 && f.num == 2) res.insert(res.end(), f.s.p, f.s.e); else canon.insert(canon.end(), start, t.p);
    }
    canon.push_back((u8)(2u << 3 | 2u)); append_varint(canon, res.size()); canon.insert(canon.end(), res.begin(), res.end());
    rp.owned.push_back(std::move(canon));
    e = Span{rp.owned.back().data(), rp.owned.back().data() + rp.owned.back().size()};
  }
  return !bad;
}

extern "C" {

int cbi_flatten_pb_g(const cbi_table* t, const uint8_t* bytes, const uint64_t* offsets, uint32_t n, const char* default_version,
                     const char* default_scope, const uint8_t* globals_pb, uint64_t globals_len, int sort, int n_threads, cbi_batch** out) {
  if (!t || !out || (n && (!bytes || !offsets)) || (globals_len && !globals_pb)) return fail("cbi_flatten_pb: null argument");
  Source src; src.bytes = bytes; src.offsets = offsets;
  if (globals_len) src.globals = Span{globals_pb, globals_pb + globals_len};
  return flatten_source(t, src, n, default_version, default_scope, sort, n_threads, out);
}

int cbi_flatten_pb_mt(const cbi_table* t, const uint8_t* bytes, const uint64_t* offsets, uint32_t n, const char* default_version,
                      const char* default_scope, int sort, int n_threads, cbi_batch** out) {
  return cbi_flatten_pb_g(t, bytes, offsets, n, default_version, default_scope, nullptr, 0, sort, n_threads, out);
}

int cbi_flatten_request_pb(const cbi_table* t, const uint8_t* request, uint64_t request_len, const uint8_t* aux_data, uint64_t aux_len,
                           const char* default_version, const char* default_scope, int sort, int n_threads, cbi_batch** out) {
  if (!t || !out || !request) return fail("cbi_flatten_request_pb: null argument");
  RequestParts rp;
  if (!split_request(request, request_len, rp)) return fail("malformed CheckResourcesRequest");
  if (rp.entries.size() > 0xFFFFFFFFull) return fail("batch too large");
  Source src; src.request = true; src.principal = rp.principal; src.entries = rp.entries.data();
  if (aux_data) src.aux = Span{aux_data, aux_data + aux_len};
  return flatten_source(t, src, (u32)rp.entries.size(), default_version, default_scope, sort, n_threads, out);
}

int cbi_flatten_pb(const cbi_table* t, const uint8_t* bytes, const uint64_t* offsets, uint32_t n, const char* default_version,
                   const char* default_scope, int sort, cbi_batch** out) {
  return cbi_flatten_pb_mt(t, bytes, offsets, n, default_version, default_scope, sort, 1, out);
}

}  // extern "C"

// ---- response assembly ---------------------------------------------------------------------------------
// Policy key of a device policy word (enum cbh_policy_kind), as namer.PolicyKeyFromFQN gives it (namer.go:95-134).
// Returns an error text, or nullptr.
static const char* policy_key(const cbi_table* t, u32 word, const Party& P, const Party& Rs, std::string_view dver,
                              std::string& pol, std::string& kbuf, std::string& vbuf) {
  const u32 kind = word >> 28, ident = word & 0x0FFFFFFFu;
  pol.clear();
  switch (kind) {
    case CBH_P_EMPTY: return nullptr;
    case CBH_P_NO_MATCH: pol = "NO_MATCH"; return nullptr;
    case CBH_P_NO_MATCH_SCOPE_PERMISSIONS: pol = "NO_MATCH_FOR_SCOPE_PERMISSIONS"; return nullptr;
    case CBH_P_TABLE:
      if (ident >= t->policy_keys.size()) return "policy id out of range";
      pol = t->policy_keys[ident];
      return nullptr;
    case CBH_P_RESOURCE: case CBH_P_PRINCIPAL: {
      if (ident >= t->scopes.size()) return "scope index out of range";
      const bool rp = kind == CBH_P_RESOURCE;
      std::string_view ver = rp ? Rs.version : P.version;
      pol = rp ? "resource." : "principal.";
      pol += sanitize(rp ? Rs.kind : P.id, kbuf); pol += ".v"; pol += sanitize(ver.empty() ? dver : ver, vbuf);
      if (!t->scopes[ident].empty()) { pol += '/'; pol += t->scopes[ident]; }
      return nullptr;
    }
    default: return "unknown policy word";
  }
}

static void put_varint(std::vector<u8>& o, u64 v) { while (v >= 0x80) { o.push_back((u8)(v | 0x80)); v >>= 7; } o.push_back((u8)v); }
static size_t varint_size(u64 v) { size_t n = 1; while (v >= 0x80) { v >>= 7; ++n; } return n; }
static void put_ld(std::vector<u8>& o, u32 field, std::string_view s) {
  put_varint(o, (u64)field << 3 | 2); put_varint(o, s.size()); o.insert(o.end(), s.begin(), s.end());
}
static void put_str(std::vector<u8>& o, u32 field, std::string_view s) { if (!s.empty()) put_ld(o, field, s); }   // proto3 default: omitted

// The CheckOutputs of inputs 0 .. n - 1.  inv: input-order tuple k -> the tuple of `res` that answers it (null: k itself);
// edr: effective derived roles per input; first[i]: input-order index of input i's first tuple; in_span / act_span: where the
// strings sit in the messages (null: every message is walked again).
static int assemble_outputs(const cbi_table* t, const cbh_result* res, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                            std::string_view dver, int n_threads, const u32* inv, const u64* edr, const u64* first,
                            const u32* in_span, const u32* act_span, cbi_outputs** out) {
  // inputs [lo, hi) -> o (offsets relative to the part)
  auto assemble_range = [&](u32 lo, u32 hi, cbi_outputs* o, std::string& err) -> int {
    auto bail = [&](const std::string& m) { err = m; return -1; };
    o->offsets.reserve(hi - lo + 1); o->offsets.push_back(0); o->flags.assign(hi - lo, 0);
    o->bytes.reserve((size_t)(hi - lo) * 96);
    struct Act { std::string_view name; u32 j; };
    std::vector<Act> acts;
    std::string pol, kbuf, vbuf;
    auto o_flags = [&](u32 x) -> u8& { return o->flags[x]; };
    u64 k = first[lo];
    const u64 k_hi = first[hi];
    const bool spans = in_span != nullptr && act_span != nullptr;
    for (u32 i = lo; i < hi; ++i) {
      Span m{bytes + offsets[i], bytes + offsets[i + 1]};
      Span principal{nullptr, nullptr}, resource{nullptr, nullptr};
      std::string_view request_id;
      acts.clear();
      bool bad = false; Field f;
      Party P, Rs;
      if (spans) {
        // the flattener noted where these strings sit in the message (cbi_batch::in_span): no second walk
        const size_t mlen = (size_t)(m.e - m.p);
        const u32* sp = in_span + (size_t)i * 2 * IN_SPAN_N;
        bool ok = true;
        auto at = [&](u32 which) { const u32 o = sp[2 * which], l = sp[2 * which + 1]; if ((size_t)o + l > mlen) { ok = false; return std::string_view(); } return std::string_view((const char*)m.p + o, l); };
        request_id = at(SPAN_REQUEST_ID); P.id = at(SPAN_P_ID); P.version = at(SPAN_P_VERSION); Rs.kind = at(SPAN_R_KIND); Rs.version = at(SPAN_R_VERSION); Rs.id = at(SPAN_R_ID);
        const u64 k_end = first[i + 1];
        for (; k < k_end; ++k) {
          const u32 o = act_span[2 * k], l = act_span[2 * k + 1];
          if ((size_t)o + l > mlen) { ok = false; break; }
          std::string_view name((const char*)m.p + o, l); const u32 j = inv ? inv[k] : (u32)k; bool dup = false;
          for (Act& a : acts) if (a.name == name) {
            if (res->effect[j] == CBH_EFFECT_DENY || res->effect[a.j] != CBH_EFFECT_DENY) a.j = j;
            dup = true; break;
          }
          if (!dup) acts.push_back(Act{name, j});
          if (res->status) { u8 st = res->status[j]; if (st == CBH_ST_UNSUPPORTED) o_flags(i - lo) |= CBI_OUT_UNSUPPORTED; else if (st == CBH_ST_CEL_ERROR) o_flags(i - lo) |= CBI_OUT_CEL_ERROR; else if (st == CBH_ST_WANTS_TRACE) o_flags(i - lo) |= CBI_OUT_WANTS_TRACE; }
        }
        if (!ok) return bail("batch does not belong to these inputs");
      } else {
      while (next(m, f, bad)) { if (f.wt != 2) continue;
        if (f.num == 1) request_id = sv(f.s); else if (f.num == 2) resource = f.s; else if (f.num == 3) principal = f.s;
        else if (f.num == 4) {
          if (k >= k_hi) return bail("batch does not belong to these inputs");
          // setEffect (check.go:513-530): a later duplicate replaces an earlier one unless that one is a DENY and it is not
          std::string_view name = sv(f.s); u32 j = inv ? inv[k] : (u32)k; ++k; bool dup = false;
          for (Act& a : acts) if (a.name == name) {
            if (res->effect[j] == CBH_EFFECT_DENY || res->effect[a.j] != CBH_EFFECT_DENY) a.j = j;
            dup = true; break;
          }
          if (!dup) acts.push_back(Act{name, j});
          if (res->status) { u8 st = res->status[j]; if (st == CBH_ST_UNSUPPORTED) o->flags[i - lo] |= CBI_OUT_UNSUPPORTED; else if (st == CBH_ST_CEL_ERROR) o->flags[i - lo] |= CBI_OUT_CEL_ERROR; else if (st == CBH_ST_WANTS_TRACE) o->flags[i - lo] |= CBI_OUT_WANTS_TRACE; }
        } }
      { Span s = principal; while (next(s, f, bad)) { if (f.wt != 2) continue; if (f.num == 1) P.id = sv(f.s); else if (f.num == 2) P.version = sv(f.s); } }
      { Span s = resource; while (next(s, f, bad)) { if (f.wt != 2) continue; if (f.num == 1) Rs.kind = sv(f.s); else if (f.num == 2) Rs.version = sv(f.s); else if (f.num == 3) Rs.id = sv(f.s); } }
      if (bad) return bail("malformed CheckInput at index " + std::to_string(i));
      }
      std::vector<u8>& ob = o->bytes;
      // One reservation for everything but the derived-role names, then plain pointer writes: the bound is the sum of the
      // strings plus a fixed allowance per field (two bytes of tag + at most five of a length); policy keys come from
      // policy_key (checked against its allowance below).
      size_t bound = 32 + request_id.size() + Rs.id.size();
      for (const Act& a : acts) bound += a.name.size() + 64;
      const size_t at0 = ob.size();
      size_t cap = at0 + bound + acts.size() * 256;   // + room for a policy key and a scope per action
      if (ob.size() < cap) ob.resize(cap);
      u8* w = ob.data() + at0;
      auto wv = [&](u64 v) { while (v >= 0x80) { *w++ = (u8)(v | 0x80); v >>= 7; } *w++ = (u8)v; };
      auto wld = [&](u32 field, std::string_view v) { wv((u64)field << 3 | 2); wv(v.size()); if (!v.empty()) { memcpy(w, v.data(), v.size()); w += v.size(); } };
      auto wstr = [&](u32 field, std::string_view v) { if (!v.empty()) wld(field, v); };
      wstr(1, request_id);
      wstr(2, Rs.id);
      u32 pol_word = 0xFFFFFFFFu;   // `pol` holds the key of this word (the actions of one input mostly share it)
      for (const Act& a : acts) {
        if (res->policy && res->policy[a.j] != pol_word) {
          pol_word = res->policy[a.j];
          if (const char* e = policy_key(t, pol_word, P, Rs, dver, pol, kbuf, vbuf)) return bail(e);
        }
        std::string_view scope_s;
        if (res->scope && res->scope[a.j] != 0xFFFFFFFFu) {
          if (res->scope[a.j] >= t->scopes.size()) return bail("scope index out of range");
          scope_s = t->scopes[res->scope[a.j]];
        }
        // sizes first, then one pass of writes: entry {1: action, 2: ActionEffect {1: effect, 2: policy, 3: scope}}
        const std::string_view pol_s = res->policy ? std::string_view(pol) : std::string_view();
        if (pol_s.size() + scope_s.size() > 224) {   // longer than the allowance: grow (keys this long are unusual)
          const size_t used = (size_t)(w - ob.data());
          ob.resize(ob.size() + pol_s.size() + scope_s.size());
          w = ob.data() + used;
        }
        const u8 effect = res->effect[a.j];
        auto ld_size = [](size_t n) { return 1 + varint_size(n) + n; };
        const size_t eff_len = (effect ? 1 + varint_size(effect) : 0) + (pol_s.empty() ? 0 : ld_size(pol_s.size())) +
                               (scope_s.empty() ? 0 : ld_size(scope_s.size()));
        const size_t ent_len = ld_size(a.name.size()) + ld_size(eff_len);
        wv(3u << 3 | 2); wv(ent_len);
        wld(1, a.name);
        wv(2u << 3 | 2); wv(eff_len);
        if (effect) { *w++ = 1 << 3 | 0; wv(effect); }
        wstr(2, pol_s);
        wstr(3, scope_s);
      }
      ob.resize((size_t)(w - ob.data()));
      for (u32 d = 0; d < 64 && d < t->dr_names.size(); ++d) if ((edr[i] >> d) & 1) put_ld(ob, 4, t->dr_names[d]);
      o->offsets.push_back(ob.size());
    }
    if (k != k_hi) return bail("batch does not belong to these inputs");
    return 0;
  };

  u32 P = n_threads > 1 ? (u32)n_threads : 1u;
  if (P > 64) P = 64;
  if (n < 1024u * P) P = n / 1024u ? n / 1024u : 1u;
  auto o = new cbi_outputs();
  std::string err;
  if (P == 1) {
    if (assemble_range(0, n, o, err) != 0) { delete o; return fail(err); }
  } else {
    std::vector<cbi_outputs> parts(P);
    std::vector<std::string> errs(P);
    std::vector<int> rcs(P, 0);
    std::vector<u32> base(P + 1);
    for (u32 k = 0; k <= P; ++k) base[k] = (u32)((u64)n * k / P);
    {
      std::vector<std::thread> th;
      for (u32 k = 0; k < P; ++k) th.emplace_back([&, k]() { rcs[k] = assemble_range(base[k], base[k + 1], &parts[k], errs[k]); });
      for (auto& x : th) x.join();
    }
    for (u32 k = 0; k < P; ++k) if (rcs[k] != 0) { delete o; return fail(errs[k]); }
    size_t total = 0;
    for (const cbi_outputs& p : parts) total += p.bytes.size();
    o->bytes.resize(total); o->flags.resize(n); o->offsets.resize((size_t)n + 1);
    size_t at = 0;
    for (u32 k = 0; k < P; ++k) {
      const cbi_outputs& p = parts[k];
      if (!p.bytes.empty()) std::memcpy(o->bytes.data() + at, p.bytes.data(), p.bytes.size());
      if (!p.flags.empty()) std::memcpy(o->flags.data() + base[k], p.flags.data(), p.flags.size());
      for (u32 i = base[k]; i <= base[k + 1]; ++i) o->offsets[i] = at + p.offsets[i - base[k]];
      at += p.bytes.size();
    }
  }
  o->bytes.reserve(1);
  *out = o;
  return 0;
}

extern "C" {

int cbi_assemble_pb_mt(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint8_t* bytes, const uint64_t* offsets,
                       uint32_t n, const char* default_version, int n_threads, cbi_outputs** out) {
  if (!t || !b || !res || !res->effect || !out || (n && (!bytes || !offsets))) return fail("cbi_assemble_pb: null argument");
  std::string_view dver = default_version ? default_version : "default";
  const u32 T = b->view.n_tuples, R = b->view.n_requests;
  // input-order tuple k lives at device tuple inv[k]; derived roles of an input = OR over its device requests
  std::vector<u32> inv(T);
  for (u32 j = 0; j < T; ++j) { if (b->tuple_perm[j] >= T) return fail("corrupt tuple permutation"); inv[b->tuple_perm[j]] = j; }
  std::vector<u64> edr(n, 0), first(n + 1, 0);   // first[i] = input-order index of input i's first tuple
  for (u32 q = 0; q < R; ++q) {
    const u32 i = b->req_input[q];
    if (i >= n) return fail("batch does not belong to these inputs");
    if (res->edr_mask) edr[i] |= res->edr_mask[q];
    first[i + 1] += b->req[(size_t)RQ_ACT_CNT * R + q];
  }
  for (u32 i = 0; i < n; ++i) first[i + 1] += first[i];
  if (first[n] != T) return fail("batch does not belong to these inputs");

  return assemble_outputs(t, res, bytes, offsets, n, dver, n_threads, inv.data(), edr.data(), first.data(),
                          b->in_span.size() == (size_t)n * 2 * IN_SPAN_N && b->act_span.size() == (size_t)T * 2 ? b->in_span.data() : nullptr,
                          b->act_span.data(), out);
}

// Results of a batch the DEVICE flattened (cerbos_hip.h cbh_wire_flatten: tuples in input order, one request per input) ->
// serialized CheckOutputs; in_span / act_span / act_off as cbh_wire_spans_download returns them.
int cbi_assemble_wire_pb(const cbi_table* t, const cbh_result* res, const uint8_t* bytes, const uint64_t* offsets, uint32_t n,
                         uint32_t n_tuples, const uint32_t* in_span, const uint32_t* act_span, const uint32_t* act_off,
                         const char* default_version, int n_threads, cbi_outputs** out) {
  if (!t || !res || !res->effect || !out || (n && (!bytes || !offsets || !in_span || !act_off)) || (n_tuples && !act_span)) return fail("cbi_assemble_wire_pb: null argument");
  std::string_view dver = default_version ? default_version : "default";
  std::vector<u64> first((size_t)n + 1, 0), edr(res->edr_mask ? 0 : n, 0);
  for (u32 i = 0; i <= n; ++i) {
    first[i] = n ? act_off[i] : 0;
    if (first[i] > n_tuples || (i && first[i] < first[i - 1])) return fail("cbi_assemble_wire_pb: action offsets are not a partition of the tuples");
  }
  if (first[n] != n_tuples) return fail("cbi_assemble_wire_pb: action offsets are not a partition of the tuples");
  return assemble_outputs(t, res, bytes, offsets, n, dver, n_threads, nullptr, res->edr_mask ? res->edr_mask : edr.data(), first.data(), in_span, act_span, out);
}

int cbi_assemble_pb(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint8_t* bytes, const uint64_t* offsets,
                    uint32_t n, const char* default_version, cbi_outputs** out) {
  return cbi_assemble_pb_mt(t, b, res, bytes, offsets, n, default_version, 1, out);
}

int cbi_assemble_response_pb(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint8_t* request, uint64_t request_len,
                             const char* default_version, cbi_outputs** out) {
  return cbi_assemble_response_traced_pb(t, b, res, request, request_len, default_version, nullptr, out);
}

int cbi_assemble_response_traced_pb(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint8_t* request, uint64_t request_len,
                                    const char* default_version, const cbi_outputs* traced, cbi_outputs** out) {
  if (!t || !b || !res || !res->effect || !out || !request) return fail("cbi_assemble_response_pb: null argument");
  const std::string_view dver = default_version ? default_version : "default";
  RequestParts rp;
  if (!split_request(request, request_len, rp)) return fail("malformed CheckResourcesRequest");
  const u32 n = (u32)rp.entries.size(), T = b->view.n_tuples, R = b->view.n_requests;
  std::vector<u32> inv(T);
  for (u32 j = 0; j < T; ++j) { if (b->tuple_perm[j] >= T) return fail("corrupt tuple permutation"); inv[b->tuple_perm[j]] = j; }
  std::vector<u64> edr(n, 0);
  for (u32 q = 0; q < R; ++q) {
    if (b->req_input[q] >= n) return fail("batch does not belong to this request");
    if (res->edr_mask) edr[b->req_input[q]] |= res->edr_mask[q];
  }
  auto o = new cbi_outputs();
  auto bail = [&](const std::string& m) { delete o; return fail(m); };
  o->flags.assign(n, 0);
  std::vector<u8>& ob = o->bytes;
  ob.reserve((size_t)n * 96 + 64);
  // CheckResourcesResponse (response.proto:187-300): request_id = 1, results = 2 {resource = 1 {id, kind, policy_version,
  // scope}, actions = 2 map<string, Effect>, meta = 4 {actions = 1 map<string, {matched_policy = 1, matched_scope = 2}>,
  // effective_derived_roles = 2}} - what CheckResources assembles from the outputs (cerbos_svc.go:297-343)
  put_str(ob, 1, rp.request_id);
  Party P;
  { Span s = rp.principal; Field f; bool bad = false;
    while (next(s, f, bad)) { if (f.wt != 2) continue; if (f.num == 1) P.id = sv(f.s); else if (f.num == 2) P.version = sv(f.s); }
    if (bad) return bail("malformed principal"); }
  struct Act { std::string_view name; u32 j; };
  std::vector<Act> acts;
  std::vector<u8> entry_buf, meta_buf, tmp;
  std::string pol, kbuf, vbuf;
  u64 k = 0;
  auto ld_size = [](size_t len) { return 1 + varint_size(len) + len; };
  for (u32 i = 0; i < n; ++i) {
    Span e = rp.entries[i]; Field f; bool bad = false;
    Span resource{nullptr, nullptr};
    acts.clear();
    while (next(e, f, bad)) { if (f.wt != 2) continue;
      if (f.num == 2) resource = f.s;
      else if (f.num == 1) {
        if (k >= T) return bail("batch does not belong to this request");
        std::string_view name = sv(f.s); const u32 j = inv[k++]; bool dup = false;
        for (Act& a : acts) if (a.name == name) { if (res->effect[j] == CBH_EFFECT_DENY || res->effect[a.j] != CBH_EFFECT_DENY) a.j = j; dup = true; break; }
        if (!dup) acts.push_back(Act{name, j});
        if (res->status) { const u8 st = res->status[j]; if (st == CBH_ST_UNSUPPORTED) o->flags[i] |= CBI_OUT_UNSUPPORTED; else if (st == CBH_ST_CEL_ERROR) o->flags[i] |= CBI_OUT_CEL_ERROR; else if (st == CBH_ST_WANTS_TRACE) o->flags[i] |= CBI_OUT_WANTS_TRACE; }
      } }
    Party Rs;
    { Span s = resource; while (next(s, f, bad)) { if (f.wt != 2) continue;
        if (f.num == 1) Rs.kind = sv(f.s); else if (f.num == 2) Rs.version = sv(f.s); else if (f.num == 3) Rs.id = sv(f.s); else if (f.num == 5) Rs.scope = sv(f.s); } }
    if (bad) return bail("malformed resource entry " + std::to_string(i));
    entry_buf.clear(); meta_buf.clear();
    tmp.clear();
    put_str(tmp, 1, Rs.id); put_str(tmp, 2, Rs.kind); put_str(tmp, 3, Rs.version); put_str(tmp, 4, Rs.scope);
    put_ld(entry_buf, 1, std::string_view((const char*)tmp.data(), tmp.size()));
    u32 pol_word = 0xFFFFFFFFu;
    for (const Act& a : acts) {
      const u8 effect = res->effect[a.j];
      // actions map entry {1: name, 2: effect (enum varint, left out when 0)}
      put_varint(entry_buf, 2u << 3 | 2); put_varint(entry_buf, ld_size(a.name.size()) + (effect ? 1 + varint_size(effect) : 0));
      put_ld(entry_buf, 1, a.name);
      if (effect) { entry_buf.push_back(2 << 3 | 0); put_varint(entry_buf, effect); }
      if (rp.include_meta) {
        if (res->policy && res->policy[a.j] != pol_word) {
          pol_word = res->policy[a.j];
          if (const char* err = policy_key(t, pol_word, P, Rs, dver, pol, kbuf, vbuf)) return bail(err);
        }
        std::string_view scope_s;
        if (res->scope && res->scope[a.j] != 0xFFFFFFFFu) {
          if (res->scope[a.j] >= t->scopes.size()) return bail("scope index out of range");
          scope_s = t->scopes[res->scope[a.j]];
        }
        const std::string_view pol_s = res->policy ? std::string_view(pol) : std::string_view();
        const size_t em_len = (pol_s.empty() ? 0 : ld_size(pol_s.size())) + (scope_s.empty() ? 0 : ld_size(scope_s.size()));
        put_varint(meta_buf, 1u << 3 | 2); put_varint(meta_buf, ld_size(a.name.size()) + ld_size(em_len));
        put_ld(meta_buf, 1, a.name);
        put_varint(meta_buf, 2u << 3 | 2); put_varint(meta_buf, em_len);
        put_str(meta_buf, 1, pol_s); put_str(meta_buf, 2, scope_s);
      }
    }
    if (rp.include_meta) {
      for (u32 d = 0; d < 64 && d < t->dr_names.size(); ++d) if ((edr[i] >> d) & 1) put_ld(meta_buf, 2, t->dr_names[d]);
      put_ld(entry_buf, 4, std::string_view((const char*)meta_buf.data(), meta_buf.size()));   // present (possibly empty) when asked for
    }
    if (traced) {   // ResultEntry.outputs = 5 <- CheckOutput.outputs = 6 of the trace consumer's bytes for this entry (cerbos_svc.go:325-327)
      if (traced->offsets.size() != (size_t)n + 1) return bail("traced outputs do not belong to this request");
      Span ts{traced->bytes.data() + traced->offsets[i], traced->bytes.data() + traced->offsets[i + 1]}; Field tf; bool tbad = false;
      while (next(ts, tf, tbad)) if (tf.num == 6 && tf.wt == 2) put_ld(entry_buf, 5, sv(tf.s));
      o->flags[i] |= traced->flags[i] & (CBI_TRACE_ERRORS_INCOMPLETE | CBI_TRACE_OUTPUTS_INCOMPLETE);
    }
    put_ld(ob, 2, std::string_view((const char*)entry_buf.data(), entry_buf.size()));
  }
  if (k != T) return bail("batch does not belong to this request");
  o->offsets.assign({0, (u64)ob.size()});
  ob.reserve(1);
  *out = o;
  return 0;
}

}  // extern "C"

// ---- the trace pass's consumer: cbh_trace records -> the evaluation_errors / outputs fields of a CheckOutput ------------
// (cerbos_amd/trace.py is the same consumer in Python; nothing here evaluates CEL)
namespace {
struct TraceIncomplete {};

// %s of a value (cel-go strings.go formatString and the per-type formatters)
void format_value(const TVal& v, bool nested, std::string& out) {
  switch (v.k) {
    case TVal::Null: out += "null"; return;
    case TVal::Bool: out += v.b ? "true" : "false"; return;
    case TVal::String: if (nested) { out += '"'; out += v.s; out += '"'; } else out += v.s; return;
    case TVal::Int: out += std::to_string(v.i); return;
    case TVal::Double: {
      if (std::isnan(v.d) || std::isinf(v.d)) throw TraceIncomplete{};
      if (v.d == std::floor(v.d) && std::fabs(v.d) < 1e21) {
        if (std::fabs(v.d) < 9e18) { out += std::to_string((long long)v.d); return; }
        char buf[64]; std::snprintf(buf, sizeof buf, "%.0f", v.d); out += buf; return;
      }
      char buf[64]; auto r = std::to_chars(buf, buf + sizeof buf, v.d); out.append(buf, r.ptr); return;   // shortest round-trip, as Python's repr
    }
    case TVal::List: {
      out += '[';
      for (size_t i = 0; i < v.items.size(); ++i) { if (i) out += ", "; format_value(v.items[i], true, out); }
      out += ']'; return;
    }
    case TVal::Map: {
      std::vector<std::pair<std::string, size_t>> order;   // by the keys' text
      for (size_t i = 0; i + 1 < v.items.size(); i += 2) { std::string k; format_value(v.items[i], false, k); order.emplace_back(std::move(k), i); }
      std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
      out += '{';
      for (size_t n = 0; n < order.size(); ++n) {
        if (n) out += ", ";
        format_value(v.items[order[n].second], true, out); out += ": "; format_value(v.items[order[n].second + 1], true, out);
      }
      out += '}'; return;
    }
  }
}
// cel-go ext.Strings format: %s, %d, %% (anything else is left to the caller's engine)
TVal format_call(const std::string& fmt, const std::vector<TVal>& args) {
  TVal r; r.k = TVal::String;
  size_t ai = 0;
  for (size_t i = 0; i < fmt.size();) {
    const char c = fmt[i++];
    if (c != '%') { r.s += c; continue; }
    if (i >= fmt.size()) throw TraceIncomplete{};
    const char spec = fmt[i++];
    if (spec == '%') { r.s += '%'; continue; }
    if (ai >= args.size() || (spec != 's' && spec != 'd')) throw TraceIncomplete{};
    const TVal& a = args[ai++];
    if (spec == 'd') {
      if (a.k == TVal::Int) r.s += std::to_string(a.i);
      else if (a.k == TVal::Double && a.d == std::floor(a.d) && std::fabs(a.d) < 9e18) r.s += std::to_string((long long)a.d);
      else throw TraceIncomplete{};
    } else format_value(a, false, r.s);
  }
  return r;
}
bool tval_key_equal(const TVal& a, const TVal& b) {
  if (a.k != b.k) return false;
  switch (a.k) { case TVal::Bool: return a.b == b.b; case TVal::Int: return a.i == b.i; case TVal::Double: return a.d == b.d; case TVal::String: return a.s == b.s; default: return false; }
}
TVal assemble_template(const TNode& t, const std::vector<TVal>& holes) {
  switch (t.kind) {
    case 0: if (t.hole >= holes.size()) throw TraceIncomplete{}; return holes[t.hole];
    case 1: return t.cst;
    case 2: { TVal r; r.k = TVal::List; for (const TNode& k : t.kids) r.items.push_back(assemble_template(k, holes)); return r; }
    case 3: {
      TVal r; r.k = TVal::Map;
      for (size_t i = 0; i + 1 < t.kids.size(); i += 2) {
        TVal key = assemble_template(t.kids[i], holes);
        if (key.k == TVal::Null || key.k == TVal::List || key.k == TVal::Map) throw TraceIncomplete{};   // "unsupported key type"
        for (size_t j = 0; j < r.items.size(); j += 2) if (tval_key_equal(r.items[j], key)) throw TraceIncomplete{};   // a repeated key
        r.items.push_back(std::move(key)); r.items.push_back(assemble_template(t.kids[i + 1], holes));
      }
      return r;
    }
    default: { std::vector<TVal> args; for (const TNode& k : t.kids) args.push_back(assemble_template(k, holes)); return format_call(t.fmt, args); }
  }
}
// google.protobuf.Value (struct.proto): null 1, number 2, string 3, bool 4, struct 5 {fields 1: map<string, Value>}, list 6 {values 1}
void put_value(std::vector<u8>& o, const TVal& v) {
  auto ld = [&](u32 field, const std::vector<u8>& b) { put_varint(o, field << 3 | 2); put_varint(o, b.size()); o.insert(o.end(), b.begin(), b.end()); };
  switch (v.k) {
    case TVal::Null: o.push_back(1 << 3 | 0); o.push_back(0); return;
    case TVal::Bool: o.push_back(4 << 3 | 0); o.push_back(v.b ? 1 : 0); return;
    case TVal::Int: case TVal::Double: { const double d = v.k == TVal::Int ? (double)v.i : v.d; o.push_back(2 << 3 | 1); u8 b[8]; memcpy(b, &d, 8); o.insert(o.end(), b, b + 8); return; }
    case TVal::String: put_ld(o, 3, v.s); return;
    case TVal::List: {
      std::vector<u8> lst;
      for (const TVal& e : v.items) { std::vector<u8> eb; put_value(eb, e); put_varint(lst, 1 << 3 | 2); put_varint(lst, eb.size()); lst.insert(lst.end(), eb.begin(), eb.end()); }
      ld(6, lst); return;
    }
    case TVal::Map: {
      std::vector<u8> st;
      for (size_t i = 0; i + 1 < v.items.size(); i += 2) {
        std::string key; format_value(v.items[i], false, key);
        std::vector<u8> vb, ent; put_value(vb, v.items[i + 1]);
        put_ld(ent, 1, key); put_varint(ent, 2 << 3 | 2); put_varint(ent, vb.size()); ent.insert(ent.end(), vb.begin(), vb.end());
        put_varint(st, 1 << 3 | 2); put_varint(st, ent.size()); st.insert(st.end(), ent.begin(), ent.end());
      }
      ld(5, st); return;
    }
  }
}
}  // namespace

namespace {
// what the consumer reads of input i: the messages its attribute paths resolve in, and its action names in order
struct InputView { Span principal{nullptr, nullptr}, resource{nullptr, nullptr}, aux{nullptr, nullptr}; std::vector<std::string_view> actions; };
}  // namespace

static int trace_decode(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint32_t* records, uint32_t count, uint32_t n,
                        const std::function<void(u32, InputView&)>& view_of, cbi_outputs** out) {
  if (!t->has_trace) return fail("the table was lowered without the trace sections");
  const u32 T = b->view.n_tuples, R = b->view.n_requests;
  auto RQ = [&](u32 f, u32 q) { return b->req[(size_t)f * R + q]; };
  std::vector<u64> first(n + 1, 0);
  for (u32 q = 0; q < R; ++q) { const u32 i = b->req_input[q]; if (i >= n) return fail("batch does not belong to these inputs"); first[i + 1] += RQ(RQ_ACT_CNT, q); }
  for (u32 i = 0; i < n; ++i) first[i + 1] += first[i];
  if (first[n] != T) return fail("batch does not belong to these inputs");
  const u32 K = t->K;
  auto str = [&](u64 id) -> std::string_view {
    if (id < K) return t->at((u32)id);
    const u64 l = id - K;
    if (l >= b->str_flags.size()) throw TraceIncomplete{};
    return std::string_view((const char*)b->str_bytes.data() + b->str_off[l], b->str_off[l + 1] - b->str_off[l]);
  };
  std::function<TVal(u32, u64, int)> to_tval = [&](u32 tag, u64 v, int depth) -> TVal {
    TVal r;
    if (depth > 64) throw TraceIncomplete{};
    switch (tag) {
      case 0: return r;
      case 1: r.k = TVal::Bool; r.b = v != 0; return r;
      case 2: r.k = TVal::Int; r.i = (long long)v; return r;
      case 3: if (v >> 63) { r.k = TVal::Double; r.d = (double)v; } else { r.k = TVal::Int; r.i = (long long)v; } return r;
      case 4: r.k = TVal::Double; memcpy(&r.d, &v, 8); return r;
      case 5: r.k = TVal::String; r.s = std::string(str(v & 0xFFFFFFFFu)); return r;
      case 6: case 7: {
        const u32 sel = (u32)(v >> 62), off = (u32)((v >> 32) & 0x3FFFFFFFu), len = (u32)v;
        r.k = tag == 6 ? TVal::List : TVal::Map;
        if (sel == CBH_HEAP_ROLES) { if ((u64)off + len > b->roles.size()) throw TraceIncomplete{}; for (u32 i = 0; i < len; ++i) { TVal e; e.k = TVal::String; e.s = std::string(str(b->roles[off + i])); r.items.push_back(std::move(e)); } return r; }
        if (sel == CBH_HEAP_LOCAL && tag == 6 && depth == 0) { r.local = true; r.i = len; return r; }
        const u8* tags; const u64* vals; u64 cap;
        if (sel == CBH_HEAP_BATCH) { tags = b->heap_tag.data(); vals = b->heap_val.data(); cap = b->heap_tag.size(); }
        else if (sel == CBH_HEAP_TABLE) { tags = t->theap_tag; vals = t->theap_val; cap = t->theap_len; }
        else throw TraceIncomplete{};
        const u64 cnt = tag == 6 ? (u64)len : 2ull * len;
        if ((u64)off + cnt > cap) throw TraceIncomplete{};
        for (u64 i = 0; i < cnt; ++i) r.items.push_back(to_tval(tags[off + i], vals[off + i], depth + 1));
        return r;
      }
      case 11: {   // CBH_T_EDRSET: runtime.effectiveDerivedRoles as a value - the names of the mask's bits, sorted (check.go:593-610)
        r.k = TVal::List;
        std::vector<std::string> names;
        for (u32 d = 0; d < 64 && d < t->dr_names.size(); ++d) if ((v >> d) & 1) names.push_back(t->dr_names[d]);
        std::sort(names.begin(), names.end());
        for (auto& nm : names) { TVal e; e.k = TVal::String; e.s = std::move(nm); r.items.push_back(std::move(e)); }
        return r;
      }
      default: throw TraceIncomplete{};   // timestamps / durations as output values
    }
  };
  // the text of an error (include/cerbos_hip.h CBH_ERR_*) for input `iv`
  auto message = [&](u64 payload, const InputView& iv) -> std::string {
    const u32 code = (u32)(payload & 0xFF); const u64 detail = payload >> 8;
    switch (code) {
      case CBH_ERR_NO_SUCH_OVERLOAD: return "no such overload";
      case CBH_ERR_DIV_BY_ZERO: return "division by zero";
      case CBH_ERR_MOD_BY_ZERO: return "modulus by zero";
      case CBH_ERR_INT_OVERFLOW: return "integer overflow";
      case CBH_ERR_UINT_OVERFLOW: return "unsigned integer overflow";
      case CBH_ERR_NO_SUCH_KEY: return "no such key: " + std::string(str(detail & 0xFFFFFFFFu));
      case CBH_ERR_UNDEFINED_FIELD: if (detail >= t->trace_strings.size()) throw TraceIncomplete{}; return "undefined field '" + t->trace_strings[detail] + "'";
      case CBH_ERR_EDR_FAILED: {
        std::vector<std::string> names;
        for (u32 d = 0; d < 56 && d < t->dr_names.size(); ++d) if ((detail >> d) & 1) names.push_back(t->dr_names[d]);
        std::sort(names.begin(), names.end());
        std::string m = "failed to compute effective derived roles [";
        for (size_t i = 0; i < names.size(); ++i) { if (i) m += ", "; m += names[i]; }
        return m + "]";
      }
      case CBH_ERR_ATTR_MISSING: {
        // the column's path did not resolve in this input: the step that failed decides the text
        if (detail >= t->columns.size()) throw TraceIncomplete{};
        const Column& col = t->columns[detail];
        bool bad = false;
        Span holder = col.root == 0 ? iv.principal : col.root == 1 ? iv.resource : iv.aux;   // the message whose map field is the root
        u32 fnum = col.root == 2 ? 1u : col.root == 3 ? 2u : 4u;
        size_t k = 0;
        Span cur{nullptr, nullptr};
        if (col.keys.empty()) throw TraceIncomplete{};
        if (!map_get(holder, fnum, col.keys[0], cur, bad)) return "no such key: " + col.keys[0];
        k = 1;
        if (col.root == 3) {   // name -> {"claims": {...}}
          if (k >= col.keys.size()) throw TraceIncomplete{};
          if (col.keys[k] != "claims") return "no such key: " + col.keys[k];
          ++k;
          if (k >= col.keys.size()) throw TraceIncomplete{};
          Span inner{nullptr, nullptr};
          if (!map_get(cur, 1, col.keys[k], inner, bad)) return "no such key: " + col.keys[k];
          cur = inner; ++k;
        }
        for (; k < col.keys.size(); ++k) {
          Val v;
          if (!value(cur, v, bad) || v.kind != 5) return "no such overload";
          Span nxt{nullptr, nullptr};
          if (!map_get(v.s, 1, col.keys[k], nxt, bad)) return "no such key: " + col.keys[k];
          cur = nxt;
        }
        throw TraceIncomplete{};
      }
      default: throw TraceIncomplete{};
    }
  };

  struct Visit { u32 src = 0; u64 mask = 0; bool drfail = false; std::map<u32, std::pair<bool, TVal>> ok_parts; std::map<u32, std::string> err_parts;
                 std::map<u32, std::map<u32, TVal>> elems; };
  struct Key { u32 q, pass, ri, site, rule; bool operator<(const Key& o) const { return std::tie(q, pass, ri, site, rule) < std::tie(o.q, o.pass, o.ri, o.site, o.rule); } };
  std::vector<std::set<std::pair<std::string, std::string>>> errs(n);
  std::vector<std::map<Key, Visit>> visits(n);
  std::vector<u8> flags(n, 0), have_view(n, 0);
  std::vector<InputView> views(n);
  if (res->status) for (u32 q = 0; q < R; ++q) { const u32 o0 = RQ(RQ_ACT_OFF, q), c = RQ(RQ_ACT_CNT, q); for (u32 k = 0; k < c; ++k) if (res->status[o0 + k] == CBH_ST_UNSUPPORTED) flags[b->req_input[q]] |= CBI_TRACE_ERRORS_INCOMPLETE | CBI_TRACE_OUTPUTS_INCOMPLETE; }
  for (u32 x = 0; x < count; ++x) {
    const uint32_t* rec = records + (size_t)x * CBH_TRACE_RECORD_WORDS;
    const u32 q = rec[0], w1 = rec[1], w2 = rec[2], w3 = rec[3], kind = w1 & 0xF;
    if (q >= R) return fail("trace record refers to a request outside the batch");
    const u32 i = b->req_input[q];
    if (!have_view[i]) { view_of(i, views[i]); have_view[i] = 1; }
    const InputView& msg = views[i];
    try {
      if (kind == CBH_TR_INCOMPLETE) flags[i] |= CBI_TRACE_OUTPUTS_INCOMPLETE;
      else if (kind == CBH_TR_ERROR) {
        if (w2 >= t->trace_strings.size()) return fail("trace record refers to an unknown string");
        errs[i].emplace(t->trace_strings[w2], message((u64)rec[4] | ((u64)rec[5] << 32), msg));
      } else if (kind == CBH_TR_OUTPUT || kind == CBH_TR_OUTPUT_ERROR || kind == CBH_TR_OUTPUT_ELEMENT) {
        const u32 rule = kind != CBH_TR_OUTPUT_ERROR ? (w3 >> 8) : rec[4];
        Visit& v = visits[i][Key{q, (w1 >> 4) & 1, (w1 >> 12) & 0xFF, w1 >> 20, rule}];
        const u32 part = (w1 >> 6) & 63;
        v.src = w2; v.drfail = (w1 & 32u) != 0;
        if (kind == CBH_TR_OUTPUT_ELEMENT) { v.elems[part][rec[6]] = to_tval(w3 & 0xFF, (u64)rec[4] | ((u64)rec[5] << 32), 1); continue; }
        v.mask = (u64)rec[6] | ((u64)rec[7] << 32);
        if (kind == CBH_TR_OUTPUT) v.ok_parts[part] = {true, to_tval(w3 & 0xFF, (u64)rec[4] | ((u64)rec[5] << 32), 0)};
        else v.err_parts[part] = message((u64)w3 | ((u64)rec[5] << 32), msg);
      }
    } catch (const TraceIncomplete&) { flags[i] |= kind == CBH_TR_ERROR ? CBI_TRACE_ERRORS_INCOMPLETE : CBI_TRACE_OUTPUTS_INCOMPLETE; }
  }

  auto o = new cbi_outputs();
  o->offsets.reserve((size_t)n + 1); o->offsets.push_back(0); o->flags = flags;
  std::vector<u8>& ob = o->bytes;
  struct Entry { u64 a; u32 pass, ri, site; bool drfail; u32 rule; std::vector<u8> body; std::string_view action; };
  for (u32 i = 0; i < n; ++i) {
    // outputs (field 6), in the order check.go's loops reach them: action, policy kind, role, rule
    if (!visits[i].empty() && !have_view[i]) { view_of(i, views[i]); have_view[i] = 1; }
    const std::vector<std::string_view>& actions = views[i].actions;
    std::vector<Entry> entries;
    for (auto& kv : visits[i]) {
      const Key& key = kv.first; Visit& v = kv.second;
      auto tm = t->trace_templates.find(key.rule);
      if (tm == t->trace_templates.end() || v.src >= t->trace_strings.size()) { delete o; return fail("trace record refers to an unknown output"); }
      std::vector<u8> body;
      put_ld(body, 1, t->trace_strings[v.src]);
      bool lost = false;
      try {
        if (v.ok_parts.size() + v.err_parts.size() != tm->second.second) throw TraceIncomplete{};
        if (!v.err_parts.empty()) put_ld(body, 4, v.err_parts.begin()->second);   // the first part to fail in evaluation order
        else {
          std::vector<TVal> holes;
          for (auto& pk : v.ok_parts) {
            if (pk.first != holes.size()) throw TraceIncomplete{};
            TVal hv = pk.second.second;
            if (hv.local) {   // its elements were logged one by one
              const auto& el = v.elems[pk.first];
              if (el.size() != (size_t)hv.i) throw TraceIncomplete{};
              hv.local = false;
              u32 want = 0;
              for (const auto& ek : el) { if (ek.first != want++ || ek.second.local) throw TraceIncomplete{}; hv.items.push_back(ek.second); }
            }
            holes.push_back(std::move(hv));
          }
          std::vector<u8> vb; put_value(vb, assemble_template(tm->second.first, holes));
          put_varint(body, 2 << 3 | 2); put_varint(body, vb.size()); body.insert(body.end(), vb.begin(), vb.end());
        }
      } catch (const TraceIncomplete&) { lost = true; }
      if (lost) { o->flags[i] |= CBI_TRACE_OUTPUTS_INCOMPLETE; continue; }
      const u32 o0 = RQ(RQ_ACT_OFF, key.q), cnt = RQ(RQ_ACT_CNT, key.q);
      for (u32 k = 0; k < cnt && k < 64; ++k) if ((v.mask >> k) & 1) {
        const u64 tp = b->tuple_perm[o0 + k];
        if (tp < first[i] || tp >= first[i + 1] || tp - first[i] >= actions.size()) { delete o; return fail("batch does not belong to these inputs"); }
        entries.push_back(Entry{tp - first[i], key.pass, key.ri, key.site, v.drfail, key.rule & 0x7FFFFFu, body, actions[tp - first[i]]});
      }
    }
    std::stable_sort(entries.begin(), entries.end(), [](const Entry& x, const Entry& y) { return std::tie(x.a, x.pass, x.ri, x.site) < std::tie(y.a, y.pass, y.ri, y.site); });
    std::set<u32> seen_drfail;
    for (Entry& e : entries) {
      // the first visit of a rule whose derived-role condition fails emits nothing (check.go:343-347 caches "false" under the
      // evaluation key and moves on); later visits find the cached outcome and emit conditionNotMet
      if (e.drfail && seen_drfail.insert(e.rule).second) continue;
      put_ld(e.body, 3, e.action);
      put_varint(ob, 6 << 3 | 2); put_varint(ob, e.body.size()); ob.insert(ob.end(), e.body.begin(), e.body.end());
    }
    // evaluation_errors (field 7), sorted and deduplicated as CELErrors.All() (cel_errors.go:98-118)
    for (const auto& em : errs[i]) {
      std::vector<u8> ce; put_str(ce, 1, em.first); put_str(ce, 2, em.second);
      std::vector<u8> ee; put_varint(ee, 1 << 3 | 2); put_varint(ee, ce.size()); ee.insert(ee.end(), ce.begin(), ce.end());
      put_varint(ob, 7 << 3 | 2); put_varint(ob, ee.size()); ob.insert(ob.end(), ee.begin(), ee.end());
    }
    o->offsets.push_back(ob.size());
  }
  ob.reserve(1);
  *out = o;
  return 0;
}

extern "C" int cbi_trace_pb(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint32_t* records, uint32_t count,
                            const uint8_t* bytes, const uint64_t* offsets, uint32_t n, cbi_outputs** out) {
  if (!t || !b || !res || !out || (count && !records) || (n && (!bytes || !offsets))) return fail("cbi_trace_pb: null argument");
  return trace_decode(t, b, res, records, count, n, [&](u32 i, InputView& v) {   // CheckInput: resource 2, principal 3, actions 4, aux_data 5
    Span s{bytes + offsets[i], bytes + offsets[i + 1]}; Field f; bool bad = false;
    while (next(s, f, bad)) { if (f.wt != 2) continue;
      if (f.num == 2) v.resource = f.s; else if (f.num == 3) v.principal = f.s; else if (f.num == 5) v.aux = f.s; else if (f.num == 4) v.actions.push_back(sv(f.s)); }
  }, out);
}

// The same for the batch cbi_flatten_request_pb made of ONE CheckResourcesRequest: input i = its i-th resource entry, the principal
// is the request's, the auxiliary data what the caller passed to the flattener.
extern "C" int cbi_trace_request_pb(const cbi_table* t, const cbi_batch* b, const cbh_result* res, const uint32_t* records, uint32_t count,
                                    const uint8_t* request, uint64_t request_len, const uint8_t* aux_data, uint64_t aux_len, cbi_outputs** out) {
  if (!t || !b || !res || !out || (count && !records) || !request) return fail("cbi_trace_request_pb: null argument");
  RequestParts rp;
  if (!split_request(request, request_len, rp)) return fail("malformed CheckResourcesRequest");
  const Span aux = aux_data ? Span{aux_data, aux_data + aux_len} : Span{nullptr, nullptr};
  return trace_decode(t, b, res, records, count, (u32)rp.entries.size(), [&](u32 i, InputView& v) {   // entry: actions 1, resource 2
    v.principal = rp.principal; v.aux = aux;
    Span s = rp.entries[i]; Field f; bool bad = false;
    while (next(s, f, bad)) { if (f.wt != 2) continue; if (f.num == 1) v.actions.push_back(sv(f.s)); else if (f.num == 2) v.resource = f.s; }
  }, out);
}

extern "C" {
void cbi_outputs_free(cbi_outputs* o) { delete o; }
const uint8_t* cbi_outputs_bytes(const cbi_outputs* o) { return o ? o->bytes.data() : nullptr; }
const uint64_t* cbi_outputs_offsets(const cbi_outputs* o) { return o ? o->offsets.data() : nullptr; }
const uint8_t* cbi_outputs_flags(const cbi_outputs* o) { return o ? o->flags.data() : nullptr; }

void cbi_batch_free(cbi_batch* b) { delete b; }
const cbh_batch* cbi_batch_view(const cbi_batch* b) { return b ? &b->view : nullptr; }
const uint64_t* cbi_batch_tuple_perm(const cbi_batch* b) { return b ? b->tuple_perm.data() : nullptr; }
const uint32_t* cbi_batch_request_input(const cbi_batch* b) { return b ? b->req_input.data() : nullptr; }

}  // extern "C"
