// cbh_wire_req.h - the device road for what the server actually receives: serialized cerbos.request.v1.CheckResourcesRequest
// messages (request.proto:222-273: 1 request_id, 2 include_meta, 3 principal, 4 resources [1 actions, 2 resource], 5 aux_data,
// 6 request_context) instead of the CheckInputs svc.CheckResources builds from them (cerbos_svc.go:274-288: one CheckInput per
// resource entry - the request's id and principal, the entry's resource and actions, the AuxData the server derived from the
// request's JWT).
//
// Two launches in front of the flattener of cbh_wire.h, which then runs unchanged on messages that never existed on the host:
//
//   cbh_wire_req_count_kernel   one lane per request: its resource entries and the bytes of their CheckInputs
//   cbh_wire_req_split_kernel   one wave per request: writes those CheckInputs (engine.proto: 1 request_id, 2 resource,
//                               3 principal, 4 actions, 5 aux_data) into the message buffer, every byte range copied by the
//                               wave's 64 lanes side by side, and the messages' offsets
//
// between them the host turns the per-request counts into offsets (two prefix sums over n_requests numbers).  Same grammar as
// cbh_ingest.cpp split_request / cbi_flatten_request_pb (the host road), which tests/test_request_road.py holds it against: a scalar
// named twice keeps its last value, a message field named twice (the principal, an entry's resource) is the MERGE of its occurrences -
// their bytes back to back, which is how protobuf defines the merge; the request's own aux_data (a raw JWT) is the server's to verify - what a CheckInput carries is the engine
// AuxData the caller hands over per request, or nothing.
#pragma once
#include "cbh_wire.h"

#define CBH_WREQ_BAD 0xFFFFFFFFu   /* n_inputs of a malformed request */

struct WireReqArgs {
  const CBH_G u8* req; const CBH_G u64* roff; u32 n; u32 end;   // request r = req[roff[r] .. roff[r + 1]); end = bytes of all requests
  const CBH_G u8* aux; const CBH_G u64* aoff; u64 aux_end;      // serialized engine AuxData of request r = aux[aoff[r] .. aoff[r + 1]), or null; aux_end = its bytes
  CBH_G u32* n_inputs;            // [n] resource entries of request r (CBH_WREQ_BAD: malformed)
  CBH_G u64* n_bytes;             // [n] bytes of their CheckInputs
  CBH_G u8* flags;                // [n] bit 0 = include_meta
  const CBH_G u32* first_input;   // [n] exclusive prefix of n_inputs  (split)
  const CBH_G u64* first_byte;    // [n] exclusive prefix of n_bytes   (split)
  CBH_G u8* msg; CBH_G u64* moff; // the CheckInputs: message first_input[r] + e = msg[moff[..] ..); moff[total inputs] = total bytes
};

struct WReqTop { WSpan rid, principal; u32 n_principal, principal_total; bool has_rid, has_principal, include_meta; };
// the request's own fields; the entries are walked by the caller.  A scalar (request_id, include_meta): the last occurrence wins.
// The principal is a MESSAGE field: protobuf merges its occurrences (what proto.Unmarshal did to the request the server validated,
// logged and took its JWT from), and parsing the concatenation of their bytes IS that merge - so a request that names field 3 more
// than once gets a CheckInput principal made of all of them back to back (n_principal, principal_total; `principal` = the last one).
__device__ __forceinline__ void w_req_top(WMsg m, WSpan s, WReqTop& t, u32& n_entries, bool& bad) {
  t.rid.p = t.rid.e = t.principal.p = t.principal.e = 0; t.has_rid = t.has_principal = t.include_meta = false; n_entries = 0;
  t.n_principal = 0; t.principal_total = 0;
  WField f;
  while (w_next(m, s, f, bad)) {
    if (f.num == 2u && f.wt == 0u) t.include_meta = f.v != 0;
    if (f.wt != 2u) continue;
    if (f.num == 1u) { t.rid = f.s; t.has_rid = true; }
    else if (f.num == 3u) { t.principal = f.s; t.has_principal = true; ++t.n_principal; t.principal_total += f.s.e - f.s.p; }
    else if (f.num == 4u) ++n_entries;
  }
}
__device__ __forceinline__ u32 w_ld_size(u32 len) { return 1u + w_varint_size((u64)len) + len; }   // tag (fields 1-5: one byte), length, bytes

#ifdef CBH_HOSTSIM
static void cbh_wire_req_count_kernel(WireReqArgs a)
#else
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_req_count_kernel(WireReqArgs a)
#endif
{
  const u32 r = blockIdx.x * CBH_BLOCK + threadIdx.x;
  if (r >= a.n) return;
  const u64 o0 = a.roff[r], o1 = a.roff[r + 1u];
  bool bad = o1 < o0 || o1 > (u64)a.end;
  u32 n_entries = 0; u64 bytes = 0; WReqTop top; top.include_meta = false;
  if (!bad) {
    WMsg m = (WMsg)a.req;
    WSpan s; s.p = (u32)o0; s.e = (u32)o1;
    w_req_top(m, s, top, n_entries, bad);
    u32 aux_len = 0;
    if (a.aoff) { const u64 a0 = a.aoff[r], a1 = a.aoff[r + 1u]; if (a1 < a0 || a1 > a.aux_end || a1 - a0 > 0xFFFFFFFFull) bad = true; else aux_len = (u32)(a1 - a0); }
    // what every CheckInput of the request repeats
    const u64 shared = (top.has_rid ? w_ld_size(top.rid.e - top.rid.p) : 0u) + (top.has_principal ? w_ld_size(top.principal_total) : 0u)
                     + (aux_len ? w_ld_size(aux_len) : 0u);
    WField f;
    while (!bad && w_next(m, s, f, bad)) {
      if (f.num != 4u || f.wt != 2u) continue;
      WSpan e = f.s; WField g; u32 res_len = 0; bool has_res = false;
      bytes += shared;
      while (w_next(m, e, g, bad)) {
        if (g.wt != 2u) continue;
        if (g.num == 1u) bytes += w_ld_size(g.s.e - g.s.p);
        else if (g.num == 2u) { res_len += g.s.e - g.s.p; has_res = true; }   // (a message field: its occurrences merge, see w_req_top)
      }
      if (has_res) bytes += w_ld_size(res_len);
    }
  }
  a.n_inputs[r] = bad ? CBH_WREQ_BAD : n_entries;
  a.n_bytes[r] = bad ? 0ull : bytes;
  a.flags[r] = (u8)((!bad && top.include_meta) ? 1u : 0u);
}

// one length-delimited field of a CheckInput at msg[out ..): the header by lane 0, the bytes by all lanes (`src` = where they sit)
__device__ __forceinline__ void w_req_emit_header(const WireReqArgs& a, u32 lane, u64& out, u32 field, u32 len) {
  const u32 vs = w_varint_size((u64)len);
  if (lane == 0u) {
    a.msg[out] = (u8)((field << 3) | 2u);
    u32 v = len;
    for (u32 k = 0; k < vs; ++k) { a.msg[out + 1u + k] = (u8)((v & 0x7Fu) | (k + 1u < vs ? 0x80u : 0u)); v >>= 7; }
  }
  out += 1u + vs;
}
__device__ __forceinline__ void w_req_emit_bytes(const WireReqArgs& a, u32 lane, u64& out, const CBH_G u8* src, u32 len) {
  CBH_G u8* dst = a.msg + out;
  for (u32 j = lane; j < len; j += 64u) dst[j] = src[j];
  out += len;
}
__device__ __forceinline__ void w_req_emit(const WireReqArgs& a, u32 lane, u64& out, u32 field, const CBH_G u8* src, u32 len) {
  w_req_emit_header(a, lane, out, field, len);
  w_req_emit_bytes(a, lane, out, src, len);
}
// ... and one made of every occurrence of field `fnum` in `in`, back to back (`total` = the sum of their lengths)
__device__ __forceinline__ void w_req_emit_merged(const WireReqArgs& a, u32 lane, u64& out, u32 field, WMsg m, WSpan in, u32 fnum, u32 total) {
  w_req_emit_header(a, lane, out, field, total);
  WField g; bool bad = false;
  while (w_next(m, in, g, bad)) if (g.wt == 2u && g.num == fnum) w_req_emit_bytes(a, lane, out, a.req + g.s.p, g.s.e - g.s.p);
}

#ifdef CBH_HOSTSIM
static void cbh_wire_req_split_kernel(WireReqArgs a)
#else
__global__ __launch_bounds__(CBH_BLOCK) void cbh_wire_req_split_kernel(WireReqArgs a)
#endif
{
  const u32 lane = threadIdx.x & 63u;
  const u32 r = blockIdx.x * (CBH_BLOCK / 64u) + threadIdx.x / 64u;   // every lane of the wave walks the same bytes: uniform control flow
  if (r >= a.n) return;
  const u32 ni = a.n_inputs[r];
  u64 out = a.first_byte[r];
  u32 idx = a.first_input[r];
  if (r + 1u == a.n && lane == 0u) a.moff[idx + (ni == CBH_WREQ_BAD ? 0u : ni)] = out + a.n_bytes[r];   // the end of the last message
  if (ni == CBH_WREQ_BAD || ni == 0u) return;
  WMsg m = (WMsg)a.req;
  WSpan s; s.p = (u32)a.roff[r]; s.e = (u32)a.roff[r + 1u];
  WReqTop top; u32 n_entries = 0; bool bad = false;
  w_req_top(m, s, top, n_entries, bad);
  const CBH_G u8* aux = nullptr; u32 aux_len = 0;
  if (a.aoff) { aux = a.aux + a.aoff[r]; aux_len = (u32)(a.aoff[r + 1u] - a.aoff[r]); }
  WField f;
  while (w_next(m, s, f, bad)) {
    if (f.num != 4u || f.wt != 2u) continue;
    if (lane == 0u) a.moff[idx] = out;
    ++idx;
    if (top.has_rid) w_req_emit(a, lane, out, 1u, a.req + top.rid.p, top.rid.e - top.rid.p);
    WSpan e = f.s, res; WField g; u32 n_res = 0, res_total = 0;
    res.p = res.e = 0;
    while (w_next(m, e, g, bad)) { if (g.wt == 2u && g.num == 2u) { res = g.s; ++n_res; res_total += g.s.e - g.s.p; } }
    if (n_res == 1u) w_req_emit(a, lane, out, 2u, a.req + res.p, res.e - res.p);
    else if (n_res) w_req_emit_merged(a, lane, out, 2u, m, f.s, 2u, res_total);
    if (top.n_principal == 1u) w_req_emit(a, lane, out, 3u, a.req + top.principal.p, top.principal.e - top.principal.p);
    else if (top.has_principal) { WSpan whole; whole.p = (u32)a.roff[r]; whole.e = (u32)a.roff[r + 1u]; w_req_emit_merged(a, lane, out, 3u, m, whole, 3u, top.principal_total); }
    e = f.s;
    while (w_next(m, e, g, bad)) { if (g.wt == 2u && g.num == 1u) w_req_emit(a, lane, out, 4u, a.req + g.s.p, g.s.e - g.s.p); }
    if (aux_len) w_req_emit(a, lane, out, 5u, aux, aux_len);
  }
}
