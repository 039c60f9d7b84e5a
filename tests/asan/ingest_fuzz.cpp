// Mutation fuzz of libcerbos_ingest's parsers under AddressSanitizer + UBSan (tests/test_ingest_asan.py builds this file
// together with cerbos_amd/csrc/cbh_ingest.cpp using -fsanitize=address,undefined and runs it on files the test writes).
// The wire walkers take client bytes (integration/go/gpu_cgo.go hands the request over unparsed): every mutated input
// must either flatten or be refused with an error - never read outside its buffers.
//   ingest_fuzz <dir> <iterations> <seed>      <dir>: table.blob, messages.bin, offsets.bin, request.bin
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "cerbos_ingest.h"

static std::vector<uint8_t> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static uint64_t rng_state;
static uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }

static void mutate(std::vector<uint8_t>& m) {
  const uint32_t kind = rnd() % 8;
  if (m.empty()) { m.push_back((uint8_t)rnd()); return; }
  const size_t at = rnd() % m.size();
  switch (kind) {
    case 0: m[at] ^= (uint8_t)(1u << (rnd() % 8)); break;                       // bit flip
    case 1: m[at] = (uint8_t)rnd(); break;                                       // byte
    case 2: m.resize(at); break;                                                 // truncate
    case 3: m.insert(m.begin() + at, (uint8_t)rnd()); break;                     // insert
    case 4: m.erase(m.begin() + at); break;                                      // delete
    case 5: { const uint8_t v[2] = {0x18, 0x05}; m.insert(m.begin() + at, v, v + 2); break; }   // a varint where a Value expects bytes
    case 6: m[at] = 0xFF; if (at + 1 < m.size()) m[at + 1] = 0xFF; break;        // runaway varint
    default: { const size_t n = 1 + rnd() % 16; m.insert(m.begin() + at, n, (uint8_t)0x80); break; }   // continuation bytes
  }
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const std::string dir = argv[1];
  const int iters = std::atoi(argv[2]);
  rng_state = (uint64_t)std::atoll(argv[3]) * 2654435761u + 1;
  const auto blob = slurp(dir + "/table.blob"), data = slurp(dir + "/messages.bin"), offb = slurp(dir + "/offsets.bin"), request = slurp(dir + "/request.bin");
  const uint64_t* off = (const uint64_t*)offb.data();
  const uint32_t n = (uint32_t)(offb.size() / 8 - 1);
  cbi_table* t = nullptr;
  if (cbi_table_open(blob.data(), blob.size(), &t)) { std::fprintf(stderr, "table: %s\n", cbi_last_error()); return 1; }
  long ok = 0, refused = 0;
  for (int it = 0; it < iters; ++it) {
    // a few messages of the set, one or two of them mutated, through the flattener and the assembler
    const uint32_t first = rnd() % n, cnt = 1 + rnd() % 4;
    std::vector<uint8_t> bytes; std::vector<uint64_t> offs{0};
    for (uint32_t k = 0; k < cnt; ++k) {
      const uint32_t i = (first + k) % n;
      std::vector<uint8_t> m(data.begin() + off[i], data.begin() + off[i + 1]);
      if (rnd() % 2) { mutate(m); if (rnd() % 3 == 0) mutate(m); }
      bytes.insert(bytes.end(), m.begin(), m.end());
      offs.push_back(bytes.size());
    }
    // exact-size heap copy: a read one byte past the input is an ASan error, not a lucky hit on vector slack
    uint8_t* exact = (uint8_t*)std::malloc(bytes.size() ? bytes.size() : 1);
    if (!bytes.empty()) std::memcpy(exact, bytes.data(), bytes.size());
    cbi_batch* b = nullptr;
    if (cbi_flatten_pb_mt(t, exact, offs.data(), cnt, "default", "", 1, (int)(1 + rnd() % 2), &b) == 0) {
      const cbh_batch* v = cbi_batch_view(b);
      std::vector<uint8_t> eff(v->n_tuples + 1, 1), st(v->n_tuples + 1, 0);
      std::vector<uint32_t> pol(v->n_tuples + 1, 2u << 28), sc(v->n_tuples + 1, 0xFFFFFFFFu);
      std::vector<uint64_t> edr(v->n_requests + 1, 0);
      cbh_result res{eff.data(), pol.data(), sc.data(), st.data(), edr.data()};
      cbi_outputs* o = nullptr;
      if (cbi_assemble_pb(t, b, &res, exact, offs.data(), cnt, "default", &o) == 0) cbi_outputs_free(o);
      // the trace log's consumer on records made up at random (kinds, string ids, value tags, heap references, rule words):
      // every record is either decoded, reported incomplete or refused - never a read outside the batch / the table
      if (cbi_table_trace_scope(t) != 0) {
        const uint32_t nrec = rnd() % 24;
        std::vector<uint32_t> rec((size_t)nrec * CBH_TRACE_RECORD_WORDS + 1);
        for (uint32_t r = 0; r < nrec; ++r) {
          uint32_t* w = &rec[(size_t)r * CBH_TRACE_RECORD_WORDS];
          w[0] = rnd() % 8 ? rnd() % (v->n_requests + 1) : rnd();
          w[1] = (1 + rnd() % 4) | ((rnd() & 0xFFFFFFu) << 4);
          w[2] = rnd() % 4 ? rnd() % 64 : rnd();
          w[3] = rnd() % 3 ? (rnd() % 12) | ((rnd() % 40) << 8) : rnd();
          w[4] = rnd() % 2 ? rnd() % 64 : rnd(); w[5] = rnd() % 2 ? 0 : rnd(); w[6] = rnd(); w[7] = rnd() % 2 ? 0 : rnd();
        }
        cbi_outputs* to = nullptr;
        if (cbi_trace_pb(t, b, &res, rec.data(), nrec, exact, offs.data(), cnt, &to) == 0) cbi_outputs_free(to);
      }
      cbi_batch_free(b);
      ++ok;
    } else ++refused;
    std::free(exact);
    // the request-level entry: one CheckResourcesRequest, mutated
    std::vector<uint8_t> rq = request;
    mutate(rq); if (rnd() % 2) mutate(rq);
    uint8_t* rexact = (uint8_t*)std::malloc(rq.size() ? rq.size() : 1);
    if (!rq.empty()) std::memcpy(rexact, rq.data(), rq.size());
    cbi_batch* rb = nullptr;
    if (cbi_flatten_request_pb(t, rexact, rq.size(), nullptr, 0, "default", "", 1, 1, &rb) == 0) {
      const cbh_batch* v = cbi_batch_view(rb);
      std::vector<uint8_t> eff(v->n_tuples + 1, 2), st(v->n_tuples + 1, 0);
      std::vector<uint32_t> pol(v->n_tuples + 1, 1u << 28), sc(v->n_tuples + 1, 0xFFFFFFFFu);
      std::vector<uint64_t> edr(v->n_requests + 1, 0);
      cbh_result res{eff.data(), pol.data(), sc.data(), st.data(), edr.data()};
      cbi_outputs* o = nullptr;
      if (cbi_assemble_response_pb(t, rb, &res, rexact, rq.size(), "default", &o) == 0) cbi_outputs_free(o);
      if (cbi_table_trace_scope(t) != 0) {   // the request-level trace consumer and the assembly that folds its outputs in
        const uint32_t nrec = rnd() % 16;
        std::vector<uint32_t> rec((size_t)nrec * CBH_TRACE_RECORD_WORDS + 1);
        for (uint32_t r = 0; r < nrec; ++r) {
          uint32_t* w = &rec[(size_t)r * CBH_TRACE_RECORD_WORDS];
          w[0] = rnd() % (v->n_requests + 1); w[1] = (1 + rnd() % 4) | ((rnd() & 0xFFFFFFu) << 4); w[2] = rnd() % 64;
          w[3] = (rnd() % 12) | ((rnd() % 40) << 8); w[4] = rnd() % 64; w[5] = 0; w[6] = rnd(); w[7] = 0;
        }
        cbi_outputs* to = nullptr;
        if (cbi_trace_request_pb(t, rb, &res, rec.data(), nrec, rexact, rq.size(), nullptr, 0, &to) == 0) {
          cbi_outputs* o2 = nullptr;
          if (cbi_assemble_response_traced_pb(t, rb, &res, rexact, rq.size(), "default", to, &o2) == 0) cbi_outputs_free(o2);
          cbi_outputs_free(to);
        }
      }
      cbi_batch_free(rb);
      ++ok;
    } else ++refused;
    std::free(rexact);
  }
  cbi_table_close(t);
  std::printf("ok %ld refused %ld\n", ok, refused);
  return 0;
}
