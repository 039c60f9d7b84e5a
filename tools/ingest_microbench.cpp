// Host path per request on one thread: cbi_flatten_pb + cbi_assemble_pb over an exported message set, best of 12 passes.
//   python tools/export_wire.py C2 131072 /tmp/c2w
//   g++ -O2 -std=c++17 -pthread -Iinclude tools/ingest_microbench.cpp cerbos_amd/csrc/cbh_ingest.cpp -o /tmp/ingest_microbench
//   /tmp/ingest_microbench /tmp/c2w
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
#include "cerbos_ingest.h"
static std::vector<char> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
int main(int argc, char** argv) {
  const std::string dir = argv[1];
  const auto blob = slurp(dir + "/table.blob"), data = slurp(dir + "/messages.bin"), offb = slurp(dir + "/offsets.bin");
  const uint64_t* off = (const uint64_t*)offb.data();
  const uint32_t n = (uint32_t)(offb.size() / 8 - 1);
  cbi_table* t = nullptr; cbi_table_open(blob.data(), blob.size(), &t);
  double bf = 1e9, ba = 1e9;
  const uint32_t S = 8192;
  for (int it = 0; it < 12; ++it) {
    double tf = 0, ta = 0;
    for (uint32_t a = 0; a < n; a += S) {
      const uint32_t c = std::min(S, n - a);
      std::vector<uint64_t> rel(c + 1); for (uint32_t k = 0; k <= c; ++k) rel[k] = off[a + k] - off[a];
      auto t0 = std::chrono::steady_clock::now();
      cbi_batch* b = nullptr; cbi_flatten_pb(t, (const uint8_t*)data.data() + off[a], rel.data(), c, "default", "", 1, &b);
      auto t1 = std::chrono::steady_clock::now();
      const cbh_batch* v = cbi_batch_view(b);
      std::vector<uint8_t> eff(v->n_tuples + 1, 1), st(v->n_tuples + 1, 0); std::vector<uint32_t> pol(v->n_tuples + 1, 2u << 28), sc(v->n_tuples + 1, 0xFFFFFFFFu); std::vector<uint64_t> edr(v->n_requests + 1, 0);
      cbh_result res{eff.data(), pol.data(), sc.data(), st.data(), edr.data()};
      auto t2 = std::chrono::steady_clock::now();
      cbi_outputs* o = nullptr; cbi_assemble_pb(t, b, &res, (const uint8_t*)data.data() + off[a], rel.data(), c, "default", &o);
      auto t3 = std::chrono::steady_clock::now();
      cbi_outputs_free(o); cbi_batch_free(b);
      tf += std::chrono::duration<double>(t1 - t0).count(); ta += std::chrono::duration<double>(t3 - t2).count();
    }
    bf = std::min(bf, tf); ba = std::min(ba, ta);
  }
  std::printf("flatten %.1f ns/request  assemble %.1f ns/request  -> %.2f M decisions/s (4 per request)\n", bf / n * 1e9, ba / n * 1e9, 4.0 * n / (bf + ba) / 1e6);
}
