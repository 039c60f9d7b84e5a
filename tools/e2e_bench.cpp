// End-to-end throughput without Python in the loop: T host threads, each working through slices of a file of
// serialized CheckInputs:  cbi_flatten_pb -> cbh_check_batch -> cbi_assemble_pb.
//
//   python tools/export_wire.py C2 131072 /tmp/c2            # blob + messages + offsets
//   g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip \
//       -Wl,-rpath,$PWD/cerbos_amd -o /tmp/e2e_bench
//   /tmp/e2e_bench /tmp/c2 8192 3 1,8,32,64,128        [--host-only: skip the GPU call, results all zero]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "cerbos_ingest.h"

static std::vector<char> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: %s <dir> <slice_requests> <seconds> <threads,threads,...> [--host-only]\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  const uint32_t slice = (uint32_t)std::atoi(argv[2]);
  const double seconds = std::atof(argv[3]);
  const bool host_only = argc > 5 && !std::strcmp(argv[5], "--host-only");
  const auto blob = slurp(dir + "/table.blob"), data = slurp(dir + "/messages.bin"), offb = slurp(dir + "/offsets.bin");
  const uint64_t* off = (const uint64_t*)offb.data();
  const uint32_t n = (uint32_t)(offb.size() / 8 - 1);
  cbi_table* it = nullptr; cbh_table* gt = nullptr;
  if (cbi_table_open(blob.data(), blob.size(), &it)) { std::fprintf(stderr, "%s\n", cbi_last_error()); return 1; }
  if (!host_only) {
    cbh_config cfg{CBH_ABI_VERSION, 0};
    if (cbh_init(&cfg) || cbh_table_load(blob.data(), blob.size(), &gt)) { std::fprintf(stderr, "%s\n", cbh_last_error()); return 1; }
  }
  std::vector<uint32_t> starts;
  for (uint32_t a = 0; a < n; a += slice) starts.push_back(a);
  for (char* tok = std::strtok(argv[4], ","); tok; tok = std::strtok(nullptr, ",")) {
    const int T = std::atoi(tok);
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> decisions{0};
    std::atomic<int> failed{0};
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < T; ++k) th.emplace_back([&, k]() {
      std::vector<uint64_t> rel;
      std::vector<uint8_t> eff, st; std::vector<uint32_t> pol, sc; std::vector<uint64_t> edr;
      for (size_t i = (size_t)k; !stop.load(std::memory_order_relaxed); ++i) {
        const uint32_t a = starts[i % starts.size()], b = std::min(n, a + slice);
        rel.assign(off + a, off + b + 1);
        for (auto& x : rel) x -= off[a];
        const uint8_t* bytes = (const uint8_t*)data.data() + off[a];
        cbi_batch* batch = nullptr;
        if (cbi_flatten_pb(it, bytes, rel.data(), b - a, "default", "", 1, &batch)) { failed = 1; return; }
        const cbh_batch* v = cbi_batch_view(batch);
        eff.assign(v->n_tuples + 1, 0); st.assign(v->n_tuples + 1, 0); pol.assign(v->n_tuples + 1, 0); sc.assign(v->n_tuples + 1, 0xFFFFFFFFu);
        edr.assign(v->n_requests + 1, 0);
        cbh_result res{eff.data(), pol.data(), sc.data(), st.data(), edr.data()};
        cbh_params p{1700000000000000000ll, CBH_F_WANT_DERIVED_ROLES, 0};
        if (!host_only && cbh_check_batch(gt, v, &p, &res)) { failed = 2; return; }
        cbi_outputs* o = nullptr;
        if (cbi_assemble_pb(it, batch, &res, bytes, rel.data(), b - a, "default", &o)) { failed = 3; return; }
        decisions += v->n_tuples;
        cbi_outputs_free(o);
        cbi_batch_free(batch);
      }
    });
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop = true;
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (failed) { std::fprintf(stderr, "failed at step %d: %s / %s\n", failed.load(), cbi_last_error(), host_only ? "" : cbh_last_error()); return 1; }
    std::printf("{\"threads\": %d, \"host_only\": %s, \"decisions_per_s\": %.0f}\n", T, host_only ? "true" : "false", (double)decisions / dt);
    std::fflush(stdout);
  }
  return 0;
}
