// Wire-inclusive end to end, BOTH roads, without Python in the loop: T host threads, each working through slices of a file of
// serialized CheckInputs and producing serialized CheckOutputs.
//   host road    cbi_flatten_pb  -> cbh_check_batch                                        -> cbi_assemble_pb
//   device road  cbh_wire_flatten -> cbh_check_resident -> cbh_result_download + cbh_wire_spans_download -> cbi_assemble_wire_pb
//                (the GPU parses the messages, cbh_wire.h; the host only assembles the answers)
//   device_out   cbh_wire_flatten -> cbh_check_resident -> cbh_wire_outputs
//                (the GPU writes the answers too: the host thread moves bytes and nothing else)
//
//   python tools/export_wire.py C2 262144 /tmp/c2w
//   g++ -O2 -std=c++17 -pthread -Iinclude tools/e2e_wire_bench.cpp -Lcerbos_amd -lcerbos_ingest -lcerbos_hip -Wl,-rpath,$PWD/cerbos_amd -o /tmp/e2e_wire_bench
//   /tmp/e2e_wire_bench /tmp/c2w <slice_requests> <seconds> <threads,threads,...> [device|device_out|host|both] [verify]
// `verify`: before timing, every slice goes down both roads and the serialized outputs must be identical.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <thread>
#include <vector>

#include "cerbos_ingest.h"

static std::vector<char> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
using Clock = std::chrono::steady_clock;
static double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct Slice { uint8_t* bytes; std::vector<uint64_t> rel; uint32_t n; };   // bytes: page-locked copy of the slice's messages

struct Scratch {   // per thread, page-locked: results and spans come back by DMA
  uint8_t *eff = nullptr, *st = nullptr; uint32_t *pol = nullptr, *sc = nullptr, *in_span = nullptr, *act_span = nullptr, *act_off = nullptr; uint64_t* edr = nullptr;
  uint8_t* out = nullptr; size_t out_cap = 0; uint64_t* out_off = nullptr; uint8_t* out_flags = nullptr;
  void alloc(uint32_t max_req, uint32_t max_tup) {
    out_cap = (size_t)max_req * 256 + 4096; out = (uint8_t*)cbh_alloc_pinned(out_cap);
    out_off = (uint64_t*)cbh_alloc_pinned(8 * ((size_t)max_req + 2)); out_flags = (uint8_t*)cbh_alloc_pinned((size_t)max_req + 1);
    eff = (uint8_t*)cbh_alloc_pinned(max_tup + 1); st = (uint8_t*)cbh_alloc_pinned(max_tup + 1);
    pol = (uint32_t*)cbh_alloc_pinned(4 * ((size_t)max_tup + 1)); sc = (uint32_t*)cbh_alloc_pinned(4 * ((size_t)max_tup + 1));
    edr = (uint64_t*)cbh_alloc_pinned(8 * ((size_t)max_req + 1));
    in_span = (uint32_t*)cbh_alloc_pinned(48 * ((size_t)max_req + 1)); act_span = (uint32_t*)cbh_alloc_pinned(8 * ((size_t)max_tup + 1));
    act_off = (uint32_t*)cbh_alloc_pinned(4 * ((size_t)max_req + 2));
  }
};

static const cbh_params PARAMS{1700000000000000000ll, CBH_F_WANT_DERIVED_ROLES, 0};

// one slice down the device road; returns the tuples decided, 0 on failure
static uint32_t device_road(cbh_table* gt, const cbi_table* it, const Slice& s, Scratch& x, cbi_outputs** out, double* phase) {
  const auto t0 = Clock::now();
  cbh_device_batch* db = nullptr; cbh_wire_info info;
  if (cbh_wire_flatten(gt, 0, s.bytes, s.rel.data(), s.n, "default", "", nullptr, 0, &db, &info) != 0) { std::fprintf(stderr, "cbh_wire_flatten: %s\n", cbh_last_error()); return 0; }
  const auto t1 = Clock::now();
  cbh_result res{x.eff, x.pol, x.sc, x.st, x.edr};
  if (cbh_check_resident(gt, db, &PARAMS) != 0 || cbh_result_download(gt, db, &res) != 0 || cbh_wire_spans_download(gt, db, x.in_span, x.act_span, x.act_off) != 0) {
    std::fprintf(stderr, "device road: %s\n", cbh_last_error()); cbh_batch_release(db); return 0;
  }
  const auto t2 = Clock::now();
  if (cbi_assemble_wire_pb(it, &res, s.bytes, s.rel.data(), s.n, info.n_tuples, x.in_span, x.act_span, x.act_off, "default", 1, out) != 0) {
    std::fprintf(stderr, "cbi_assemble_wire_pb: %s\n", cbi_last_error()); cbh_batch_release(db); return 0;
  }
  const auto t3 = Clock::now();
  cbh_batch_release(db);
  if (phase) { phase[0] += secs(t0, t1); phase[1] += secs(t1, t2); phase[2] += secs(t2, t3); }
  return info.n_tuples;
}

// one slice, the answers written by the device into x.out / x.out_off / x.out_flags
static uint32_t device_out_road(cbh_table* gt, const Slice& s, Scratch& x, double* phase) {
  const auto t0 = Clock::now();
  cbh_device_batch* db = nullptr; cbh_wire_info info;
  if (cbh_wire_flatten(gt, 0, s.bytes, s.rel.data(), s.n, "default", "", nullptr, 0, &db, &info) != 0) { std::fprintf(stderr, "cbh_wire_flatten: %s\n", cbh_last_error()); return 0; }
  const auto t1 = Clock::now();
  if (cbh_check_resident(gt, db, &PARAMS) != 0) { std::fprintf(stderr, "cbh_check_resident: %s\n", cbh_last_error()); cbh_batch_release(db); return 0; }
  size_t need = 0;
  int rc = cbh_wire_outputs(gt, db, x.out, x.out_cap, x.out_off, x.out_flags, &need);
  if (rc == 2) { cbh_free_pinned(x.out); x.out_cap = need + need / 4; x.out = (uint8_t*)cbh_alloc_pinned(x.out_cap); rc = cbh_wire_outputs(gt, db, x.out, x.out_cap, x.out_off, x.out_flags, &need); }
  if (rc != 0) { std::fprintf(stderr, "cbh_wire_outputs: %s\n", cbh_last_error()); cbh_batch_release(db); return 0; }
  const auto t2 = Clock::now();
  cbh_batch_release(db);
  if (phase) { phase[0] += secs(t0, t1); phase[1] += secs(t1, t2); }
  return info.n_tuples;
}

// one slice through cbh_wire_check_pb: the whole device road in ONE call (the library cuts the slice up and overlaps the pieces)
static uint32_t onecall_road(cbh_table* gt, const Slice& s, Scratch& x, double* phase) {
  const auto t0 = Clock::now();
  cbh_wire_info info; size_t need = 0;
  int rc = cbh_wire_check_pb(gt, 0, s.bytes, s.rel.data(), s.n, "default", "", nullptr, 0, &PARAMS, x.out, x.out_cap, x.out_off, x.out_flags, &need, &info);
  if (rc == 2) {
    cbh_free_pinned(x.out); x.out_cap = need + need / 4; x.out = (uint8_t*)cbh_alloc_pinned(x.out_cap);
    rc = cbh_wire_check_pb(gt, 0, s.bytes, s.rel.data(), s.n, "default", "", nullptr, 0, &PARAMS, x.out, x.out_cap, x.out_off, x.out_flags, &need, &info);
  }
  if (rc != 0) { std::fprintf(stderr, "cbh_wire_check_pb: %s\n", cbh_last_error()); return 0; }
  if (phase) phase[0] += secs(t0, Clock::now());
  return info.n_tuples;
}

static uint32_t host_road(cbh_table* gt, const cbi_table* it, const Slice& s, Scratch& x, cbi_outputs** out, double* phase) {
  const auto t0 = Clock::now();
  cbi_batch* b = nullptr;
  if (cbi_flatten_pb(it, s.bytes, s.rel.data(), s.n, "default", "", 1, &b) != 0) { std::fprintf(stderr, "cbi_flatten_pb: %s\n", cbi_last_error()); return 0; }
  const auto t1 = Clock::now();
  const cbh_batch* v = cbi_batch_view(b);
  cbh_result res{x.eff, x.pol, x.sc, x.st, x.edr};
  if (cbh_check_batch(gt, v, &PARAMS, &res) != 0) { std::fprintf(stderr, "cbh_check_batch: %s\n", cbh_last_error()); cbi_batch_free(b); return 0; }
  const auto t2 = Clock::now();
  if (cbi_assemble_pb(it, b, &res, s.bytes, s.rel.data(), s.n, "default", out) != 0) { std::fprintf(stderr, "cbi_assemble_pb: %s\n", cbi_last_error()); cbi_batch_free(b); return 0; }
  const auto t3 = Clock::now();
  const uint32_t T = v->n_tuples;
  cbi_batch_free(b);
  if (phase) { phase[0] += secs(t0, t1); phase[1] += secs(t1, t2); phase[2] += secs(t2, t3); }
  return T;
}

int main(int argc, char** argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: %s <dir> <slice_requests> <seconds> <threads,...> [device|device_out|onecall|host|both] [verify]\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  const uint32_t slice = (uint32_t)std::atoi(argv[2]);
  const double seconds = std::atof(argv[3]);
  const std::string roads = argc > 5 ? argv[5] : "both";
  const bool verify = argc > 6 && !std::strcmp(argv[6], "verify");
  const auto blob = slurp(dir + "/table.blob"), data = slurp(dir + "/messages.bin"), offb = slurp(dir + "/offsets.bin");
  const uint64_t* off = (const uint64_t*)offb.data();
  const uint32_t n = (uint32_t)(offb.size() / 8 - 1);
  cbi_table* it = nullptr; cbh_table* gt = nullptr;
  if (cbi_table_open(blob.data(), blob.size(), &it)) { std::fprintf(stderr, "%s\n", cbi_last_error()); return 1; }
  cbh_config cfg; std::memset(&cfg, 0, sizeof(cfg)); cfg.abi_version = CBH_ABI_VERSION; cfg.n_devices = 1;
  if (cbh_init(&cfg) || cbh_table_load(blob.data(), blob.size(), &gt)) { std::fprintf(stderr, "%s\n", cbh_last_error()); return 1; }
  std::vector<Slice> slices;
  uint32_t max_tup = 0;
  for (uint32_t a = 0; a < n; a += slice) {
    const uint32_t b = std::min(n, a + slice);
    Slice s; s.n = b - a; s.rel.assign(off + a, off + b + 1);
    for (auto& v : s.rel) v -= off[a];
    s.bytes = (uint8_t*)cbh_alloc_pinned(s.rel.back() + 64);
    std::memcpy(s.bytes, data.data() + off[a], s.rel.back());
    slices.push_back(std::move(s));
  }
  max_tup = slice * 64;   // generous: results sized for the widest request shape the device road takes
  if (max_tup > (1u << 24)) max_tup = 1u << 24;
  std::printf("{\"messages\": %u, \"wire_bytes_per_request\": %.1f, \"slice_requests\": %u}\n", n, (double)off[n] / n, slice);
  if (verify) {
    Scratch x; x.alloc(slice, max_tup);
    size_t same = 0;
    for (const Slice& s : slices) {
      cbi_outputs *a = nullptr, *b = nullptr;
      if (!device_road(gt, it, s, x, &a, nullptr) || !host_road(gt, it, s, x, &b, nullptr) || !device_out_road(gt, s, x, nullptr)) return 1;
      {   // ... and the answers the device wrote
        const uint64_t* ob2 = cbi_outputs_offsets(b);
        if (x.out_off[s.n] != ob2[s.n] || std::memcmp(x.out_off, ob2, ((size_t)s.n + 1) * 8) || std::memcmp(x.out, cbi_outputs_bytes(b), ob2[s.n]) ||
            std::memcmp(x.out_flags, cbi_outputs_flags(b), s.n)) { std::fprintf(stderr, "VERIFY FAILED: the device-written outputs differ\n"); return 1; }
      }
      const uint64_t *oa = cbi_outputs_offsets(a), *ob = cbi_outputs_offsets(b);
      if (oa[s.n] != ob[s.n] || std::memcmp(oa, ob, ((size_t)s.n + 1) * 8) || std::memcmp(cbi_outputs_bytes(a), cbi_outputs_bytes(b), oa[s.n]) ||
          std::memcmp(cbi_outputs_flags(a), cbi_outputs_flags(b), s.n)) { std::fprintf(stderr, "VERIFY FAILED: the two roads disagree\n"); return 1; }
      same += s.n;
      cbi_outputs_free(a); cbi_outputs_free(b);
    }
    std::printf("{\"verified_identical_outputs\": %zu}\n", same);
  }
  for (const char* road : {"onecall", "device_out", "device", "host"}) {
    if (roads != road && (roads != "both" || !std::strcmp(road, "onecall"))) continue;
    const bool dev = !std::strcmp(road, "device"), dev_out = !std::strcmp(road, "device_out"), one = !std::strcmp(road, "onecall");
    std::string tl = argv[4];
    for (char* tok = std::strtok(tl.data(), ","); tok; tok = std::strtok(nullptr, ",")) {
      const int T = std::atoi(tok);
      std::atomic<bool> stop{false};
      std::atomic<uint64_t> decisions{0};
      std::atomic<int> failed{0};
      std::vector<std::vector<double>> phases(T, std::vector<double>(3, 0.0));
      std::vector<std::thread> th;
      const auto t0 = Clock::now();
      for (int k = 0; k < T; ++k) th.emplace_back([&, k]() {
        Scratch x; x.alloc(slice, max_tup);
        uint64_t mine = 0;
        for (size_t i = (size_t)k; !stop.load(std::memory_order_relaxed); ++i) {
          const Slice& s = slices[i % slices.size()];
          cbi_outputs* o = nullptr;
          const uint32_t d = one ? onecall_road(gt, s, x, phases[k].data()) : dev_out ? device_out_road(gt, s, x, phases[k].data()) : dev ? device_road(gt, it, s, x, &o, phases[k].data()) : host_road(gt, it, s, x, &o, phases[k].data());
          if (!d) { failed = 1; break; }
          if (o) cbi_outputs_free(o);
          mine += d;
        }
        decisions += mine;
      });
      std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
      stop = true;
      for (auto& t : th) t.join();
      const double dt = secs(t0, Clock::now());
      if (failed) return 1;
      double p[3] = {0, 0, 0};
      for (auto& v : phases) for (int j = 0; j < 3; ++j) p[j] += v[j];
      const double sum = p[0] + p[1] + p[2] > 0 ? p[0] + p[1] + p[2] : 1;
      std::printf("{\"road\": \"%s\", \"threads\": %d, \"decisions_per_s\": %.4g, \"share_flatten\": %.3f, \"share_check_and_download\": %.3f, \"share_assemble\": %.3f}\n",
                  road, T, (double)decisions.load() / dt, p[0] / sum, p[1] / sum, p[2] / sum);
      std::fflush(stdout);
    }
  }
  return 0;
}
