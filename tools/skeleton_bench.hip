// tools/skeleton_bench.hip - what bounds a one-wave-per-64-requests kernel over a 1M-tuple C2 batch, compute aside?
// Kernels with the decision kernel's memory skeleton (same bytes, same dependency depth) and nothing else:
//   empty64 / empty256     launch + dispatch of 3907 x 64 / 977 x 256 threads
//   load1                  10 request fields + 6 columns x 9 B (all independent, 1 round trip)
//   load2                  + roles / actions through the offsets (2 dependent round trips)
//   load2_store            + 48 B of results per request
//   load2_store_spin N     + N dependent FMA iterations per lane between the loads and the stores (compute stand-in)
// Buffers rotate over 16 sets (> 1 GB) so every launch reads HBM.    hipcc --offload-arch=gfx950 -O3 tools/skeleton_bench.hip -o /tmp/skel
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64; typedef uint8_t u8;
struct Set { const u32* req; const u32* roles; const u32* act; const u8* ctag; const u64* cval; u8* eff; u8* st; u32* pol; u32* scp; u64* edr; };
static const u32 NR = 250000, NF = 10, NC = 6;

__global__ void empty_k(u32* sink) { if (sink == (u32*)1) sink[0] = 1; }

template <int LEVEL, bool STORE>
__global__ __launch_bounds__(64) void skel(Set s, u32 spin, u32* sink) {
  const u32 r = blockIdx.x * 64 + threadIdx.x;
  if (r >= NR) return;
  u32 f[NF];
#pragma unroll
  for (u32 k = 0; k < NF; ++k) f[k] = s.req[(size_t)k * NR + r];
  u64 acc = 0;
#pragma unroll
  for (u32 k = 0; k < NC; ++k) acc += s.cval[(size_t)k * NR + r] + s.ctag[(size_t)k * NR + r];
  u32 x = 0;
#pragma unroll
  for (u32 k = 0; k < NF; ++k) x ^= f[k];
  if (LEVEL >= 2) {
    const u32 role_off = f[6] % (NR), act_off = (f[8] % NR) * 4;
    x ^= s.roles[role_off] ^ s.act[act_off] ^ s.act[act_off + 1] ^ s.act[act_off + 2] ^ s.act[act_off + 3];
  }
  float v = (float)(x & 0xFF) + (float)(acc & 0xFF);
  for (u32 i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  x ^= (u32)v;
  if (STORE) {
    *(u32*)(s.eff + 4 * (size_t)r) = x; *(u32*)(s.st + 4 * (size_t)r) = x >> 1;
    uint4 p = make_uint4(x, x + 1, x + 2, x + 3);
    *(uint4*)(s.pol + 4 * (size_t)r) = p; *(uint4*)(s.scp + 4 * (size_t)r) = p;
    s.edr[r] = x;
  } else if (x == 0x12345678u) sink[0] = x;
}

int main() {
  const int NSETS = 16;
  std::vector<Set> sets(NSETS);
  auto dm = [](size_t bytes) { void* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) { std::puts("alloc failed"); exit(1); } (void)hipMemset(p, 0, bytes); return p; };
  for (auto& s : sets) {
    s.req = (const u32*)dm((size_t)NF * NR * 4); s.roles = (const u32*)dm((size_t)NR * 2 * 4); s.act = (const u32*)dm((size_t)NR * 4 * 4);
    s.ctag = (const u8*)dm((size_t)NC * NR + 64); s.cval = (const u64*)dm((size_t)NC * NR * 8);
    s.eff = (u8*)dm((size_t)NR * 4); s.st = (u8*)dm((size_t)NR * 4); s.pol = (u32*)dm((size_t)NR * 16); s.scp = (u32*)dm((size_t)NR * 16); s.edr = (u64*)dm((size_t)NR * 8);
  }
  u32* sink = (u32*)dm(256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int REPS = 64;
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 8; ++i) launch(i);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < REPS; ++i) launch(i);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::printf("%-28s %8.2f us per launch (back to back)\n", name, ms * 1000.0f / REPS);
  };
  const u32 G = (NR + 63) / 64;
  timeit("empty 3907 x 64", [&](int) { hipLaunchKernelGGL(empty_k, dim3(G), dim3(64), 0, 0, sink); });
  timeit("empty 977 x 256", [&](int) { hipLaunchKernelGGL(empty_k, dim3((NR + 255) / 256), dim3(256), 0, 0, sink); });
  timeit("load1", [&](int i) { hipLaunchKernelGGL((skel<1, false>), dim3(G), dim3(64), 0, 0, sets[i % NSETS], 0u, sink); });
  timeit("load2", [&](int i) { hipLaunchKernelGGL((skel<2, false>), dim3(G), dim3(64), 0, 0, sets[i % NSETS], 0u, sink); });
  timeit("load2_store", [&](int i) { hipLaunchKernelGGL((skel<2, true>), dim3(G), dim3(64), 0, 0, sets[i % NSETS], 0u, sink); });
  for (u32 spin : {250u, 500u, 1000u, 2000u}) {
    char nm[64]; std::snprintf(nm, sizeof nm, "load2_store_spin %u", spin);
    timeit(nm, [&](int i) { hipLaunchKernelGGL((skel<2, true>), dim3(G), dim3(64), 0, 0, sets[i % NSETS], spin, sink); });
  }
  return 0;
}
