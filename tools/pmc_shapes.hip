// tools/pmc_shapes.hip - how many bytes the memory side sees (rocprofv3 --pmc FETCH_SIZE) and how long it takes (--kernel-trace --stats)
// to bring 512 MiB of 8-byte values into LDS in the shapes the column cache could use.  One wave per workgroup; every shape reads
// every byte of the buffer exactly once (logically):
//   shape_lds_b32_pairs   two global_load_lds_dword per 64 values: the low dwords (stride 8 B), then the high dwords   (round 2-6's fill)
//   shape_lds_b128_half   one global_load_lds_dwordx4 per 64 values: lanes 0-31, 16 B each, 512 contiguous bytes
//   shape_reg_b64         one global_load_dwordx2 per lane into registers, then ds_write_b64
//   shape_u8              global_load_ubyte, 1 B/lane (the packed tags)
// hipcc --offload-arch=gfx950 -O3 tools/pmc_shapes.hip -o /tmp/pmc_shapes
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

static const size_t BYTES = (size_t)512 << 20;
#define G __attribute__((address_space(1)))
#define L __attribute__((address_space(3)))

__global__ __launch_bounds__(64) void shape_lds_b32_pairs(const uint32_t* src, uint32_t* sink, size_t n_vals) {
  __shared__ uint32_t buf[128];
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 64; i < n_vals; i += (size_t)gridDim.x * 64) {
    __builtin_amdgcn_global_load_lds((const G void*)(src + 2 * (i + threadIdx.x)), (L void*)buf, 4, 0, 0);
    __builtin_amdgcn_global_load_lds((const G void*)(src + 2 * (i + threadIdx.x) + 1), (L void*)(buf + 64), 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    acc ^= buf[threadIdx.x] ^ buf[64 + threadIdx.x];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(64) void shape_lds_b128_half(const uint32_t* src, uint32_t* sink, size_t n_vals) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[128];
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 64; i < n_vals; i += (size_t)gridDim.x * 64) {
    if (threadIdx.x < 32) __builtin_amdgcn_global_load_lds((const G void*)(src + 2 * i + 4 * threadIdx.x), (L void*)buf, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    acc ^= buf[2 * threadIdx.x] ^ buf[2 * threadIdx.x + 1];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(64) void shape_reg_b64(const uint64_t* src, uint32_t* sink, size_t n_vals) {
  __shared__ uint64_t buf[64];
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 64; i < n_vals; i += (size_t)gridDim.x * 64) {
    buf[threadIdx.x] = src[i + threadIdx.x];
    __builtin_amdgcn_s_waitcnt(0);
    acc ^= (uint32_t)buf[threadIdx.x ^ 1];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(64) void shape_u8(const uint8_t* src, uint32_t* sink, size_t n_bytes) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n_bytes; i += (size_t)gridDim.x * 64) acc ^= src[i];
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  void *a = nullptr, *b = nullptr; uint32_t* sink = nullptr;
  if (hipMalloc(&a, BYTES) != hipSuccess || hipMalloc(&b, BYTES) != hipSuccess || hipMalloc((void**)&sink, 256) != hipSuccess) { std::puts("alloc failed"); return 1; }
  (void)hipMemset(a, 1, BYTES); (void)hipMemset(b, 2, BYTES);
  (void)hipDeviceSynchronize();
  const int grid = 256 * 16;
  for (int rep = 0; rep < 3; ++rep) {   // alternate buffers so that no launch finds its data in the Infinity Cache
    hipLaunchKernelGGL(shape_lds_b32_pairs, dim3(grid), dim3(64), 0, 0, (const uint32_t*)a, sink, BYTES / 8);
    hipLaunchKernelGGL(shape_lds_b128_half, dim3(grid), dim3(64), 0, 0, (const uint32_t*)b, sink, BYTES / 8);
    hipLaunchKernelGGL(shape_reg_b64, dim3(grid), dim3(64), 0, 0, (const uint64_t*)a, sink, BYTES / 8);
    hipLaunchKernelGGL(shape_u8, dim3(grid), dim3(64), 0, 0, (const uint8_t*)b, sink, BYTES / 8);   // (an eighth of the buffer: 64 MiB of bytes)
  }
  if (hipDeviceSynchronize() != hipSuccess) { std::puts("kernel failed"); return 1; }
  std::printf("{\"bytes_per_launch\": %zu, \"u8_bytes\": %zu}\n", BYTES, BYTES / 8);
  return 0;
}
