// tools/pmc_calib.hip - calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the decision
// kernel uses (MI355X_MICROARCH.md, "HBM": FETCH_SIZE halves 16 B/lane streaming reads; "other access widths and
// WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
//
// Each kernel moves exactly BYTES (512 MiB: past the 256 MiB Infinity Cache) once, with one access shape:
//   calib_read_b32      global_load_dword, 4 B/lane, a wave reads 256 consecutive bytes  (req_u32 fields, roles, actions)
//   calib_read_lds_b32  global_load_lds_dword, same addresses, straight into LDS           (the column cache)
//   calib_read_b128     global_load_dwordx4, 16 B/lane                                      (the guide's reference shape)
//   calib_write_b32     global_store_dword, 4 B/lane                                        (effect / status words)
//   calib_write_b128    global_store_dwordx4                                                (policy / scope words)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); the factor for a
// shape = BYTES / reported bytes.   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

static const size_t BYTES = (size_t)512 << 20;
#define G __attribute__((address_space(1)))
#define L __attribute__((address_space(3)))

__global__ __launch_bounds__(64) void calib_read_b32(const uint32_t* src, uint32_t* sink, size_t n_words) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 64) acc ^= src[i];
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(64) void calib_read_lds_b32(const uint32_t* src, uint32_t* sink, size_t n_words) {
  __shared__ uint32_t buf[64];
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 64; i < n_words; i += (size_t)gridDim.x * 64) {
    __builtin_amdgcn_global_load_lds((const G void*)(src + i + threadIdx.x), (L void*)buf, 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    acc ^= buf[threadIdx.x];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(64) void calib_read_b128(const uint4* src, uint32_t* sink, size_t n_vec) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 64) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(64) void calib_write_b32(uint32_t* dst, size_t n_words) {
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 64) dst[i] = (uint32_t)i;
}
__global__ __launch_bounds__(64) void calib_write_b128(uint4* dst, size_t n_vec) {
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 64) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main() {
  void *a = nullptr, *b = nullptr; uint32_t* sink = nullptr;
  if (hipMalloc(&a, BYTES) != hipSuccess || hipMalloc(&b, BYTES) != hipSuccess || hipMalloc((void**)&sink, 256) != hipSuccess) { std::puts("alloc failed"); return 1; }
  (void)hipMemset(a, 1, BYTES); (void)hipMemset(b, 2, BYTES);
  (void)hipDeviceSynchronize();
  const int grid = 256 * 16;
  for (int rep = 0; rep < 3; ++rep) {   // alternate buffers so that no launch finds its data in the Infinity Cache
    hipLaunchKernelGGL(calib_read_b32, dim3(grid), dim3(64), 0, 0, (const uint32_t*)a, sink, BYTES / 4);
    hipLaunchKernelGGL(calib_write_b32, dim3(grid), dim3(64), 0, 0, (uint32_t*)b, BYTES / 4);
    hipLaunchKernelGGL(calib_read_lds_b32, dim3(grid), dim3(64), 0, 0, (const uint32_t*)a, sink, BYTES / 4);
    hipLaunchKernelGGL(calib_write_b128, dim3(grid), dim3(64), 0, 0, (uint4*)b, BYTES / 16);
    hipLaunchKernelGGL(calib_read_b128, dim3(grid), dim3(64), 0, 0, (const uint4*)a, sink, BYTES / 16);
  }
  if (hipDeviceSynchronize() != hipSuccess) { std::puts("kernel failed"); return 1; }
  std::printf("{\"bytes_per_launch\": %zu}\n", BYTES);
  return 0;
}
