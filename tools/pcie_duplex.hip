// Can the host link carry both directions at once?  Times a 48 MB upload and a 41 MB download (page-locked host memory, two streams):
// each alone, both together by the copy engines, and the download as a KERNEL writing host memory beside the engine's upload.
//   hipcc --offload-arch=gfx950 -O3 tools/pcie_duplex.hip -o /tmp/pcie_duplex && /tmp/pcie_duplex
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void push(const uint4* src, uint4* dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t up = 48u << 20, down = 41u << 20;
  void *h_up, *h_down, *d_up, *d_down;
  CK(hipHostMalloc(&h_up, up, hipHostMallocMapped)); CK(hipHostMalloc(&h_down, down, hipHostMallocMapped));
  CK(hipMalloc(&d_up, up)); CK(hipMalloc(&d_down, down));
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now(); CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, a)); CK(hipStreamSynchronize(a)); double t_up = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, b)); CK(hipStreamSynchronize(b)); double t_down = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, a)); CK(hipMemcpyAsync(h_down, d_down, down, hipMemcpyDeviceToHost, b));
    CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b)); double t_both = now() - t0;
    t0 = now(); hipLaunchKernelGGL(push, dim3(1024), dim3(256), 0, b, (const uint4*)d_down, (uint4*)h_down, down / 16); CK(hipStreamSynchronize(b)); double t_kdown = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, a)); hipLaunchKernelGGL(push, dim3(1024), dim3(256), 0, b, (const uint4*)d_down, (uint4*)h_down, down / 16);
    CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b)); double t_kboth = now() - t0;
    t0 = now(); hipLaunchKernelGGL(push, dim3(1024), dim3(256), 0, a, (const uint4*)h_up, (uint4*)d_up, up / 16); CK(hipStreamSynchronize(a)); double t_kup = now() - t0;
    std::printf("upload %.0f us (%.1f GB/s)  download %.0f us (%.1f GB/s)  both by the engines %.0f us (sum %.0f)  download by a kernel %.0f us (%.1f GB/s)  engine upload + kernel download %.0f us  upload by a kernel %.0f us (%.1f GB/s)\n",
                t_up, up / t_up / 1e3, t_down, down / t_down / 1e3, t_both, t_up + t_down, t_kdown, down / t_kdown / 1e3, t_kboth, t_kup, up / t_kup / 1e3);
  }
  return 0;
}
