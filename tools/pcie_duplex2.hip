// Follow-up to pcie_duplex.hip: the sliced road's pattern - 4 x 12 MB up, 4 x 10 MB down - (a) every slice on its own stream (upload,
// then its download), (b) all uploads on ONE stream and all downloads on ANOTHER, (c) as (a) but the downloads enqueued only after a
// host wait for the slice's upload (what the road's threads do).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t up = 12u << 20, down = 10u << 20;
  char *h_up, *h_down, *d_up, *d_down;
  CK(hipHostMalloc((void**)&h_up, 4 * up, hipHostMallocPortable)); CK(hipHostMalloc((void**)&h_down, 4 * down, hipHostMallocPortable));
  CK(hipMalloc((void**)&d_up, 4 * up)); CK(hipMalloc((void**)&d_down, 4 * down));
  hipStream_t s[6]; for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    for (int k = 0; k < 4; ++k) { CK(hipMemcpyAsync(d_up + k * up, h_up + k * up, up, hipMemcpyHostToDevice, s[k])); CK(hipMemcpyAsync(h_down + k * down, d_down + k * down, down, hipMemcpyDeviceToHost, s[k])); }
    for (int k = 0; k < 4; ++k) CK(hipStreamSynchronize(s[k]));
    const double ta = now() - t0;
    t0 = now();
    for (int k = 0; k < 4; ++k) CK(hipMemcpyAsync(d_up + k * up, h_up + k * up, up, hipMemcpyHostToDevice, s[4]));
    for (int k = 0; k < 4; ++k) CK(hipMemcpyAsync(h_down + k * down, d_down + k * down, down, hipMemcpyDeviceToHost, s[5]));
    CK(hipStreamSynchronize(s[4])); CK(hipStreamSynchronize(s[5]));
    const double tb = now() - t0;
    t0 = now();
    for (int k = 0; k < 4; ++k) { CK(hipMemcpyAsync(d_up + k * up, h_up + k * up, up, hipMemcpyHostToDevice, s[k])); CK(hipStreamSynchronize(s[k])); CK(hipMemcpyAsync(h_down + k * down, d_down + k * down, down, hipMemcpyDeviceToHost, s[k])); }
    for (int k = 0; k < 4; ++k) CK(hipStreamSynchronize(s[k]));
    const double tc = now() - t0;
    std::printf("4 x (12 MB up, 10 MB down): (a) a stream per slice %.0f us   (b) one upload stream + one download stream %.0f us   (c) as (a), download enqueued after the upload landed %.0f us   [one direction after the other: %.0f us]\n",
                ta, tb, tc, (4 * up + 4 * down) / 56.5e3);
  }
  return 0;
}
