// anyorder_test.hip — does hipExtAnyOrderLaunch (a launch without the barrier bit) take effect on gfx950 / ROCm 7.2? hip_ext.h says "not supported on
// AMD GFX9xx boards". One stream: Spin(T) -> Spin(T) -> Spin(T); the third launched (a) normally, (b) with hipExtAnyOrderLaunch. If the flag is honoured
// the third kernel overlaps the second one: total ~2T instead of ~3T.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void Spin(unsigned long long ticks, unsigned* out) {   // 100 MHz wall clock
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) atomicAdd(out, 1u);
}
int main() {
  unsigned* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const unsigned long long T = 20000;   // 200 us
  for (int variant = 0; variant < 2; variant++) {
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s, T, d);
      hipLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s, T, d);
      if (variant == 0) hipLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s, T, d);
      else hipExtLaunchKernelGGL(Spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, T, d);
      CK(hipGetLastError());
      CK(hipStreamSynchronize(s));
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("%s: three 200-us kernels on one stream took %.0f us\n", variant ? "third with hipExtAnyOrderLaunch" : "all in order               ", us);
    }
  }
  return 0;
}
