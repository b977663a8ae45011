// event_gap2.hip — cost of a hipStreamWaitEvent on stream A (event recorded on stream B) between two kernels of A.
// variants (T grid = (v+2)*64 threads): 0 no wait; 1 wait on an event that completed long ago; 2 wait on an event whose kernel on B
// is still running when A reaches the wait (B runs W2 of ~same length started together with A's W).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void W(uint4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, 1, 2, 3); }
__global__ void T(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int main() {
  size_t n = 33ull * 1024 * 1024 / 16;
  uint4 *buf, *buf2; unsigned* flag; CK(hipMalloc(&buf, n * 16)); CK(hipMalloc(&buf2, n * 16)); CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int v = 0; v < 3; v++) {
    for (int it = 0; it < 12; it++) {
      CK(hipDeviceSynchronize());
      if (v == 1) { hipLaunchKernelGGL(T, dim3(1), dim3(64), 0, b, flag + 8); CK(hipEventRecord(ev, b)); CK(hipStreamSynchronize(b)); }
      if (v == 2) { hipLaunchKernelGGL(W, dim3(1024), dim3(256), 0, b, buf2, n / 2); CK(hipEventRecord(ev, b)); }
      hipLaunchKernelGGL(W, dim3(2048), dim3(256), 0, a, buf, n);
      if (v) CK(hipStreamWaitEvent(a, ev, 0));
      hipLaunchKernelGGL(T, dim3(v + 2), dim3(64), 0, a, flag);
      hipLaunchKernelGGL(W, dim3(2048), dim3(256), 0, a, buf, n);
      CK(hipDeviceSynchronize());
    }
    printf("variant %d done\n", v);
  }
  return 0;
}
