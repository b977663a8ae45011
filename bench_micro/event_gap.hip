// event_gap.hip — what does an event record between two kernels of one stream cost when the first kernel left tens of MB dirty?
// stream A: W (writes `mb` MB) -> [event record, variant] -> T (tiny);  stream B: wait(event) -> T.  Read the gaps from
// `rocprofv3 --kernel-trace`: W end -> next T start on stream A, per variant (the variant is encoded in T's grid size).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void W(uint4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, 1, 2, 3); }
__global__ void T(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int main(int argc, char** argv) {
  size_t mb = argc > 1 ? atoi(argv[1]) : 33;
  size_t n = mb * 1024 * 1024 / 16;
  uint4* buf; unsigned* flag; CK(hipMalloc(&buf, n * 16)); CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  unsigned flagsv[4] = {0xFFFFFFFFu /*no event*/, hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence, hipEventDisableTiming | hipEventReleaseToDevice};
  const char* names[4] = {"no event", "default(disable timing)", "DisableSystemFence", "ReleaseToDevice"};
  hipEvent_t t0, t1, t2; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreate(&t2));
  for (int v = 0; v < 4; v++) {
    hipEvent_t ev = nullptr;
    if (flagsv[v] != 0xFFFFFFFFu) CK(hipEventCreateWithFlags(&ev, flagsv[v]));
    float acc = 0;
    for (int it = 0; it < 12; it++) {
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(W, dim3(2048), dim3(256), 0, a, buf, n);
      if (ev) { CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(b, ev, 0)); hipLaunchKernelGGL(T, dim3(1), dim3(64), 0, b, flag + 8); }
      hipLaunchKernelGGL(T, dim3(v + 2), dim3(64), 0, a, flag);     // grid size v+2 marks the variant in the trace
      hipLaunchKernelGGL(W, dim3(2048), dim3(256), 0, a, buf, n);
      CK(hipDeviceSynchronize());
    }
    printf("variant %d: %s done\n", v, names[v]);
    if (ev) CK(hipEventDestroy(ev));
  }
  return 0;
}
