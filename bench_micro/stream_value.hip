// Microbenchmark (round 2): cost of cross-stream ordering with stream memory operations (hipStreamWriteValue64 / hipStreamWaitValue64 on
// signal memory: command-processor packets, no kernel) against the one-wave signal/wait kernels bmx_seq_* use. Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__global__ void k_busy(float* p, int iters) { float x = p[threadIdx.x]; for (int i = 0; i < iters; i++) x = x * 1.0001f + 0.5f; p[threadIdx.x] = x; }
__global__ void k_sig(unsigned long long* s, unsigned long long v) { if (threadIdx.x == 0) __hip_atomic_store(s, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_wait(const unsigned long long* s, unsigned long long v) { if (threadIdx.x == 0) while (__hip_atomic_load(s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < v) __builtin_amdgcn_s_sleep(8); }
int main() {
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, -1));
  float* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
  unsigned long long* sig = nullptr; hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
  if (e != hipSuccess) { printf("no signal memory: %s\n", hipGetErrorString(e)); return 0; }
  unsigned long long* seq; CK(hipMalloc(&seq, 8)); CK(hipMemset(seq, 0, 8)); *(volatile unsigned long long*)sig = 0;
  const int N = 200, IT = 20000;   // busy kernel ~ tens of us
  auto run = [&](int mode) {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= N; i++) {
      hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, a, p, IT);
      if (mode == 1) { hipLaunchKernelGGL(k_sig, dim3(1), dim3(64), 0, a, seq, (unsigned long long)i); hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, b, (const unsigned long long*)seq, (unsigned long long)i); }
      if (mode == 2) { CK(hipStreamWriteValue64(a, sig, (uint64_t)(1000000ull * 0 + i + 100000ull * 0), 0)); CK(hipStreamWaitValue64(b, sig, (uint64_t)i, hipStreamWaitValueGte, ~0ull)); }
      hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, (mode == 0 ? a : b), p + 256, 1);
      if (mode == 1) { hipLaunchKernelGGL(k_sig, dim3(1), dim3(64), 0, b, seq + 0, (unsigned long long)i); }   // symmetry: a second signal as the pipeline has
      if (mode == 2) { CK(hipStreamWriteValue64(b, sig, (uint64_t)i, 0)); }
    }
    CK(hipDeviceSynchronize());
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
  };
  run(0);
  double base = run(0), kern = run(1);
  *(volatile unsigned long long*)sig = 0;
  double val = run(2);
  printf("per iteration: same stream %.1f us | two streams ordered by signal+wait kernels %.1f us (+%.1f) | by hipStreamWriteValue64/WaitValue64 %.1f us (+%.1f)\n", base, kern, kern - base, val, val - base);
  return 0;
}
