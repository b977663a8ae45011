// event_gap3.hip — cost of hipStreamWriteValue64 / hipStreamWaitValue64 (command-processor memory ops) between two kernels of a stream,
// next to the one-wave kernel equivalents. Variants (T grid = (v+2)*64 threads):
//  0 nothing; 1 WriteValue64 on A between W and T; 2 WaitValue64 (already satisfied) on A; 3 WaitValue64 satisfied later by B's WriteValue
//  after a half-length kernel on B; 4 tiny signal kernel on A; 5 tiny wait kernel (satisfied) on A
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void W(uint4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, 1, 2, 3); }
__global__ void T(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void Sig(unsigned long long* s, unsigned long long v) { if (threadIdx.x == 0) __hip_atomic_store(s, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void Wait(const unsigned long long* s, unsigned long long v) { if (threadIdx.x == 0) while (__hip_atomic_load(s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < v) __builtin_amdgcn_s_sleep(8); }
int main() {
  size_t n = 33ull * 1024 * 1024 / 16;
  uint4 *buf, *buf2; unsigned* flag; CK(hipMalloc(&buf, n * 16)); CK(hipMalloc(&buf2, n * 16)); CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
  uint64_t* sig; CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory)); CK(hipMemset(sig, 0, 8));
  unsigned long long* sq; CK(hipMalloc(&sq, 64)); CK(hipMemset(sq, 0, 64));
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  uint64_t seq = 0;
  for (int v = 0; v < 6; v++) {
    for (int it = 0; it < 12; it++) {
      CK(hipDeviceSynchronize());
      seq++;
      if (v == 2) { CK(hipStreamWriteValue64(b, sig, seq, 0)); CK(hipStreamSynchronize(b)); }
      if (v == 3) { hipLaunchKernelGGL(W, dim3(1024), dim3(256), 0, b, buf2, n / 2); CK(hipStreamWriteValue64(b, sig, seq, 0)); }
      if (v == 5) { hipLaunchKernelGGL(Sig, dim3(1), dim3(64), 0, b, sq, seq); CK(hipStreamSynchronize(b)); }
      hipLaunchKernelGGL(W, dim3(2048), dim3(256), 0, a, buf, n);
      if (v == 1) CK(hipStreamWriteValue64(a, sig, seq, 0));
      if (v == 2 || v == 3) CK(hipStreamWaitValue64(a, sig, seq, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
      if (v == 4) hipLaunchKernelGGL(Sig, dim3(1), dim3(64), 0, a, sq, seq);
      if (v == 5) hipLaunchKernelGGL(Wait, dim3(1), dim3(64), 0, a, sq, seq);
      hipLaunchKernelGGL(T, dim3(v + 2), dim3(64), 0, a, flag);
      hipLaunchKernelGGL(W, dim3(2048), dim3(256), 0, a, buf, n);
      CK(hipDeviceSynchronize());
    }
    printf("variant %d done\n", v);
  }
  return 0;
}
