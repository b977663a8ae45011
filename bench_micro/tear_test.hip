// tear_test.hip — the merge kernel relies on an aligned 16-byte store never being observed half-written by an aligned 16-byte load
// (slot halves {ts,val}). This hammers that assumption: writer waves store (x, ~x ^ K) pairs to a small set of slots that sit in
// lines shared with other slots, reader waves on other workgroups (other CUs / XCDs) load them and check the pair. Any torn
// observation is counted. Also run with 8-byte halves written separately as a positive control (must show torn pairs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr uint64_t K = 0x5DEECE66DA5A5A5Aull;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_hammer(uint4* slots, uint32_t nslots, uint32_t iters, unsigned long long* torn, unsigned long long* reads) {
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  const bool writer = (blockIdx.x & 1u) == 0;                 // even workgroups write, odd ones read
  uint64_t x = (uint64_t)gid * 0x9E3779B97F4A7C15ull + 1;
  unsigned long long bad = 0, n = 0;
  for (uint32_t i = 0; i < iters; i++) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t s = (uint32_t)(x >> 40) % nslots;
    uint4* p = slots + 2 * (size_t)s + 1;                       // second half of a 32-byte slot, like (ts,val)
    if (writer) {
      const uint64_t a = x, b = ~x ^ K;
      if (SPLIT) {
        reinterpret_cast<volatile uint64_t*>(p)[0] = a; reinterpret_cast<volatile uint64_t*>(p)[1] = b;
      } else {
        *p = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
      }
    } else {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));   // one global_load_dwordx4 that goes to L2
      const uint64_t a = (uint64_t)v.x | ((uint64_t)v.y << 32), b = (uint64_t)v.z | ((uint64_t)v.w << 32);
      if (!(a == 0 && b == 0) && b != (~a ^ K)) bad++;
      n++;
    }
  }
  if (bad) atomicAdd(torn, bad);
  if (n) atomicAdd(reads, n);
}
int main(int argc, char** argv) {
  const uint32_t nslots = argc > 1 ? atoi(argv[1]) : 4096;     // small: every slot is written and read constantly
  const uint32_t iters = argc > 2 ? atoi(argv[2]) : 20000;
  uint4* slots; CK(hipMalloc(&slots, (size_t)nslots * 32)); 
  unsigned long long *d; CK(hipMalloc(&d, 16));
  for (int split = 0; split < 2; split++) {
    CK(hipMemset(slots, 0, (size_t)nslots * 32)); CK(hipMemset(d, 0, 16));
    if (split) hipLaunchKernelGGL(k_hammer<true>, dim3(4096), dim3(256), 0, 0, slots, nslots, iters, d, d + 1);
    else hipLaunchKernelGGL(k_hammer<false>, dim3(4096), dim3(256), 0, 0, slots, nslots, iters, d, d + 1);
    CK(hipDeviceSynchronize());
    unsigned long long h[2]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("%s: %llu reads of %u hot slots, torn pairs observed: %llu\n", split ? "two 8-byte stores (control)" : "one 16-byte store", h[1], nslots, h[0]);
  }
  return 0;
}
