// vmm_placement.hip — does HOW a 1.5 GB table is put together from physical memory decide the random-access rate? One probe (the merge kernel's request mix:
// 2^20 random 32-byte slot reads + an atomic exchange + a 16-byte store) over: hipMalloc, hipExtMallocWithFlags(Contiguous), and one virtual range stitched from
// hipMemCreate chunks of several sizes (hipMemAddressReserve + hipMemMap). us per probe launch, best of 5; several allocations of each kind alive at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__host__ __device__ inline uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
struct alignas(32) Slot { uint64_t id; uint32_t field, head; int64_t ts, val; };
__global__ __launch_bounds__(64) void probe(Slot* slots, uint64_t nslots, uint32_t n, uint32_t salt) {
  const uint32_t j = blockIdx.x * 64u + threadIdx.x;
  if (j >= n) return;
  const uint64_t h = mix64(((uint64_t)salt << 32) | j);
  Slot* sl = slots + __umul64hi(h, nslots);
  const uint4* q = reinterpret_cast<const uint4*>(sl);
  const uint4 lo = q[0], hi = q[1];
  const uint32_t prev = atomicExch(&sl->head, j);
  if (((lo.x ^ hi.x ^ prev) & 7u) != 5u) reinterpret_cast<uint4*>(sl)[1] = make_uint4(j, lo.y, hi.z, prev);
}
static float time_probe(Slot* p, uint64_t nslots, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int salt) {
  float best = 1e9f;
  for (int rep = 0; rep < 6; rep++) {
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(probe, dim3((1u << 20) / 64), dim3(64), 0, s, p, nslots, 1u << 20, (uint32_t)(salt * 16 + rep));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms * 1000.f < best) best = ms * 1000.f;
  }
  return best;
}
int main() {
  const uint64_t nslots = 52000000ull & ~3ull, bytes = nslots * 32;   // 1.66 GB
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int dev = 0; CK(hipGetDevice(&dev));
  hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("table %llu MB; VMM granularity (recommended) %zu KB\n", (unsigned long long)(bytes >> 20), gran >> 10);
  int salt = 0;
  for (int round = 0; round < 2; round++) {
    for (int k = 0; k < 3; k++) { Slot* p; CK(hipMalloc(&p, bytes)); printf("hipMalloc #%d: %.2f us\n", k, time_probe(p, nslots, s, e0, e1, salt++)); }   // (kept alive: like the tuner's candidates)
    for (size_t chunk_mb : {2ul, 16ul, 128ul, 1024ul}) {
      size_t chunk = chunk_mb << 20; if (chunk < gran) chunk = gran;
      const size_t total = (bytes + chunk - 1) / chunk * chunk;
      void* va = nullptr; CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
      std::vector<hipMemGenericAllocationHandle_t> hs;
      for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char*)va + off, chunk, 0, h, 0)); hs.push_back(h);
      }
      hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
      CK(hipMemSetAccess(va, total, &acc, 1));
      printf("VMM, %zu chunks of %zu MB: %.2f us\n", hs.size(), chunk >> 20, time_probe((Slot*)va, nslots, s, e0, e1, salt++));
      fflush(stdout);
    }
  }
  return 0;
}
