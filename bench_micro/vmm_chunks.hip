// vmm_chunks.hip — is the speed of a table a property of the PHYSICAL CHUNKS it is made of? Twelve 1-GB chunks (hipMemCreate), each mapped on its own and probed with the merge
// kernel's request mix (2^20 random slot reads + exchange + 16-byte store inside the chunk); then 2-GB tables stitched from the two fastest and from the two slowest chunks,
// probed the same way, beside plain hipMalloc allocations of 2 GB. If chunks differ and tables follow their chunks, a table could be assembled from chosen chunks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
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
static hipStream_t s; static hipEvent_t e0, e1; static int salt = 0;
static float time_probe(Slot* p, uint64_t nslots) {
  float best = 1e9f;
  for (int rep = 0; rep < 6; rep++) {
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(probe, dim3((1u << 20) / 64), dim3(64), 0, s, p, nslots, 1u << 20, (uint32_t)(salt * 16 + rep));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms * 1000.f < best) best = ms * 1000.f;
  }
  salt++;
  return best;
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t chunk = 1ull << 30; const int NC = 12;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int dev = 0; CK(hipGetDevice(&dev));
  hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int k = 0; k < 2; k++) { Slot* p; CK(hipMalloc(&p, 2 * chunk)); printf("hipMalloc 2 GB #%d: %.2f us\n", k, time_probe(p, 2 * chunk / 32)); }
  std::vector<hipMemGenericAllocationHandle_t> hs(NC); std::vector<std::pair<float, int>> t(NC);
  for (int k = 0; k < NC; k++) {
    CK(hipMemCreate(&hs[k], chunk, &prop, 0));
    void* va = nullptr; CK(hipMemAddressReserve(&va, chunk, 0, nullptr, 0));
    CK(hipMemMap(va, chunk, 0, hs[k], 0)); CK(hipMemSetAccess(va, chunk, &acc, 1));
    CK(hipMemset(va, 0, chunk));
    const float a = time_probe((Slot*)va, chunk / 32), b = time_probe((Slot*)va, chunk / 32);
    t[k] = {std::min(a, b), k};
    printf("chunk %2d alone (1 GB): %.2f / %.2f us\n", k, a, b);
    CK(hipDeviceSynchronize()); CK(hipMemUnmap(va, chunk)); CK(hipMemAddressFree(va, chunk));
  }
  std::sort(t.begin(), t.end());
  auto stitched = [&](int a, int b, const char* tag) {
    void* va = nullptr; CK(hipMemAddressReserve(&va, 2 * chunk, 0, nullptr, 0));
    CK(hipMemMap(va, chunk, 0, hs[a], 0)); CK(hipMemMap((char*)va + chunk, chunk, 0, hs[b], 0)); CK(hipMemSetAccess(va, 2 * chunk, &acc, 1));
    printf("2-GB table of chunks %d + %d (%s: %.2f, %.2f alone): %.2f / %.2f us\n", a, b, tag, 0.f, 0.f, time_probe((Slot*)va, 2 * chunk / 32), time_probe((Slot*)va, 2 * chunk / 32));
    CK(hipDeviceSynchronize()); CK(hipMemUnmap(va, 2 * chunk)); CK(hipMemAddressFree(va, 2 * chunk));
  };
  printf("chunks by speed:"); for (auto& x : t) printf(" %d:%.1f", x.second, x.first); printf("\n");
  stitched(t[0].second, t[1].second, "the two fastest");
  stitched(t[NC - 1].second, t[NC - 2].second, "the two slowest");
  stitched(t[0].second, t[NC - 1].second, "fastest + slowest");
  stitched(t[0].second, t[1].second, "the two fastest, again");
  return 0;
}
