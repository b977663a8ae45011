// Microbenchmark (round 2): what does a random write-back cost as a function of how much of the 128-B line is written,
// and what does a random 128-B line read cost as a function of the table size (L2 / Infinity Cache / HBM)? Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }

// one lane per probe; the lane reads RD x 16 B of its line (0 = no read) and then stores ST x 16 B (1 = half a sector, 2 = one 32-B sector,
// 4 = 64 B, 8 = the whole line) for pct % of the probes
template <int RD, int ST>
__global__ __launch_bounds__(256) void k_rw(uint4* tab, uint64_t nlines, uint32_t n, uint64_t seed, uint32_t pct, uint32_t* out) {
  uint32_t j = blockIdx.x * 256u + threadIdx.x; if (j >= n) return;
  uint64_t l = __umul64hi(mix64(j + seed), nlines);
  uint4* p = tab + 8 * l;
  uint32_t x = j;
  uint4 r[RD > 0 ? RD : 1];
#pragma unroll
  for (int k = 0; k < RD; k++) { r[k] = p[k]; }
#pragma unroll
  for (int k = 0; k < RD; k++) x ^= r[k].x ^ r[k].y ^ r[k].z ^ r[k].w;
  if ((mix64(j * 7 + seed) % 100) < pct) {
#pragma unroll
    for (int k = 0; k < ST; k++) p[k] = make_uint4(x, j, (uint32_t)seed, k);
  }
  if (RD) out[j] = x;
}
// a wave (64 lanes) cooperates: 8 lanes per line, each lane 16 B -> 8 lines per wave-instruction, whole lines read and written
template <bool WRITE>
__global__ __launch_bounds__(256) void k_line_coop(uint4* tab, uint64_t nlines, uint32_t n, uint64_t seed, uint32_t pct, uint32_t* out) {
  uint32_t t = blockIdx.x * 256u + threadIdx.x;
  uint32_t j = t >> 3, sub = t & 7; if (j >= n) return;
  uint64_t l = __umul64hi(mix64(j + seed), nlines);
  uint4* p = tab + 8 * l + sub;
  uint4 r = *p;
  uint32_t x = r.x ^ r.y ^ r.z ^ r.w;
  if (WRITE && (mix64(j * 7 + seed) % 100) < pct) *p = make_uint4(x, j, (uint32_t)seed, sub);
  if (sub == 0) out[j] = x;
}
__global__ void k_fill(uint4* t, size_t n16){ size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n16;i+=st){ uint32_t v=(uint32_t)i; t[i]=make_uint4(v,v*3,v*5,v*7);} }
template<class F> float timeN(F f, int iters, int reps){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); std::vector<float> ts;
  for(int i=0;i<iters;i++){ CK(hipEventRecord(e0)); for(int r=0;r<reps;r++) f(i*reps+r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ts.push_back(ms/reps);}
  std::sort(ts.begin(),ts.end()); return ts[ts.size()/2]*1000.f; }
#define SEED(i) ((uint64_t)(i)*1315423911ull+17)

int main() {
  const uint32_t n = 1000000; const int g = (n + 255) / 256, g8 = (n * 8 + 255) / 256;
  uint32_t* out; CK(hipMalloc(&out, (size_t)n * 4));
  const int IT = 7, R = 10;
  const uint64_t sizes_mb[7] = {16, 64, 128, 256, 400, 640, 1408};
  for (int si = 0; si < 7; si++) {
    size_t bytes = sizes_mb[si] << 20; uint64_t nlines = bytes / 128;
    uint4* tab; CK(hipMalloc(&tab, bytes));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, tab, bytes / 16); CK(hipDeviceSynchronize());
#define RUN(RD, ST, PCT) timeN([&](int i){ hipLaunchKernelGGL((k_rw<RD, ST>), dim3(g), dim3(256), 0, 0, tab, nlines, n, SEED(i), (uint32_t)(PCT), out); }, IT, R)
    float rd2 = RUN(2, 1, 0), rd8 = RUN(8, 1, 0);
    float w1 = RUN(0, 1, 84), w2 = RUN(0, 2, 84), w4 = RUN(0, 4, 84), w8 = RUN(0, 8, 84);
    float rw1 = RUN(2, 1, 84), rw2 = RUN(2, 2, 84), rw4 = RUN(8, 4, 84), rw8 = RUN(8, 8, 84);
    float w8_100 = RUN(0, 8, 100), rw8_100 = RUN(8, 8, 100), rw1_100 = RUN(2, 1, 100);
    float cr = timeN([&](int i){ hipLaunchKernelGGL((k_line_coop<false>), dim3(g8), dim3(256), 0, 0, tab, nlines, n, SEED(i), 84u, out); }, IT, R);
    float cw = timeN([&](int i){ hipLaunchKernelGGL((k_line_coop<true>), dim3(g8), dim3(256), 0, 0, tab, nlines, n, SEED(i), 84u, out); }, IT, R);
    printf("table %5llu MB | read 32B %.1f 128B %.1f | write-only(84%%) 16B %.1f 32B %.1f 64B %.1f 128B %.1f | read+write(84%%) 32r16w %.1f 32r32w %.1f 128r64w %.1f 128r128w %.1f | 100%%: w128 %.1f r128w128 %.1f r32w16 %.1f | 8-lane coop line: read %.1f read+write84 %.1f  us per 1M\n",
           (unsigned long long)sizes_mb[si], rd2, rd8, w1, w2, w4, w8, rw1, rw2, rw4, rw8, w8_100, rw8_100, rw1_100, cr, cw);
    fflush(stdout);
    CK(hipFree(tab));
  }
  return 0;
}
