// Microbenchmark (round 2): what does a probe cost WITHOUT the per-delta claim atomic?
// Pass A = random 32-B slot read + plain stores into the slot's own 32-B sector (head tag 4 B + (ts,val) 16 B) for 84 % of the probes;
// pass B (a second launch) = re-read of the same slots (verification of the claim tag after the kernel boundary).
// Compared with the round-1 shape (read + atomicExch + 16-B store). Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }

// MODE 0: read only; 1: + pair store; 2: + head store + pair store; 3: + whole-slot store (2 x 16 B); 4: + atomicExch(head) + pair store;
// 5: as 2 with nontemporal stores; 6: as 2 but the delta columns (28 B) are read as well and slot_of/wflag written (full pass-A shape)
template <int MODE>
__global__ __launch_bounds__(256) void k_passA(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t epoch, uint32_t* out,
                                               const uint64_t* id, const uint32_t* field, const int64_t* ts, const int64_t* val, uint32_t* slot_of, uint8_t* wflag) {
  uint32_t j = blockIdx.x * 256u + threadIdx.x; if (j >= n) return;
  uint64_t key = j + seed; int64_t a = 0, v = 0;
  if (MODE == 6) { key = id[j] + field[j] + seed; a = ts[j]; v = val[j]; }
  uint64_t s = __umul64hi(mix64(key), nslots);
  uint4 lo = tab[2 * s], hi = tab[2 * s + 1];
  uint32_t x = lo.x ^ lo.y ^ lo.z ^ lo.w ^ hi.x ^ hi.y ^ hi.z ^ hi.w;
  const bool win = (mix64(j * 7 + seed) % 100) < 84;
  const uint32_t tag = (epoch << 24) | j;
  if (MODE != 0 && win) {
    uint4 p = make_uint4(j, (uint32_t)seed + (uint32_t)a, x, 7u + (uint32_t)v);
    if (MODE == 1) tab[2 * s + 1] = p;
    if (MODE == 2 || MODE == 6) { reinterpret_cast<uint32_t*>(tab + 2 * s)[3] = tag; tab[2 * s + 1] = p; }
    if (MODE == 3) { lo.w = tag; tab[2 * s] = lo; tab[2 * s + 1] = p; }
    if (MODE == 4) { x ^= atomicExch(reinterpret_cast<uint32_t*>(tab + 2 * s) + 3, tag); p.z = x; tab[2 * s + 1] = p; }
    if (MODE == 5) { __builtin_nontemporal_store(tag, reinterpret_cast<uint32_t*>(tab + 2 * s) + 3);
                     __builtin_nontemporal_store(p.x, reinterpret_cast<uint32_t*>(tab + 2 * s + 1)); __builtin_nontemporal_store(p.y, reinterpret_cast<uint32_t*>(tab + 2 * s + 1) + 1);
                     __builtin_nontemporal_store(p.z, reinterpret_cast<uint32_t*>(tab + 2 * s + 1) + 2); __builtin_nontemporal_store(p.w, reinterpret_cast<uint32_t*>(tab + 2 * s + 1) + 3); }
  }
  if (MODE == 6) { slot_of[j] = (uint32_t)s; wflag[j] = win ? 3 : 0; }
  else out[j] = x;
}
// pass B: survivors re-read head + pair of their slot; FROM_ARRAY: the slot comes from slot_of[] (coalesced), else it is recomputed
template <bool FROM_ARRAY>
__global__ __launch_bounds__(256) void k_passB(const uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t epoch, const uint32_t* slot_of, uint8_t* wflag, uint32_t* blk) {
  uint32_t j = blockIdx.x * 256u + threadIdx.x; if (j >= n) return;
  bool win; uint64_t s;
  if (FROM_ARRAY) { win = wflag[j] == 3; s = slot_of[j]; }
  else { win = (mix64(j * 7 + seed) % 100) < 84; s = __umul64hi(mix64(j + seed), nslots); }
  uint32_t w = 0;
  if (win) {
    uint4 lo = tab[2 * s], hi = tab[2 * s + 1];
    w = (lo.w == ((epoch << 24) | j) && hi.x == j) ? 1u : 2u;
  }
  wflag[j] = (uint8_t)w;
  unsigned long long m = __ballot(w == 1);
  if ((threadIdx.x & 63) == 0) atomicAdd(&blk[blockIdx.x], (uint32_t)__popcll(m));   // stands in for the block summary
}
__global__ void k_fill(uint4* t, size_t n16){ size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n16;i+=st){ uint32_t v=(uint32_t)i; t[i]=make_uint4(v,v*3,v*5,v*7);} }
__global__ void k_fillcols(uint64_t* id, uint32_t* f, int64_t* ts, int64_t* val, uint32_t n){ uint32_t j=blockIdx.x*256u+threadIdx.x; if(j<n){ id[j]=mix64(j*31+5); f[j]=77; ts[j]=j; val[j]=j*3; } }

template<class F> float timeN(F f, int iters, int reps){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); std::vector<float> ts;
  for(int i=0;i<iters;i++){ CK(hipEventRecord(e0)); for(int r=0;r<reps;r++) f(i*reps+r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ts.push_back(ms/reps);}
  std::sort(ts.begin(),ts.end()); return ts[ts.size()/2]*1000.f; }
#define SEED(i) ((uint64_t)(i)*1315423911ull+17)

int main(int argc, char** argv) {
  uint32_t n = 1000000;
  uint32_t* out; CK(hipMalloc(&out, (size_t)n * 4));
  uint64_t* id; uint32_t* field; int64_t* ts; int64_t* val; uint32_t* slot_of; uint8_t* wflag; uint32_t* blk;
  CK(hipMalloc(&id, (size_t)n * 8)); CK(hipMalloc(&field, (size_t)n * 4)); CK(hipMalloc(&ts, (size_t)n * 8)); CK(hipMalloc(&val, (size_t)n * 8));
  CK(hipMalloc(&slot_of, (size_t)n * 4)); CK(hipMalloc(&wflag, n)); CK(hipMalloc(&blk, 4096 * 4)); CK(hipMemset(blk, 0, 4096 * 4));
  const int g = (n + 255) / 256;
  hipLaunchKernelGGL(k_fillcols, dim3(g), dim3(256), 0, 0, id, field, ts, val, n);
  const uint64_t sizes_mb[4] = {400, 640, 1408, 4096};
  for (int si = 0; si < 4; si++) {
    size_t bytes = sizes_mb[si] << 20; uint64_t nslots = bytes / 32;
    uint4* tab; CK(hipMalloc(&tab, bytes));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, tab, bytes / 16); CK(hipDeviceSynchronize());
    const int IT = 7, R = 10;
#define RUNA(M) timeN([&](int i){ hipLaunchKernelGGL((k_passA<M>), dim3(g), dim3(256), 0, 0, tab, nslots, n, SEED(i), (uint32_t)(i & 255), out, id, field, ts, val, slot_of, wflag); }, IT, R)
    float a0 = RUNA(0), a1 = RUNA(1), a2 = RUNA(2), a3 = RUNA(3), a4 = RUNA(4), a5 = RUNA(5), a6 = RUNA(6);
    // pass A (mode 6 / mode 2) followed by pass B on the same slots: time of the pair, and of B alone by difference
    float ab = timeN([&](int i){ hipLaunchKernelGGL((k_passA<6>), dim3(g), dim3(256), 0, 0, tab, nslots, n, SEED(i), (uint32_t)(i & 255), out, id, field, ts, val, slot_of, wflag);
                                 hipLaunchKernelGGL((k_passB<true>), dim3(g), dim3(256), 0, 0, (const uint4*)tab, nslots, n, SEED(i), (uint32_t)(i & 255), (const uint32_t*)slot_of, wflag, blk); }, IT, R);
    float ab2 = timeN([&](int i){ hipLaunchKernelGGL((k_passA<2>), dim3(g), dim3(256), 0, 0, tab, nslots, n, SEED(i), (uint32_t)(i & 255), out, id, field, ts, val, slot_of, wflag);
                                  hipLaunchKernelGGL((k_passB<false>), dim3(g), dim3(256), 0, 0, (const uint4*)tab, nslots, n, SEED(i), (uint32_t)(i & 255), (const uint32_t*)slot_of, wflag, blk); }, IT, R);
    // pass B alone on slots last written long ago (cold: not in the Infinity Cache)
    float bcold = timeN([&](int i){ hipLaunchKernelGGL((k_passB<false>), dim3(g), dim3(256), 0, 0, (const uint4*)tab, nslots, n, SEED(i + 1000), (uint32_t)(i & 255), (const uint32_t*)slot_of, wflag, blk); }, IT, R);
    printf("table %5llu MB | A: read %.1f | +pair %.1f | +head+pair %.1f | +slot32 %.1f | +xchg+pair %.1f | nt head+pair %.1f | full passA(cols,slot_of) %.1f || A6+B %.1f | A2+B %.1f | B cold %.1f  us per 1M\n",
           (unsigned long long)sizes_mb[si], a0, a1, a2, a3, a4, a5, a6, ab, ab2, bcold);
    fflush(stdout);
    CK(hipFree(tab));
  }
  return 0;
}
