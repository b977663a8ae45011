// Microbenchmark: random 32-B slot reads through the SCALAR data path (s_load_dwordx8 of a readlane'd address, 8 in flight per wave) against the
// vector path (two global_load_dwordx4 per lane). Question: does the scalar cache path sustain more outstanding misses per CU than the vector L1?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
typedef uint32_t u8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }
__global__ __launch_bounds__(256) void k_vec(const uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint4 a = tab[2*s], b = tab[2*s+1];
  out[j] = a.x^a.y^a.z^a.w^b.x^b.y^b.z^b.w;
}
template<int INFLIGHT>
__global__ __launch_bounds__(256) void k_sca(const uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint64_t addr = (uint64_t)(tab + 2*s);
  uint32_t lo = (uint32_t)addr, hi = (uint32_t)(addr>>32);
  uint32_t acc = 0;
#pragma unroll
  for (int base = 0; base < 64; base += INFLIGHT) {
    uint32_t r[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++) {
      uint32_t alo = __builtin_amdgcn_readlane(lo, base+k), ahi = __builtin_amdgcn_readlane(hi, base+k);
      const u8v __attribute__((address_space(4)))* q = (const u8v __attribute__((address_space(4)))*)(((uint64_t)ahi<<32)|alo);
      u8v v = *q;
      r[k] = v[0]^v[1]^v[2]^v[3]^v[4]^v[5]^v[6]^v[7];
    }
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++) acc = ((threadIdx.x & 63u) == (uint32_t)(base+k)) ? r[k] : acc;
  }
  if (j < n) out[j] = acc;
}
__global__ void k_fill(uint4* t, size_t n16){ size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n16;i+=st){ uint32_t v=(uint32_t)i; t[i]=make_uint4(v,v*3,v*5,v*7);} }
template<class F> float timeN(F f, int iters, int reps){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); std::vector<float> ts;
  for(int i=0;i<iters;i++){ CK(hipEventRecord(e0)); for(int r=0;r<reps;r++) f(i*reps+r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ts.push_back(ms/reps);}
  std::sort(ts.begin(),ts.end()); return ts[ts.size()/2]*1000.f; }
#define SEED(i) ((uint64_t)(i)*1315423911ull+17)
int main(){
  uint32_t n = 1u<<20; size_t bytes = 1408ull<<20; uint64_t nslots=bytes/32;
  uint32_t *out, *out2; CK(hipMalloc(&out,(size_t)n*4)); CK(hipMalloc(&out2,(size_t)n*4));
  uint4* tab; CK(hipMalloc(&tab,bytes));
  hipLaunchKernelGGL(k_fill,dim3(2048),dim3(256),0,0,tab,bytes/16); CK(hipDeviceSynchronize());
  // same results?
  hipLaunchKernelGGL(k_vec,dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(1),out);
  hipLaunchKernelGGL(k_sca<8>,dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(1),out2); CK(hipDeviceSynchronize());
  std::vector<uint32_t> a(n), b(n); CK(hipMemcpy(a.data(),out,n*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(),out2,n*4,hipMemcpyDeviceToHost));
  size_t bad=0; for(uint32_t i=0;i<n;i++) bad += a[i]!=b[i];
  const int IT=5,R=10; int g=n/256;
  float tv=timeN([&](int i){ hipLaunchKernelGGL(k_vec,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  float t4=timeN([&](int i){ hipLaunchKernelGGL(k_sca<4>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  float t8=timeN([&](int i){ hipLaunchKernelGGL(k_sca<8>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  printf("random 32-B slot reads, 1.4 GB table, 1M per launch | vector path %.1f us | scalar path, 4 in flight per wave %.1f us | 8 in flight %.1f us | mismatches %zu\n",tv,t4,t8,bad);
  return 0;
}
