// Microbenchmark v5: do instruction-level cache-policy bits (sc0/sc1/nt) change the request size / rate of random 32-B probes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }
template<int MODE> __device__ __forceinline__ u32x4 ld16(const void* p){
  u32x4 v;
  if(MODE==0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if(MODE==1) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if(MODE==2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if(MODE==3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if(MODE==4) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if(MODE==5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template<int MODE> __global__ void k_rd(const uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  u32x4 a = ld16<MODE>(tab+2*s+1);
  out[j] = a.x^a.y^a.z^a.w;
}
__global__ void k_fill(uint4* t, size_t n16){ size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n16;i+=st){ uint32_t v=(uint32_t)i; t[i]=make_uint4(v,v*3,v*5,v*7);} }
template<class F> float timeN(F f, int iters, int reps){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); std::vector<float> ts;
  for(int i=0;i<iters;i++){ CK(hipEventRecord(e0)); for(int r=0;r<reps;r++) f(i*reps+r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ts.push_back(ms/reps);} 
  std::sort(ts.begin(),ts.end()); return ts[ts.size()/2]*1000.f; }
#define SEED(i) ((uint64_t)(i)*1315423911ull+17)
int main(){
  uint32_t n = 1u<<20; size_t bytes = 1024ull<<20; uint64_t nslots=bytes/32;
  uint32_t* out; CK(hipMalloc(&out,(size_t)n*4));
  uint4* tab; CK(hipMalloc(&tab,bytes));
  hipLaunchKernelGGL(k_fill,dim3(2048),dim3(256),0,0,tab,bytes/16); CK(hipDeviceSynchronize());
  const int IT=5,R=10; int g=n/256;
  float t[6];
  t[0]=timeN([&](int i){ hipLaunchKernelGGL(k_rd<0>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  t[1]=timeN([&](int i){ hipLaunchKernelGGL(k_rd<1>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  t[2]=timeN([&](int i){ hipLaunchKernelGGL(k_rd<2>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  t[3]=timeN([&](int i){ hipLaunchKernelGGL(k_rd<3>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  t[4]=timeN([&](int i){ hipLaunchKernelGGL(k_rd<4>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  t[5]=timeN([&](int i){ hipLaunchKernelGGL(k_rd<5>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  printf("random 16-B loads, 1 GiB table | plain %.1f | sc0 %.1f | sc1 %.1f | sc0 sc1 %.1f | nt %.1f | sc0 sc1 nt %.1f us per 1M\n",t[0],t[1],t[2],t[3],t[4],t[5]);
  return 0;
}
