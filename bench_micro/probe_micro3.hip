// Microbenchmark v3: does the allocation kind (normal / fine-grained / uncached) change the request size and
// the rate of random 16-32 B probes, stores and atomics? Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }
__global__ void k_rd32(const uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint4 a = tab[2*s], b = tab[2*s+1];
  out[j] = a.x^a.y^a.z^a.w^b.x^b.y^b.z^b.w;
}
__global__ void k_rd32_st(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint4 a = tab[2*s], b = tab[2*s+1];
  uint32_t x = a.x^a.y^a.z^a.w^b.x^b.y^b.z^b.w;
  if((mix64(j*7+seed)&3)!=0){ b.x = j; b.y=(uint32_t)seed; tab[2*s+1]=b; }
  out[j]=x;
}
__global__ void k_rd32_xchg_st(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint4 a = tab[2*s], b = tab[2*s+1];
  uint32_t x = a.x^a.y^a.z^a.w^b.x^b.y^b.z^b.w;
  if((mix64(j*7+seed)&3)!=0){ uint32_t* w = (uint32_t*)(tab+2*s)+3; x ^= atomicExch(w, j); b.x=j; b.y=x; tab[2*s+1]=b; }
  out[j]=x;
}
__global__ void k_xchg(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  out[j]=atomicExch((uint32_t*)(tab+2*s)+3, j);
}
__global__ void k_st16(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  tab[2*s+1]=make_uint4(j,(uint32_t)seed,j*3,7);
}
__global__ void k_fill(uint4* t, size_t n16){ size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n16;i+=st){ uint32_t v=(uint32_t)i; t[i]=make_uint4(v,v*3,v*5,v*7);} }
template<class F> float timeN(F f, int iters, int reps){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); std::vector<float> ts;
  for(int i=0;i<iters;i++){ CK(hipEventRecord(e0)); for(int r=0;r<reps;r++) f(i*reps+r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ts.push_back(ms/reps);} 
  std::sort(ts.begin(),ts.end()); return ts[ts.size()/2]*1000.f; }
#define SEED(i) ((uint64_t)(i)*1315423911ull+17)
int main(int argc,char**argv){
  uint32_t n = 1u<<20; size_t bytes = 512ull<<20; uint64_t nslots=bytes/32;
  uint32_t* out; CK(hipMalloc(&out,(size_t)n*4));
  const char* names[3]={"hipMalloc","finegrained","uncached"};
  for(int kind=0;kind<3;kind++){
    uint4* tab=nullptr; hipError_t e;
    if(kind==0) e=hipMalloc(&tab,bytes); else e=hipExtMallocWithFlags((void**)&tab,bytes, kind==1?hipDeviceMallocFinegrained:hipDeviceMallocUncached);
    if(e!=hipSuccess){ printf("%s: alloc failed: %s\n",names[kind],hipGetErrorString(e)); continue; }
    hipLaunchKernelGGL(k_fill,dim3(2048),dim3(256),0,0,tab,bytes/16); CK(hipDeviceSynchronize());
    const int IT=7,R=10; int g=n/256;
    float a=timeN([&](int i){ hipLaunchKernelGGL(k_rd32,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float b=timeN([&](int i){ hipLaunchKernelGGL(k_rd32_st,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float c=timeN([&](int i){ hipLaunchKernelGGL(k_rd32_xchg_st,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float d=timeN([&](int i){ hipLaunchKernelGGL(k_xchg,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float f=timeN([&](int i){ hipLaunchKernelGGL(k_st16,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    printf("%-12s 512MB | rd32 %.1f | rd32+st75 %.1f | rd32+xchg+st75 %.1f | xchg %.1f | st16 %.1f us per 1M\n",names[kind],a,b,c,d,f);
    CK(hipFree(tab));
  }
  return 0;
}
