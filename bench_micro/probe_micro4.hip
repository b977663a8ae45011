// Microbenchmark v4: atomic scope (agent vs workgroup vs wavefront) on random slots; read+xchg+store with each scope.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }
template<int SCOPE> __device__ __forceinline__ uint32_t xchg(uint32_t* p, uint32_t v){
  if(SCOPE==0) return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if(SCOPE==1) return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if(SCOPE==2) return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template<int SCOPE> __global__ void k_xchg(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  out[j]=xchg<SCOPE>((uint32_t*)(tab+2*s)+3, j);
}
template<int SCOPE> __global__ void k_rd_xchg_st(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint4 a = tab[2*s], b = tab[2*s+1];
  uint32_t x = a.x^a.y^a.z^a.w^b.x^b.y^b.z^b.w;
  if((mix64(j*7+seed)&3)!=0){ uint32_t* w = (uint32_t*)(tab+2*s)+3; x ^= xchg<SCOPE>(w, j); b.x=j; b.y=x; tab[2*s+1]=b; }
  out[j]=x;
}
__global__ void k_fill(uint4* t, size_t n16){ size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n16;i+=st){ uint32_t v=(uint32_t)i; t[i]=make_uint4(v,v*3,v*5,v*7);} }
template<class F> float timeN(F f, int iters, int reps){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); std::vector<float> ts;
  for(int i=0;i<iters;i++){ CK(hipEventRecord(e0)); for(int r=0;r<reps;r++) f(i*reps+r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ts.push_back(ms/reps);} 
  std::sort(ts.begin(),ts.end()); return ts[ts.size()/2]*1000.f; }
#define SEED(i) ((uint64_t)(i)*1315423911ull+17)
int main(){
  uint32_t n = 1u<<20; size_t bytes = 512ull<<20; uint64_t nslots=bytes/32;
  uint32_t* out; CK(hipMalloc(&out,(size_t)n*4));
  uint4* tab; CK(hipMalloc(&tab,bytes));
  hipLaunchKernelGGL(k_fill,dim3(2048),dim3(256),0,0,tab,bytes/16); CK(hipDeviceSynchronize());
  const int IT=7,R=10; int g=n/256;
  float a0=timeN([&](int i){ hipLaunchKernelGGL(k_xchg<0>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  float a1=timeN([&](int i){ hipLaunchKernelGGL(k_xchg<1>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  float a2=timeN([&](int i){ hipLaunchKernelGGL(k_xchg<2>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  float a3=timeN([&](int i){ hipLaunchKernelGGL(k_xchg<3>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  printf("xchg only      | agent %.1f | workgroup %.1f | wavefront %.1f | system %.1f us per 1M\n",a0,a1,a2,a3);
  float b0=timeN([&](int i){ hipLaunchKernelGGL(k_rd_xchg_st<0>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  float b1=timeN([&](int i){ hipLaunchKernelGGL(k_rd_xchg_st<1>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  float b2=timeN([&](int i){ hipLaunchKernelGGL(k_rd_xchg_st<2>,dim3(g),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
  printf("rd+xchg+st 75%% | agent %.1f | workgroup %.1f | wavefront %.1f us per 1M\n",b0,b1,b2);
  return 0;
}
