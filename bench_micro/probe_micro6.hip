// Microbenchmark v6: does handling 2 or 4 probes per lane (all loads, then all exchanges, then all stores) shorten the
// read + exchange + store pattern of k_probe_apply at 1M probes, where the launch is only two generations of resident waves?
// Table 704 MB (the bench's 22M slots of 32 B). Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }
template<int ILP>
__global__ __launch_bounds__(256) void k_rxs(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  const uint32_t t = blockIdx.x*256u+threadIdx.x;
  const uint32_t stride = n/ILP;
  if(t>=stride) return;
  uint64_t s[ILP]; uint4 a[ILP], b[ILP]; uint32_t x[ILP]; bool w[ILP];
#pragma unroll
  for(int u=0;u<ILP;u++){ uint32_t j=t+u*stride; s[u]=__umul64hi(mix64(j+seed), nslots); }
#pragma unroll
  for(int u=0;u<ILP;u++){ a[u]=tab[2*s[u]]; b[u]=tab[2*s[u]+1]; }
#pragma unroll
  for(int u=0;u<ILP;u++){ uint32_t j=t+u*stride; x[u]=a[u].x^a[u].y^a[u].z^a[u].w^b[u].x^b[u].y^b[u].z^b[u].w; w[u]=(mix64(j*7+seed)&3)!=0; }
#pragma unroll
  for(int u=0;u<ILP;u++){ if(w[u]) x[u]^=atomicExch((uint32_t*)(tab+2*s[u])+3, t+u*stride); }
#pragma unroll
  for(int u=0;u<ILP;u++){ uint32_t j=t+u*stride; if(w[u]){ b[u].x=j; b[u].y=x[u]; tab[2*s[u]+1]=b[u]; } out[j]=x[u]; }
}
__global__ void k_fill(uint4* t, size_t n16){ size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n16;i+=st){ uint32_t v=(uint32_t)i; t[i]=make_uint4(v,v*3,v*5,v*7);} }
template<class F> float timeN(F f, int iters, int reps){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); std::vector<float> ts;
  for(int i=0;i<iters;i++){ CK(hipEventRecord(e0)); for(int r=0;r<reps;r++) f(i*reps+r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ts.push_back(ms/reps);}
  std::sort(ts.begin(),ts.end()); return ts[ts.size()/2]*1000.f; }
#define SEED(i) ((uint64_t)(i)*1315423911ull+17)
int main(){
  size_t bytes = 704ull<<20; uint64_t nslots=bytes/32;
  uint4* tab; CK(hipMalloc(&tab,bytes));
  uint32_t* out; CK(hipMalloc(&out,(size_t)(4u<<20)*4));
  hipLaunchKernelGGL(k_fill,dim3(2048),dim3(256),0,0,tab,bytes/16); CK(hipDeviceSynchronize());
  for(uint32_t n : {1u<<18, 1u<<20, 1u<<22}){
    const int IT=7,R=10;
    float a=timeN([&](int i){ hipLaunchKernelGGL(k_rxs<1>,dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float b=timeN([&](int i){ hipLaunchKernelGGL(k_rxs<2>,dim3(n/512),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float c=timeN([&](int i){ hipLaunchKernelGGL(k_rxs<4>,dim3(n/1024),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    printf("n %8u | rd+xchg+st75: 1 per lane %.1f us | 2 per lane %.1f us | 4 per lane %.1f us   (per 1M: %.1f / %.1f / %.1f)\n", n, a,b,c, a*1048576.f/n, b*1048576.f/n, c*1048576.f/n);
  }
  return 0;
}
