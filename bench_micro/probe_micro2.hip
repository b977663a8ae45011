// Microbenchmark v2: is the random-probe rate latency- or request-rate-bound? Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }

template<int ILP, bool NT>
__global__ void k_ilp(const uint4* __restrict__ tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t t = blockIdx.x*blockDim.x+threadIdx.x; uint32_t nt = gridDim.x*blockDim.x;
  uint32_t acc=0;
  for(uint32_t base=t; base<n; base+= nt*ILP){
    uint4 a[ILP];
    #pragma unroll
    for(int k=0;k<ILP;k++){ uint32_t j=base+k*nt; uint64_t s = __umul64hi(mix64((j<n?j:0)+seed), nslots);
      if(NT) { const uint4* p=&tab[2*s+1]; a[k].x=__builtin_nontemporal_load(&p->x); a[k].y=__builtin_nontemporal_load(&p->y); a[k].z=__builtin_nontemporal_load(&p->z); a[k].w=__builtin_nontemporal_load(&p->w);} else a[k]=tab[2*s+1]; }
    #pragma unroll
    for(int k=0;k<ILP;k++) acc ^= a[k].x^a[k].y^a[k].z^a[k].w;
  }
  out[t]=acc;
}
// sorted: lane j probes slot floor(j*nslots/n)+jitter -> monotone addresses, 1 probe per ~nslots/n slots
__global__ void k_sorted(const uint4* __restrict__ tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t stride = nslots/n; uint64_t s = (uint64_t)j*stride + (mix64(j+seed) % stride);
  uint4 a = tab[2*s+1]; out[j]=a.x^a.y^a.z^a.w;
}
// store variants
template<int MODE> // 0: 16B store, 1: 32B store (both halves), 2: atomicMax no return (u32), 3: atomicExch with return, 4: 16B nontemporal store
__global__ void k_wr(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint4 v = make_uint4(j,(uint32_t)seed,j*3,7);
  if(MODE==0) tab[2*s+1]=v;
  else if(MODE==1){ tab[2*s]=v; tab[2*s+1]=v; }
  else if(MODE==2){ atomicMax((uint32_t*)(tab+2*s)+3, j); }
  else if(MODE==3){ out[j]=atomicExch((uint32_t*)(tab+2*s)+3, j); }
  else if(MODE==4){ uint4* p=&tab[2*s+1]; __builtin_nontemporal_store(v.x,&p->x); __builtin_nontemporal_store(v.y,&p->y); __builtin_nontemporal_store(v.z,&p->z); __builtin_nontemporal_store(v.w,&p->w); }
}
// read then (75%) no-return 64-bit atomicMax on ts word
__global__ void k_rd_atom64(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint4 a = tab[2*s], b = tab[2*s+1];
  uint32_t x = a.x^a.y^a.z^a.w^b.x^b.y^b.z^b.w;
  if((mix64(j*7+seed)&3)!=0){ atomicMax((unsigned long long*)(tab+2*s+1), (unsigned long long)j<<20); }
  out[j]=x;
}
__global__ void k_fill(uint4* t, size_t n16){ size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n16;i+=st){ uint32_t v=(uint32_t)i; t[i]=make_uint4(v,v*3,v*5,v*7);} }
__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16){
  size_t i = (size_t)blockIdx.x*blockDim.x+threadIdx.x; if(i<n16) out[i]=in[i]; }
__global__ void k_read(const uint4* __restrict__ in, uint32_t* __restrict__ out, size_t n16){
  size_t i = (size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; uint32_t acc=0;
  for(; i<n16; i+=st){ uint4 a=in[i]; acc^=a.x^a.y^a.z^a.w; } if(acc==0x12345) out[0]=acc; }

template<class F> float timeN(F f, int iters, int reps){ // median over iters of (reps back-to-back launches)/reps, in us
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); std::vector<float> ts;
  for(int i=0;i<iters;i++){ CK(hipEventRecord(e0)); for(int r=0;r<reps;r++) f(i*reps+r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ts.push_back(ms/reps);} 
  std::sort(ts.begin(),ts.end()); return ts[ts.size()/2]*1000.f; }
#define SEED(i) ((uint64_t)(i)*1315423911ull+17)
int main(int argc,char**argv){
  uint32_t n = argc>1? atoi(argv[1]) : (1u<<20);
  uint32_t* out; CK(hipMalloc(&out, (size_t)n*4*2));
  const int R=10, IT=7;
  { size_t B=1ull<<30; uint4 *a,*b; CK(hipMalloc(&a,B)); CK(hipMalloc(&b,B)); hipLaunchKernelGGL(k_fill,dim3(2048),dim3(256),0,0,a,B/16); CK(hipDeviceSynchronize());
    float us=timeN([&](int){ hipLaunchKernelGGL(k_copy,dim3(B/16/256),dim3(256),0,0,a,b,B/16); },5,3);
    printf("copy(1 elem/thread) 1GiB: %.1f us => %.2f TB/s r+w\n", us, 2.0*B/us/1e6);
    us=timeN([&](int){ hipLaunchKernelGGL(k_read,dim3(8192),dim3(256),0,0,a,out,B/16); },5,3);
    printf("read-only 1GiB: %.1f us => %.2f TB/s\n", us, 1.0*B/us/1e6);
    CK(hipFree(a)); CK(hipFree(b)); }
  { float us=timeN([&](int){ hipLaunchKernelGGL(k_fill,dim3(1),dim3(64),0,0,(uint4*)out,(size_t)1); },20,20); printf("tiny kernel back-to-back: %.2f us each\n",us);}
  size_t sizesMB[] = {1,2,4,8,16,32,64,512};
  for(size_t mb: sizesMB){
    size_t bytes = mb<<20; uint64_t nslots = bytes/32; uint4* tab; CK(hipMalloc(&tab, bytes));
    hipLaunchKernelGGL(k_fill,dim3(2048),dim3(256),0,0,tab,bytes/16); CK(hipDeviceSynchronize());
    float t1=timeN([&](int i){ hipLaunchKernelGGL((k_ilp<1,false>),dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float t2=timeN([&](int i){ hipLaunchKernelGGL((k_ilp<2,false>),dim3(n/256/2),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float t4=timeN([&](int i){ hipLaunchKernelGGL((k_ilp<4,false>),dim3(n/256/4),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float t8=timeN([&](int i){ hipLaunchKernelGGL((k_ilp<8,false>),dim3(n/256/8),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float tn=timeN([&](int i){ hipLaunchKernelGGL((k_ilp<1,true>),dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float tn4=timeN([&](int i){ hipLaunchKernelGGL((k_ilp<4,true>),dim3(n/256/4),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float ts=0; if(nslots>=4ull*n) ts=timeN([&](int i){ hipLaunchKernelGGL(k_sorted,dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    printf("tab %4zu MB | rd ilp1 %.1f ilp2 %.1f ilp4 %.1f ilp8 %.1f | nt ilp1 %.1f ilp4 %.1f | sorted %.1f us\n", mb,t1,t2,t4,t8,tn,tn4,ts);
    float w0=timeN([&](int i){ hipLaunchKernelGGL(k_wr<0>,dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float w1=timeN([&](int i){ hipLaunchKernelGGL(k_wr<1>,dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float w2=timeN([&](int i){ hipLaunchKernelGGL(k_wr<2>,dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float w3=timeN([&](int i){ hipLaunchKernelGGL(k_wr<3>,dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float w4=timeN([&](int i){ hipLaunchKernelGGL(k_wr<4>,dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    float w5=timeN([&](int i){ hipLaunchKernelGGL(k_rd_atom64,dim3(n/256),dim3(256),0,0,tab,nslots,n,SEED(i),out); },IT,R);
    printf("            | wr 16B %.1f | wr 32B %.1f | atomicMax32 noret %.1f | atomicExch ret %.1f | wr 16B nt %.1f | rd32+atomicMax64 noret(75%%) %.1f us\n", w0,w1,w2,w3,w4,w5);
    CK(hipFree(tab));
  }
  return 0;
}
