// Microbenchmark: random-probe throughput on MI355X for candidate slot layouts.
// Informs DESIGN.md's choice of resident-table layout (AoS 32-B slot vs SoA columns,
// table size vs the 256 MiB Infinity Cache). Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__device__ __forceinline__ uint64_t mix64(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }

// A: one lane reads one 32-B slot (2 x 16 B)
__global__ void k_aos32(const uint4* __restrict__ tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint4 a = tab[2*s], b = tab[2*s+1];
  out[j] = a.x^a.y^a.z^a.w^b.x^b.y^b.z^b.w;
}
// B: one lane reads 16 B of a 32-B slot
__global__ void k_aos16(const uint4* __restrict__ tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint4 a = tab[2*s];
  out[j] = a.x^a.y^a.z^a.w;
}
// C: 4 lanes read one 128-B line (bucket of 4 slots)
__global__ void k_line128(const uint4* __restrict__ tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t t = blockIdx.x*blockDim.x+threadIdx.x; uint32_t j = t>>2; if(j>=n) return;
  uint64_t b = __umul64hi(mix64(j+seed), nslots>>2);
  uint64_t s = b*4 + (t&3);
  uint4 a = tab[2*s], c = tab[2*s+1];
  uint32_t v = a.x^a.y^a.z^a.w^c.x^c.y^c.z^c.w;
  v ^= __shfl_xor(v,1); v ^= __shfl_xor(v,2);
  if((t&3)==0) out[j]=v;
}
// D: SoA columns id(8) field(4) ts(8) val(8) at the same random index
__global__ void k_soa(const uint64_t* __restrict__ id, const uint32_t* __restrict__ fld, const int64_t* __restrict__ ts, const int64_t* __restrict__ val,
                      uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint64_t a=id[s]; uint32_t f=fld[s]; int64_t t=ts[s], v=val[s];
  out[j] = (uint32_t)(a^(a>>32))^f^(uint32_t)(t^(t>>32))^(uint32_t)(v^(v>>32));
}
// E: A + scattered 16-B store into the second half of the slot when a predicate holds (~75%)
__global__ void k_aos32_rw(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint4 a = tab[2*s], b = tab[2*s+1];
  uint32_t x = a.x^a.y^a.z^a.w^b.x^b.y^b.z^b.w;
  if((mix64(j*7+seed)&3)!=0){ b.x = j; b.y=(uint32_t)seed; tab[2*s+1]=b; }
  out[j]=x;
}
// F: A + atomicExch on a 4-B word of the slot for ~75% (claim/link step)
__global__ void k_aos32_atom(uint4* tab, uint64_t nslots, uint32_t n, uint64_t seed, uint32_t* out){
  uint32_t j = blockIdx.x*blockDim.x+threadIdx.x; if(j>=n) return;
  uint64_t s = __umul64hi(mix64(j+seed), nslots);
  uint4 a = tab[2*s], b = tab[2*s+1];
  uint32_t x = a.x^a.y^a.z^a.w^b.x^b.y^b.z^b.w;
  if((mix64(j*7+seed)&3)!=0){ uint32_t* w = (uint32_t*)(tab+2*s)+3; x ^= atomicExch(w, j); }
  out[j]=x;
}
// G: pure streaming copy for calibration
__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16){
  size_t i = (size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x;
  for(; i<n16; i+=st) out[i]=in[i];
}
__global__ void k_fill(uint4* t, size_t n16){ size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n16;i+=st){ uint32_t v=(uint32_t)i; t[i]=make_uint4(v,v*3,v*5,v*7);} }

template<class F> float timeit(F f, int iters){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for(int i=0;i<iters;i++){ CK(hipEventRecord(e0)); f(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ts.push_back(ms);} 
  std::sort(ts.begin(),ts.end()); return ts[ts.size()/2]*1000.f; // median us
}
int main(int argc,char**argv){
  uint32_t n = argc>1? atoi(argv[1]) : (1u<<20);
  int blk = argc>2? atoi(argv[2]) : 256;
  size_t sizesMB[] = {64,128,256,320,400,512,640,1024,2048};
  uint32_t* out; CK(hipMalloc(&out, (size_t)n*4));
  printf("n=%u block=%d\n", n, blk);
  // copy calibration
  { size_t B=1ull<<30; uint4 *a,*b; CK(hipMalloc(&a,B)); CK(hipMalloc(&b,B)); hipLaunchKernelGGL(k_fill,dim3(2048),dim3(256),0,0,a,B/16); CK(hipDeviceSynchronize());
    float us=timeit([&](int){ hipLaunchKernelGGL(k_copy,dim3(4096),dim3(256),0,0,a,b,B/16); },10);
    printf("copy 1GiB: %.1f us  => %.2f TB/s (r+w)\n", us, 2.0*B/us/1e6); CK(hipFree(a)); CK(hipFree(b)); }
  { // empty-kernel launch latency
    float us=timeit([&](int){ hipLaunchKernelGGL(k_fill,dim3(1),dim3(64),0,0,(uint4*)out,(size_t)1); },50); printf("tiny kernel event-to-event: %.2f us\n",us);}
  for(size_t mb: sizesMB){
    size_t bytes = mb<<20; uint64_t nslots = bytes/32; uint4* tab; CK(hipMalloc(&tab, bytes));
    hipLaunchKernelGGL(k_fill,dim3(2048),dim3(256),0,0,tab,bytes/16); CK(hipDeviceSynchronize());
    int g=(n+blk-1)/blk; int iters=30;
    float a=timeit([&](int i){ hipLaunchKernelGGL(k_aos32,dim3(g),dim3(blk),0,0,tab,nslots,n,(uint64_t)i*1315423911ull+1,out); },iters);
    float b=timeit([&](int i){ hipLaunchKernelGGL(k_aos16,dim3(g),dim3(blk),0,0,tab,nslots,n,(uint64_t)i*1315423911ull+2,out); },iters);
    float c=timeit([&](int i){ hipLaunchKernelGGL(k_line128,dim3((4*(size_t)n+blk-1)/blk),dim3(blk),0,0,tab,nslots,n,(uint64_t)i*1315423911ull+3,out); },iters);
    // SoA over same total bytes: nslots rows * 28 B ~ use same nslots rows
    const uint64_t* id=(const uint64_t*)tab; const uint32_t* fld=(const uint32_t*)((char*)tab+nslots*8); const int64_t* ts=(const int64_t*)((char*)tab+nslots*12); const int64_t* val=(const int64_t*)((char*)tab+nslots*20);
    float d=timeit([&](int i){ hipLaunchKernelGGL(k_soa,dim3(g),dim3(blk),0,0,id,fld,ts,val,nslots,n,(uint64_t)i*1315423911ull+4,out); },iters);
    float e=timeit([&](int i){ hipLaunchKernelGGL(k_aos32_rw,dim3(g),dim3(blk),0,0,tab,nslots,n,(uint64_t)i*1315423911ull+5,out); },iters);
    float f=timeit([&](int i){ hipLaunchKernelGGL(k_aos32_atom,dim3(g),dim3(blk),0,0,tab,nslots,n,(uint64_t)i*1315423911ull+6,out); },iters);
    printf("table %5zu MB | aos32 %.1f us | aos16 %.1f | line128x4 %.1f | soa4 %.1f | aos32+store %.1f | aos32+atomicExch %.1f\n", mb,a,b,c,d,e,f);
    CK(hipFree(tab));
  }
  return 0;
}
