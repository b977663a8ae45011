// Where does the streaming merge of the view patch lose its bandwidth? Variants of one pass over three columns (int32 value, u32 position, u64 id) of N rows:
//   copy16    : 16-byte loads and stores per lane, everything aligned (the ceiling for this read/write mix)
//   copy16s   : the same, destination shifted by S elements (unaligned multi-dword stores)
//   copy4     : one element per lane and instruction (4 / 4 / 8 bytes)
//   merge     : k_view_merge itself with an EMPTY patch (no deleted index, no inserted key: a plain copy through the kernel's whole prolog), and with a patch
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../include -o view_merge_micro view_merge_micro.hip      run: ./view_merge_micro [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../bullet-js_amd/csrc/view_kernels.h"
using namespace bmx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int iv4 __attribute__((ext_vector_type(4)));
typedef unsigned uv4 __attribute__((ext_vector_type(4)));
typedef unsigned long long lv2 __attribute__((ext_vector_type(2)));
template <class V, class P> __device__ __forceinline__ void stv(P* d, const V& v) { __builtin_memcpy(d, &v, sizeof(V)); }
__global__ __launch_bounds__(256) void k_copy16(const int* v, const unsigned* p, const unsigned long long* id, int* zv, unsigned* zp, unsigned long long* zi, size_t n, unsigned shift, int nt) {
  for (size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; e + 4 <= n; e += (size_t)gridDim.x * 1024) {
    iv4 a; uv4 b; lv2 c, d;
    if (nt) { a = __builtin_nontemporal_load((const iv4*)(v + e)); b = __builtin_nontemporal_load((const uv4*)(p + e)); c = __builtin_nontemporal_load((const lv2*)(id + e)); d = __builtin_nontemporal_load((const lv2*)(id + e + 2)); }
    else { a = *(const iv4*)(v + e); b = *(const uv4*)(p + e); c = *(const lv2*)(id + e); d = *(const lv2*)(id + e + 2); }
    stv(zv + e + shift, a); stv(zp + e + shift, b); stv(zi + e + shift, c); stv(zi + e + shift + 2, d);
  }
}
__global__ __launch_bounds__(256) void k_copy4(const int* v, const unsigned* p, const unsigned long long* id, int* zv, unsigned* zp, unsigned long long* zi, size_t n, unsigned shift) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) { zv[e + shift] = v[e]; zp[e + shift] = p[e]; zi[e + shift] = id[e]; }
}
int main(int argc, char** argv) {
  const size_t N = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
  const size_t cap = N + N / 8 + 4096;
  int *v, *zv; unsigned *p, *zp; unsigned long long *id, *zi;
  CK(hipMalloc(&v, cap * 4)); CK(hipMalloc(&zv, cap * 4)); CK(hipMalloc(&p, cap * 4)); CK(hipMalloc(&zp, cap * 4)); CK(hipMalloc(&id, cap * 8)); CK(hipMalloc(&zi, cap * 8));
  // a sorted view: value = i / 100000, position = i (keys unique and ascending)
  { std::vector<int> hv(N); std::vector<unsigned> hp(N); for (size_t i = 0; i < N; i++) { hv[i] = (int)(i / 100000); hp[i] = (unsigned)i; }
    CK(hipMemcpy(v, hv.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(p, hp.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(id, 1, N * 8)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best = std::min(best, ms); }
    printf("%-34s %8.1f us  %6.2f TB/s (16 B read + 16 B written per row)\n", name, best * 1e3, 32.0 * N / (best * 1e-3) / 1e12); fflush(stdout);
  };
  const unsigned G = (unsigned)std::min<size_t>((N / 4 + 255) / 256, 65536 * 4);
  timeit("copy16 aligned", [&] { hipLaunchKernelGGL(k_copy16, dim3(G), dim3(256), 0, 0, v, p, id, zv, zp, zi, N, 0u, 0); });
  timeit("copy16 aligned, nontemporal loads", [&] { hipLaunchKernelGGL(k_copy16, dim3(G), dim3(256), 0, 0, v, p, id, zv, zp, zi, N, 0u, 1); });
  timeit("copy16 shifted by 1", [&] { hipLaunchKernelGGL(k_copy16, dim3(G), dim3(256), 0, 0, v, p, id, zv, zp, zi, N, 1u, 1); });
  timeit("copy16 shifted by 2", [&] { hipLaunchKernelGGL(k_copy16, dim3(G), dim3(256), 0, 0, v, p, id, zv, zp, zi, N, 2u, 1); });
  timeit("copy16, grid = one group per lane", [&] { hipLaunchKernelGGL(k_copy16, dim3((unsigned)((N / 4 + 255) / 256)), dim3(256), 0, 0, v, p, id, zv, zp, zi, N, 0u, 1); });
  timeit("copy4", [&] { hipLaunchKernelGGL(k_copy4, dim3(65536), dim3(256), 0, 0, v, p, id, zv, zp, zi, N, 0u); });
  // the merge kernel with an empty patch, then with 2 % of the keys deleted and 1 % inserted
  const uint32_t nt = (uint32_t)((N + VIEW_TILE - 1) / VIEW_TILE);
  int* sv; unsigned *sp, *d0, *y0; int *dv, *yv; unsigned *dp, *yp;
  const uint32_t M = (uint32_t)(N / 100);
  CK(hipMalloc(&sv, (nt + 1) * 4)); CK(hipMalloc(&sp, (nt + 1) * 4)); CK(hipMalloc(&d0, (nt + 1) * 4)); CK(hipMalloc(&y0, (nt + 1) * 4));
  CK(hipMalloc(&dv, (2 * (size_t)M + 1) * 4)); CK(hipMalloc(&dp, (2 * (size_t)M + 1) * 4)); CK(hipMalloc(&yv, (M + 1) * 4)); CK(hipMalloc(&yp, (M + 1) * 4));
  unsigned* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  ViewRun<int> X{v, p, (uint64_t*)id}, Z{zv, zp, (uint64_t*)zi};
  hipLaunchKernelGGL((k_view_sample<int>), dim3((nt + 255) / 256), dim3(256), 0, 0, (const int*)v, (const unsigned*)p, nt, sv, sp);
  hipLaunchKernelGGL((k_view_tile_offsets<int>), dim3((nt + 256) / 256), dim3(256), 0, 0, (const int*)sv, (const unsigned*)sp, nt, (const int*)dv, (const unsigned*)dp, 0u, (const int*)yv, (const unsigned*)yp, 0u, d0, y0);
  CK(hipDeviceSynchronize());
  timeit("k_view_merge, empty patch", [&] { hipLaunchKernelGGL((k_view_merge<int>), dim3(nt), dim3(256), 0, 0, X, (uint32_t)N, (const int*)dv, (const unsigned*)dp, (const int*)yv, (const unsigned*)yp, (const uint64_t*)id, Z, (const unsigned*)d0, (const unsigned*)y0, err); });
  { // delete the keys at i = 100k and i = 100k + 50, insert (value(100k + 50), 100k + 50) again: keys stay unique, 2 % deleted, 1 % inserted
    std::vector<int> hdv(2 * (size_t)M), hyv(M); std::vector<unsigned> hdp(2 * (size_t)M), hyp(M);
    for (uint32_t k = 0; k < M; k++) {
      const size_t a = (size_t)k * 100, b = a + 50;
      hdv[2 * k] = (int)(a / 100000); hdp[2 * k] = (unsigned)a; hdv[2 * k + 1] = (int)(b / 100000); hdp[2 * k + 1] = (unsigned)b; hyv[k] = (int)(b / 100000); hyp[k] = (unsigned)b;
    }
    CK(hipMemcpy(dv, hdv.data(), 2 * (size_t)M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, hdp.data(), 2 * (size_t)M * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(yv, hyv.data(), M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(yp, hyp.data(), M * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_view_tile_offsets<int>), dim3((nt + 256) / 256), dim3(256), 0, 0, (const int*)sv, (const unsigned*)sp, nt, (const int*)dv, (const unsigned*)dp, 2 * M, (const int*)yv, (const unsigned*)yp, M, d0, y0);
    CK(hipDeviceSynchronize());
    timeit("k_view_merge, 2 % deleted 1 % inserted", [&] { hipLaunchKernelGGL((k_view_merge<int>), dim3(nt), dim3(256), 0, 0, X, (uint32_t)N, (const int*)dv, (const unsigned*)dp, (const int*)yv, (const unsigned*)yp, (const uint64_t*)id, Z, (const unsigned*)d0, (const unsigned*)y0, err); });
    unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    // check: Z must be X with keys 100k removed (100k + 50 removed and inserted again)
    std::vector<unsigned> hz(N - M); CK(hipMemcpy(hz.data(), zp, (N - M) * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, q = 0; for (size_t i = 0; i < N; i++) { if (i % 100 == 0) continue; if (hz[q] != (unsigned)i) bad++; q++; }
    printf("err flag %u, %zu of %zu output positions wrong\n", herr, bad, q);
  }
  { // the change run's sort: 2M random keys (value in [0, 1000), position random and unique), tile sort + merge passes; per-phase switches
    const uint32_t K = 2000000;
    std::vector<int> kv(K); std::vector<unsigned> kp(K);
    unsigned long long x = 88172645463325252ull;
    for (uint32_t i = 0; i < K; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; kv[i] = (int)(x % 1000); kp[i] = (unsigned)((x >> 20) % 100000000u); }
    int *a, *b; unsigned *ap, *bp;
    CK(hipMalloc(&a, (K + 4096) * 4)); CK(hipMalloc(&b, (K + 4096) * 4)); CK(hipMalloc(&ap, (K + 4096) * 4)); CK(hipMalloc(&bp, (K + 4096) * 4));
    CK(hipMemcpy(a, kv.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ap, kp.data(), K * 4, hipMemcpyHostToDevice));
    ViewSegs S{}; S.base[0] = 0; S.len[0] = K / 2; S.base[1] = K / 2; S.len[1] = K - K / 2;
    const uint32_t t0b = (S.len[0] + VIEW_SORT_TILE - 1) / VIEW_SORT_TILE, t1b = (S.len[1] + VIEW_SORT_TILE - 1) / VIEW_SORT_TILE;
    S.blk0[0] = 0; S.blk0[1] = t0b; S.blk0[2] = t0b + t1b;
    timeit("tile sort, 2 x 1M keys", [&] { hipLaunchKernelGGL((k_view_tile_sort<int>), dim3(S.blk0[2]), dim3(VIEW_SORT_THREADS), 0, 0, (const int*)a, (const unsigned*)ap, b, bp, S); });
    ViewSegs P = S; P.blk0[1] = (S.len[0] + VIEW_PASS_KEYS - 1) / VIEW_PASS_KEYS; P.blk0[2] = P.blk0[1] + (S.len[1] + VIEW_PASS_KEYS - 1) / VIEW_PASS_KEYS;
    for (uint32_t dbg = 0; dbg < 4; dbg++) for (uint32_t L : {2048u, 65536u, 524288u}) {
      P.dbg = dbg; char nm[96]; snprintf(nm, sizeof nm, "merge pass L = %u, switches %u", L, dbg);
      timeit(nm, [&] { hipLaunchKernelGGL((k_view_merge_pass<int>), dim3(P.blk0[2]), dim3(256), 0, 0, (const int*)b, (const unsigned*)bp, a, ap, P, L); });
    }
    // the whole sort, checked
    CK(hipMemcpy(a, kv.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ap, kp.data(), K * 4, hipMemcpyHostToDevice));
    P.dbg = 0;
    int* cv[2] = {a, b}; unsigned* cp[2] = {ap, bp};
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_view_tile_sort<int>), dim3(S.blk0[2]), dim3(VIEW_SORT_THREADS), 0, 0, (const int*)cv[0], (const unsigned*)cp[0], cv[1], cp[1], S);
    int cur = 1;
    for (uint32_t L = VIEW_SORT_TILE; L < S.len[1]; L *= 2) { hipLaunchKernelGGL((k_view_merge_pass<int>), dim3(P.blk0[2]), dim3(256), 0, 0, (const int*)cv[cur], (const unsigned*)cp[cur], cv[cur ^ 1], cp[cur ^ 1], P, L); cur ^= 1; }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<int> rv(K); std::vector<unsigned> rp(K); CK(hipMemcpy(rv.data(), cv[cur], K * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(rp.data(), cp[cur], K * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (uint32_t s2 = 0; s2 < 2; s2++) for (uint32_t i = S.base[s2] + 1; i < S.base[s2] + S.len[s2]; i++) if (rv[i - 1] > rv[i] || (rv[i - 1] == rv[i] && rp[i - 1] > rp[i])) bad++;
    printf("whole sort of 2 x 1M keys: %.1f us, %zu inversions\n", ms * 1e3, bad);
  }
  return 0;
}
