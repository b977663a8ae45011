// h2d_threads.hip — pageable host -> device staging of a 1M-delta batch (id 8 MB, field 4 MB, ts 8 MB, val 8 MB):
// one thread issuing four hipMemcpyAsync on one stream vs one thread per column on its own stream.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
  const size_t n = 1 << 20; const size_t sz[4] = {n * 8, n * 4, n * 8, n * 8};
  void* h[4]; void* d[4]; hipStream_t st[4];
  for (int i = 0; i < 4; i++) { h[i] = malloc(sz[i]); memset(h[i], i + 1, sz[i]); CK(hipMalloc(&d[i], sz[i])); CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); }
  auto now = [] { return std::chrono::steady_clock::now(); };
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = now();
    for (int it = 0; it < 20; it++) { for (int i = 0; i < 4; i++) CK(hipMemcpyAsync(d[i], h[i], sz[i], hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); }
    double a = std::chrono::duration<double, std::micro>(now() - t0).count() / 20;
    t0 = now();
    for (int it = 0; it < 20; it++) {
      std::vector<std::thread> th;
      for (int i = 0; i < 4; i++) th.emplace_back([&, i] { hipSetDevice(0); hipMemcpyAsync(d[i], h[i], sz[i], hipMemcpyHostToDevice, st[i]); hipStreamSynchronize(st[i]); });
      for (auto& t : th) t.join();
    }
    double b = std::chrono::duration<double, std::micro>(now() - t0).count() / 20;
    printf("28 MB pageable H2D: one thread %.0f us (%.1f GB/s) | four threads %.0f us (%.1f GB/s)\n", a, 29.36e6 / a / 1e3, b, 29.36e6 / b / 1e3);
  }
  // pinned reference
  void* p[4]; for (int i = 0; i < 4; i++) { CK(hipHostMalloc(&p[i], sz[i])); memcpy(p[i], h[i], sz[i]); }
  auto t0 = now();
  for (int it = 0; it < 20; it++) { for (int i = 0; i < 4; i++) CK(hipMemcpyAsync(d[i], p[i], sz[i], hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); }
  double c = std::chrono::duration<double, std::micro>(now() - t0).count() / 20;
  printf("pinned: %.0f us (%.1f GB/s)\n", c, 29.36e6 / c / 1e3);
  return 0;
}
