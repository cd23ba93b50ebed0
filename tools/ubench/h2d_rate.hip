// Aggregate host-to-device copy rate of T threads, each copying its own pinned 3.8 MB block (the size of a wire batch of 32 768 reads)
// to the device on its own stream, back to back: the ceiling the extension's H2D traffic runs against.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/h2d_rate.hip -o /tmp/h2d_rate -lpthread && /tmp/h2d_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <thread>
#include <vector>

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) : 3800000;
  const int reps = 200;
  for (int T : {1, 2, 4, 8, 12, 16, 20}) {
    std::vector<void*> h(T), d(T);
    std::vector<hipStream_t> s(T);
    for (int t = 0; t < T; ++t) {
      if (hipHostMalloc(&h[t], bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&d[t], bytes) != hipSuccess) return 1;
      (void)hipStreamCreateWithFlags(&s[t], hipStreamNonBlocking);
      std::memset(h[t], t, bytes);
    }
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        for (int r = 0; r < reps; ++r) {
          (void)hipMemcpyAsync(d[t], h[t], bytes, hipMemcpyHostToDevice, s[t]);
          (void)hipStreamSynchronize(s[t]);
        }
      });
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("threads %2d  %.1f GB/s  (%.3f ms per %zu-byte copy)\n", T, (double)bytes * reps * T / dt / 1e9, dt / reps * 1e3, bytes);
    for (int t = 0; t < T; ++t) { (void)hipHostFree(h[t]); (void)hipFree(d[t]); (void)hipStreamDestroy(s[t]); }
  }
  return 0;
}
