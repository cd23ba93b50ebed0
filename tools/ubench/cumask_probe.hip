// Which CUs / XCDs does a stream created with hipExtStreamCreateWithCUMask run on?  (tools/ubench: hipcc --offload-arch=gfx950 -O2)
//   ./cumask_probe <hex mask words, low word first>      e.g.  ffffffff ffffffff ffffffff ffffffff 0 0 0 0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
__global__ void probe(unsigned* out) {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  // stay a while so that the whole grid cannot fit on a few CUs one after the other
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) {}
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (xcc & 0xf) << 16 | ((hw >> 8) & 0xff);
}
int main(int argc, char** argv) {
  std::vector<uint32_t> mask;
  for (int i = 1; i < argc; ++i) mask.push_back((uint32_t)strtoul(argv[i], nullptr, 16));
  hipStream_t s;
  hipError_t e = mask.empty() ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("stream create failed: %s\n", hipGetErrorString(e)); return 1; }
  const int blocks = 2048, waves = blocks * 4;
  unsigned* d; hipMalloc(&d, waves * 4);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, s, d);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(waves);
  hipMemcpy(h.data(), d, waves * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, int> cus; int per_xcc[8] = {0};
  for (unsigned v : h) { ++cus[v]; ++per_xcc[(v >> 16) & 7]; }
  printf("mask words %zu: distinct (xcc, cu) %zu; waves per XCC:", mask.size(), cus.size());
  for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
  printf("\n");
  return 0;
}
