// fetch_calib.hip -- what FETCH_SIZE / WRITE_SIZE report for the access patterns of this library's kernels, against known byte
// counts (MI355X_MICROARCH.md: only 16-B-per-lane streaming accesses are calibrated; "calibrate on a known byte count in your
// own access pattern before trusting an absolute").  Build: hipcc --offload-arch=gfx950 -O2 -o fetch_calib fetch_calib.hip
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o c -- ./fetch_calib      (and once more with WRITE_SIZE)
// Every kernel touches N = 256 MiB (past the 256 MiB Infinity Cache together with its output) exactly once.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr size_t N = 256ull << 20;

__global__ void read16(const uint4* __restrict__ p, size_t n, unsigned* sink) {  // 16 B per lane, streaming (the calibrated case)
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void read4(const uint32_t* __restrict__ p, size_t n, unsigned* sink) {  // 4 B per lane, coalesced (sequence words)
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void read1(const uint8_t* __restrict__ p, size_t n, unsigned* sink) {  // 1 B per lane, coalesced (bases as bytes)
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void read_rec(const uint32_t* __restrict__ p, size_t n_rec, unsigned* sink) {  // one 32 B record per WAVE, scalar-style (task records)
  unsigned acc = 0;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t r = wave; r < n_rec; r += nw) {
    const uint32_t* q = p + 8 * r * 5;  // records 160 B apart: each touches its own 128-B line(s)
    for (int k = 0; k < 8; ++k) acc ^= q[k];
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void atomics(int* counter, int per_wave) {  // returning device-scope atomics on one word (the task queue)
  int acc = 0;
  for (int k = 0; k < per_wave; ++k)
    if ((threadIdx.x & 63) == 0) acc += atomicAdd(counter, 1);
  if (acc == -1) *counter = 0;
}
__global__ void write16(uint4* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, (unsigned)i);
}
__global__ void write_rec20(uint32_t* __restrict__ p, size_t n_rec) {  // 20 B per record by lane 0 of a wave, records back to back (extension results)
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t r = wave; r < n_rec; r += nw)
    if ((threadIdx.x & 63) == 0) { uint32_t* o = p + 5 * ((r * 2654435761ull) % n_rec); o[0] = 1; o[1] = 2; o[2] = 3; o[3] = 4; o[4] = (uint32_t)r; }
}

int main() {
  void *a, *b; unsigned* sink; int* counter;
  hipMalloc(&a, N); hipMalloc(&b, N); hipMalloc(&sink, 4); hipMalloc(&counter, 4);
  hipMemset(a, 1, N); hipMemset(b, 0, N); hipMemset(counter, 0, 4);
  hipDeviceSynchronize();
  const int blocks = 256 * 8, threads = 256;
  hipLaunchKernelGGL(read16, dim3(blocks), dim3(threads), 0, 0, (const uint4*)a, N / 16, sink);
  hipLaunchKernelGGL(write16, dim3(blocks), dim3(threads), 0, 0, (uint4*)b, N / 16);
  hipLaunchKernelGGL(read4, dim3(blocks), dim3(threads), 0, 0, (const uint32_t*)a, N / 4, sink);
  hipLaunchKernelGGL(write16, dim3(blocks), dim3(threads), 0, 0, (uint4*)b, N / 16);
  hipLaunchKernelGGL(read1, dim3(blocks), dim3(threads), 0, 0, (const uint8_t*)a, N, sink);
  hipLaunchKernelGGL(write16, dim3(blocks), dim3(threads), 0, 0, (uint4*)b, N / 16);
  hipLaunchKernelGGL(read_rec, dim3(blocks), dim3(threads), 0, 0, (const uint32_t*)a, N / 160, sink);   // 1 677 721 records x 32 B = 53.7 MB useful
  hipLaunchKernelGGL(atomics, dim3(256), dim3(256), 0, 0, counter, 1024);                               // 1 048 576 atomics
  hipLaunchKernelGGL(write_rec20, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)b, (size_t)4 << 20);    // 4 194 304 records x 20 B = 83.9 MB
  hipDeviceSynchronize();
  printf("N = %zu bytes; read_rec useful %zu; atomics %d; write_rec20 useful %zu\n", N, (N / 160) * 32, 256 * 4 * 1024, ((size_t)4 << 20) * 20);
  return 0;
}
