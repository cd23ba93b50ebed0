// Micro-benchmark: issue rate of the instruction kinds the SW kernels are made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Each kernel runs N iterations of 32 independent instructions of one kind per wave; blocks = 256 CUs x waves-per-SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(int* out, int n, int seed) {
  int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  int b = seed * 77 + 1;
  for (int i = 0; i < n; ++i) {
    if (KIND == 0) {  // v_max_i32
      REP8(asm volatile("v_max_i32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_max_i32 %0, %0, %1" : "+v"(a1) : "v"(b));
           asm volatile("v_max_i32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_max_i32 %0, %0, %1" : "+v"(a3) : "v"(b));)
    } else if (KIND == 1) {  // v_pk_max_u16
      REP8(asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a1) : "v"(b));
           asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a3) : "v"(b));)
    } else if (KIND == 2) {  // v_pk_sub_u16 clamp
      REP8(asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a0) : "v"(b)); asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a1) : "v"(b));
           asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a2) : "v"(b)); asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a3) : "v"(b));)
    } else if (KIND == 3) {  // v_mov_b32_dpp wave_shr:1
      REP8(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(a4)); asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(a5));
           asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a2) : "v"(a6)); asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(a7));)
    } else if (KIND == 4) {  // v_mov_b32_dpp row_shr:1
      REP8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(a4)); asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(a5));
           asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a2) : "v"(a6)); asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(a7));)
    } else if (KIND == 5) {  // v_perm_b32
      REP8(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(a4)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(a5));
           asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(a6)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(a7));)
    } else if (KIND == 6) {  // v_pk_mad_u16
      REP8(asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(a4)); asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(a5));
           asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(a6)); asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(a7));)
    } else if (KIND == 7) {  // v_max3_i32
      REP8(asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(a4)); asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(a5));
           asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(a6)); asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(a7));)
    } else if (KIND == 8) {  // dependent chain v_pk_max_u16 (latency)
      REP8(asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a0) : "v"(b));
           asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a0) : "v"(b));)
    } else if (KIND == 9) {  // dependent chain v_max_i32
      REP8(asm volatile("v_max_i32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_max_i32 %0, %0, %1" : "+v"(a0) : "v"(b));
           asm volatile("v_max_i32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_max_i32 %0, %0, %1" : "+v"(a0) : "v"(b));)
    } else if (KIND == 10) {  // s_add (scalar)
      int s0 = seed;
      REP8(asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0) : : "scc"); asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0) : : "scc"); asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0) : : "scc"); asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0) : : "scc");)
      a0 += s0;
    } else if (KIND == 11) {  // dependent: dpp wave_shr -> pk_max -> dpp ...
      REP8(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(a0)); asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a0) : "v"(a1));
           asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(a0)); asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a0) : "v"(a1));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int KIND>
void run(const char* name, int* d, int waves_per_simd, int n) {
  const int blocks = 256 * waves_per_simd;  // 4 waves per block -> one per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 16, 1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, n, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = 32.0 * n * waves_per_simd;
  printf("%-34s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
         1e6 * ms / instr_per_simd, 2.4 * 1e6 * ms / instr_per_simd);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  int* d;
  hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
  const int n = getenv("N") ? atoi(getenv("N")) : 20000;
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_max_i32", d, w, n);
    run<1>("v_pk_max_u16", d, w, n);
    run<2>("v_pk_sub_u16 clamp", d, w, n);
    run<3>("v_mov_b32_dpp wave_shr:1", d, w, n);
    run<4>("v_mov_b32_dpp row_shr:1", d, w, n);
    run<5>("v_perm_b32", d, w, n);
    run<6>("v_pk_mad_u16", d, w, n);
    run<7>("v_max3_i32", d, w, n);
    run<8>("dependent v_pk_max_u16", d, w, n);
    run<9>("dependent v_max_i32", d, w, n);
    run<10>("s_add_i32 (dependent)", d, w, n);
    run<11>("dependent dpp->pk_max", d, w, n);
  }
  return 0;
}
