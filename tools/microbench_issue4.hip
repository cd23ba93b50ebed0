// fourth look: compare+select pairs, 16-bit VOP2 forms
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define K(name, STR, ...)                                                        \
  __global__ void name(int* out, int seed) {                                      \
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    int b = seed * 7 + 1;                                                         \
    for (int i = 0; i < ITER; ++i) {                                              \
      asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : __VA_ARGS__); } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
  }
#define R8(f) f("%0","%1") f("%1","%2") f("%2","%3") f("%3","%4") f("%4","%5") f("%5","%6") f("%6","%7") f("%7","%0")
#define PAIR_VCC(d,s) "v_cmp_gt_i32 vcc, " d ", " s "\n\tv_cndmask_b32 " d ", " d ", " s ", vcc\n\t"
#define PAIR_SG(d,s) "v_cmp_gt_i32 s[20:21], " d ", " s "\n\tv_cndmask_b32 " d ", " d ", " s ", s[20:21]\n\t"
#define PAIR_SG_ROT(d,s) "v_cmp_gt_i32 s[20:21], " d ", " s "\n\tv_cndmask_b32 " d ", " d ", " s ", s[22:23]\n\tv_cmp_gt_i32 s[22:23], " s ", %8\n\t"
#define MAXI16(d,s) "v_max_i16 " d ", " d ", " s "\n\t"
#define MINI16(d,s) "v_min_i16 " d ", " d ", " s "\n\t"
#define ADDU16(d,s) "v_add_u16 " d ", " d ", " s "\n\t"
#define SUBU16(d,s) "v_sub_u16 " d ", " d ", " s "\n\t"
#define MAXU16(d,s) "v_max_u16 " d ", " d ", " s "\n\t"
#define MAXI16DPP(d,s) "v_max_i16_dpp " d ", " s ", " d " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define ADDDPP(d,s) "v_add_u32_dpp " d ", " s ", " d " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define MAXI16SDWA(d,s) "v_max_i16_sdwa " d ", " d ", " s " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n\t"
#define ASHR(d,s) "v_ashrrev_i32 " d ", 16, " d "\n\t"
#define CMPX(d,s) "v_cmp_gt_i16 vcc, " d ", " s "\n\t"
K(k1, R8(PAIR_VCC), "vcc") K(k2, R8(PAIR_SG), "s20", "s21") K(k3, R8(PAIR_SG_ROT), "s20", "s21", "s22", "s23")
K(k4, R8(MAXI16), "memory") K(k5, R8(MINI16), "memory") K(k6, R8(ADDU16), "memory") K(k7, R8(SUBU16), "memory") K(k8, R8(MAXU16), "memory")
K(k9, R8(MAXI16DPP), "memory") K(k10, R8(ADDDPP), "memory") K(k11, R8(MAXI16SDWA), "memory") K(k12, R8(ASHR), "memory") K(k13, R8(CMPX), "vcc")
__global__ void k_upper(int* out) {  // what does v_max_i16 leave in the upper half of the destination?
  int a = 0x12340005, b = 0x7fff0009, d = (int)0xdeadbeef;
  asm volatile("v_max_i16 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
  int e = (int)0xdeadbeef;
  asm volatile("v_add_u16 %0, %1, %2" : "+v"(e) : "v"(a), "v"(b));
  if (threadIdx.x == 0) { out[0] = d; out[1] = e; }
}
int main() {
  int* d; const int blocks = 256 * 8, threads = 256;
  CHK(hipMalloc(&d, sizeof(int) * blocks * threads));
  struct { const char* n; void (*f)(int*, int); int per; } ks[] = {{"v_cmp vcc + v_cndmask vcc", k1, 2}, {"v_cmp s[20:21] + v_cndmask s[20:21]", k2, 2},
    {"cmp->s20, cndmask s22, cmp->s22 (3 instr)", k3, 3}, {"v_max_i16", k4, 1}, {"v_min_i16", k5, 1}, {"v_add_u16", k6, 1}, {"v_sub_u16", k7, 1},
    {"v_max_u16", k8, 1}, {"v_max_i16_dpp row_shr:1", k9, 1}, {"v_add_u32_dpp row_shr:1", k10, 1}, {"v_max_i16_sdwa WORD_1", k11, 1}, {"v_ashrrev_i32 16", k12, 1},
    {"v_cmp_gt_i16 vcc", k13, 1}};
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, d, 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, d, r);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double instr_per_simd = 8.0 * ITER * 8.0 * k.per * (blocks / (double)(p.multiProcessorCount * 8));
    printf("%-44s %8.3f ms   %6.2f ns per wave-instr per SIMD\n", k.n, ms, ms * 1e6 / instr_per_simd);
  }
  hipLaunchKernelGGL(k_upper, dim3(1), dim3(64), 0, 0, d);
  int h[2]; CHK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
  printf("v_max_i16(0x12340005, 0x7fff0009) into 0xdeadbeef -> 0x%08x ; v_add_u16 -> 0x%08x\n", h[0], h[1]);
  return 0;
}
