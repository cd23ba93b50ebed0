// third look: float forms of the DP operations, select forms
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define K(name, PRE, STR)                                                         \
  __global__ void name(int* out, int seed) {                                      \
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    int b = seed * 7 + 1; unsigned long long m = 0x5555555555555555ull * (unsigned)(seed + 1);      \
    asm volatile(PRE :: "v"(b));                                                  \
    for (int i = 0; i < ITER; ++i) {                                              \
      asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m)); } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
  }
#define R8(f) f("%0","%1") f("%1","%2") f("%2","%3") f("%3","%4") f("%4","%5") f("%5","%6") f("%6","%7") f("%7","%0")
#define NOPRE ""
#define VCCPRE "v_cmp_gt_i32 vcc, %0, 3\n\t"
#define MAXF(d,s) "v_max_f32 " d ", " d ", " s "\n\t"
#define MAX3F(d,s) "v_max3_f32 " d ", " d ", " s ", %8\n\t"
#define ADDF(d,s) "v_add_f32 " d ", " d ", " s "\n\t"
#define SUBF(d,s) "v_sub_f32 " d ", " d ", " s "\n\t"
#define FMAF(d,s) "v_fma_f32 " d ", " d ", " s ", %8\n\t"
#define MAXFDPP(d,s) "v_max_f32_dpp " d ", " s ", " d " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define ADDFDPP(d,s) "v_add_f32_dpp " d ", " s ", " d " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define CMPF(d,s) "v_cmp_gt_f32 vcc, " d ", " s "\n\t"
#define CND_VCC(d,s) "v_cndmask_b32 " d ", " d ", " s ", vcc\n\t"
#define CND_SG(d,s) "v_cndmask_b32 " d ", " d ", " s ", %9\n\t"
#define PKMAXF(d,s) "v_pk_max_f16 " d ", " d ", " s "\n\t"
#define PKADDF(d,s) "v_pk_add_f16 " d ", " d ", " s "\n\t"
#define PKFMA(d,s) "v_pk_fma_f16 " d ", " d ", " s ", %8\n\t"
#define MAXI16(d,s) "v_max_i16 " d ", " d ", " s "\n\t"
#define MINF(d,s) "v_min_f32 " d ", " d ", " s "\n\t"
#define MED3F(d,s) "v_med3_f32 " d ", " d ", " s ", %8\n\t"
#define OR(d,s) "v_or_b32 " d ", " d ", " s "\n\t"
#define XOR(d,s) "v_xor_b32 " d ", " d ", " s "\n\t"
#define LSHL(d,s) "v_lshlrev_b32 " d ", 3, " d "\n\t"
#define MADU24(d,s) "v_mad_u32_u24 " d ", " d ", " s ", %8\n\t"
#define PERM(d,s) "v_perm_b32 " d ", " d ", " s ", %8\n\t"
#define SUBREV(d,s) "v_subrev_u32 " d ", " d ", " s "\n\t"
#define ADDCO(d,s) "v_add_co_u32 " d ", vcc, " d ", " s "\n\t"
K(k1, NOPRE, R8(MAXF)) K(k2, NOPRE, R8(MAX3F)) K(k3, NOPRE, R8(ADDF)) K(k4, NOPRE, R8(SUBF)) K(k5, NOPRE, R8(FMAF)) K(k6, NOPRE, R8(MAXFDPP))
K(k7, NOPRE, R8(ADDFDPP)) K(k8, NOPRE, R8(CMPF)) K(k9, VCCPRE, R8(CND_VCC)) K(k10, NOPRE, R8(CND_SG)) K(k11, NOPRE, R8(PKMAXF)) K(k12, NOPRE, R8(PKADDF))
K(k13, NOPRE, R8(PKFMA)) K(k14, NOPRE, R8(MAXI16)) K(k15, NOPRE, R8(MINF)) K(k16, NOPRE, R8(MED3F)) K(k17, NOPRE, R8(OR)) K(k18, NOPRE, R8(XOR))
K(k19, NOPRE, R8(LSHL)) K(k20, NOPRE, R8(MADU24)) K(k21, NOPRE, R8(PERM)) K(k22, NOPRE, R8(SUBREV)) K(k23, NOPRE, R8(ADDCO))
int main() {
  int* d; const int blocks = 256 * 8, threads = 256;
  CHK(hipMalloc(&d, sizeof(int) * blocks * threads));
  struct { const char* n; void (*f)(int*, int); } ks[] = {{"v_max_f32", k1}, {"v_max3_f32", k2}, {"v_add_f32", k3}, {"v_sub_f32", k4}, {"v_fma_f32", k5},
    {"v_max_f32_dpp row_shr:1", k6}, {"v_add_f32_dpp row_shr:1", k7}, {"v_cmp_gt_f32 vcc", k8}, {"v_cndmask vcc (set once)", k9}, {"v_cndmask sgpr-pair mask", k10},
    {"v_pk_max_f16", k11}, {"v_pk_add_f16", k12}, {"v_pk_fma_f16", k13}, {"v_max_i16", k14}, {"v_min_f32", k15}, {"v_med3_f32", k16}, {"v_or_b32", k17},
    {"v_xor_b32", k18}, {"v_lshlrev_b32", k19}, {"v_mad_u32_u24", k20}, {"v_perm_b32", k21}, {"v_subrev_u32", k22}, {"v_add_co_u32", k23}};
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, d, 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, d, r);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double instr_per_simd = 8.0 * ITER * 8.0 * (blocks / (double)(p.multiProcessorCount * 8));
    printf("%-28s %8.3f ms   %6.2f ns per wave-instr per SIMD\n", k.n, ms, ms * 1e6 / instr_per_simd);
  }
  return 0;
}
