// second look: does operand choice change the issue cost?  (same harness as microbench_issue.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define K(name, STR)                                                              \
  __global__ void name(int* out, int seed) {                                      \
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    int b = seed * 7 + 1;                                                         \
    for (int i = 0; i < ITER; ++i) {                                              \
      asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(seed)); } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
  }
#define R8(f) f("%0","%1") f("%1","%2") f("%2","%3") f("%3","%4") f("%4","%5") f("%5","%6") f("%6","%7") f("%7","%0")
#define MAX_RR(d,s) "v_max_i32 " d ", " d ", " s "\n\t"
#define MAX_RB(d,s) "v_max_i32 " d ", " d ", %8\n\t"
#define MAX_RS(d,s) "v_max_i32 " d ", %9, " d "\n\t"
#define MAX_RC(d,s) "v_max_i32 " d ", 7, " d "\n\t"
#define ADD_RR(d,s) "v_add_u32 " d ", " d ", " s "\n\t"
#define ADD_SAME(d,s) "v_add_u32 " d ", " d ", " d "\n\t"
#define SUB_RR(d,s) "v_sub_u32 " d ", " d ", " s "\n\t"
#define MAX3_RR(d,s) "v_max3_i32 " d ", " d ", " s ", %8\n\t"
#define CND_RR(d,s) "v_cndmask_b32 " d ", " d ", " s ", vcc\n\t"
#define MOV_RR(d,s) "v_mov_b32 " d ", " s "\n\t"
#define DPP_RR(d,s) "v_max_i32_dpp " d ", " s ", " d " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define MOVDPP(d,s) "v_mov_b32_dpp " d ", " s " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define ADD3(d,s) "v_add3_u32 " d ", " d ", " s ", %8\n\t"
#define LSHLOR(d,s) "v_lshl_or_b32 " d ", " d ", 8, " s "\n\t"
#define BFE_RR(d,s) "v_bfe_i32 " d ", " d ", " s ", 8\n\t"
#define AND_RR(d,s) "v_and_b32 " d ", " d ", " s "\n\t"
#define CMP_RR(d,s) "v_cmp_gt_i32 vcc, " d ", " s "\n\t"
#define MAXU(d,s) "v_max_u32 " d ", " d ", " s "\n\t"
#define PKMAX(d,s) "v_pk_max_i16 " d ", " d ", " s "\n\t"
#define PKADD(d,s) "v_pk_add_i16 " d ", " d ", " s "\n\t"
K(k1, R8(MAX_RR)) K(k2, R8(MAX_RB)) K(k3, R8(MAX_RS)) K(k4, R8(MAX_RC)) K(k5, R8(ADD_RR)) K(k6, R8(ADD_SAME)) K(k7, R8(SUB_RR))
K(k8, R8(MAX3_RR)) K(k9, R8(CND_RR)) K(k10, R8(MOV_RR)) K(k11, R8(DPP_RR)) K(k12, R8(MOVDPP)) K(k13, R8(ADD3)) K(k14, R8(LSHLOR))
K(k15, R8(BFE_RR)) K(k16, R8(AND_RR)) K(k17, R8(CMP_RR)) K(k18, R8(MAXU)) K(k19, R8(PKMAX)) K(k20, R8(PKADD))
int main() {
  int* d; const int blocks = 256 * 8, threads = 256;
  CHK(hipMalloc(&d, sizeof(int) * blocks * threads));
  struct { const char* n; void (*f)(int*, int); } ks[] = {{"v_max_i32 d,d,s", k1}, {"v_max_i32 d,d,b(shared)", k2}, {"v_max_i32 d,sgpr,d", k3},
    {"v_max_i32 d,7,d", k4}, {"v_add_u32 d,d,s", k5}, {"v_add_u32 d,d,d", k6}, {"v_sub_u32 d,d,s", k7}, {"v_max3_i32 d,d,s,b", k8},
    {"v_cndmask d,d,s,vcc", k9}, {"v_mov_b32 d,s", k10}, {"v_max_i32_dpp d,s,d", k11}, {"v_mov_b32_dpp d,s", k12}, {"v_add3_u32 d,d,s,b", k13},
    {"v_lshl_or_b32 d,d,8,s", k14}, {"v_bfe_i32 d,d,s,8", k15}, {"v_and_b32 d,d,s", k16}, {"v_cmp_gt_i32 vcc,d,s", k17}, {"v_max_u32 d,d,s", k18},
    {"v_pk_max_i16 d,d,s", k19}, {"v_pk_add_i16 d,d,s", k20}};
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
