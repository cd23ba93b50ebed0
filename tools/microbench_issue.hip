// microbench_issue.hip -- issue cost of the cross-lane instructions the SW kernels lean on (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_issue.hip ; run on an MI355X.
// Each kernel runs ITER iterations of 8 independent chains of one instruction kind, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define BODY8(INS) \
  asm volatile(INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7") \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));

#define K(name, INS)                                                              \
  __global__ void name(int* out, int seed) {                                      \
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    for (int i = 0; i < ITER; ++i) { BODY8(INS) }                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
  }
#define I_PLAIN(r) "v_max_i32 " r ", " r ", " r "\n\t"
#define I_ADD(r) "v_add_u32 " r ", 1, " r "\n\t"
#define I_SHR(r) "v_max_i32_dpp " r ", " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_ROR(r) "v_max_i32_dpp " r ", " r ", " r " row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_BC15(r) "v_max_i32_dpp " r ", " r ", " r " row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
#define I_BC31(r) "v_max_i32_dpp " r ", " r ", " r " row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
#define I_WSHR(r) "v_mov_b32_dpp " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_MAX3(r) "v_max3_i32 " r ", " r ", " r ", 0\n\t"
#define I_BFE(r) "v_bfe_i32 " r ", " r ", 3, 8\n\t"
#define I_CNDM(r) "v_cndmask_b32 " r ", " r ", 1, vcc\n\t"
#define I_CMP(r) "v_cmp_gt_i32 vcc, " r ", 5\n\t"
#define I_RDLN(r) "v_readlane_b32 s20, " r ", 63\n\t"
#define I_SWZ(r) "ds_swizzle_b32 " r ", " r " offset:0x0010\n\ts_waitcnt lgkmcnt(0)\n\t"
#define I_SALU(r) "s_add_u32 s20, s20, 1\n\t"
#define I_SFLB(r) "s_flbit_i32_b64 s20, s[22:23]\n\t"
#define I_NOP(r) "s_nop 0\n\t"
K(k_plain, I_PLAIN) K(k_add, I_ADD) K(k_shr, I_SHR) K(k_ror, I_ROR) K(k_bc15, I_BC15) K(k_bc31, I_BC31) K(k_wshr, I_WSHR)
K(k_max3, I_MAX3) K(k_bfe, I_BFE) K(k_cndm, I_CNDM) K(k_cmp, I_CMP) K(k_rdln, I_RDLN) K(k_swz, I_SWZ) K(k_salu, I_SALU)
K(k_sflb, I_SFLB) K(k_nop, I_NOP)

int main() {
  int* d; const int blocks = 256 * 8, threads = 256;
  CHK(hipMalloc(&d, sizeof(int) * blocks * threads));
  struct { const char* n; void (*f)(int*, int); } ks[] = {{"v_max_i32", k_plain}, {"v_add_u32", k_add}, {"dpp row_shr:1", k_shr},
    {"dpp row_ror:1", k_ror}, {"dpp row_bcast:15", k_bc15}, {"dpp row_bcast:31", k_bc31}, {"dpp wave_shr:1 (mov)", k_wshr}, {"v_max3_i32", k_max3},
    {"v_bfe_i32", k_bfe}, {"v_cndmask", k_cndm}, {"v_cmp->vcc", k_cmp}, {"v_readlane", k_rdln}, {"ds_swizzle+wait", k_swz}, {"s_add_u32", k_salu},
    {"s_flbit_i32_b64", k_sflb}, {"s_nop 0", k_nop}};
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  printf("device %s, %d CUs, clock %d kHz; 8 waves/SIMD, %d iterations x 8 instr\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, ITER);
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, d, 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, d, r);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    // per SIMD: waves = blocks*4/(CUs*4) = 8 waves; instr per SIMD = 8 waves * ITER * 8
    const double instr_per_simd = 8.0 * ITER * 8.0 * (blocks / (double)(p.multiProcessorCount * 8));
    printf("%-24s %8.3f ms   %6.2f ns per wave-instr per SIMD  (= %5.2f cycles at 2.1 GHz)\n", k.n, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.1);
  }
  return 0;
}
