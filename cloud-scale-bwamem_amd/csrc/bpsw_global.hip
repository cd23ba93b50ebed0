// bpsw_global.hip -- banded global alignment with backtrack -> CIGAR (SWUtil.SWGlobal, SWUtil.scala:233-397,
// == ksw_global2, native/ksw.c:501-584), gfx950, for jobs handed over as bytes (bpsw_global_batch).  The DP itself is
// in bpsw_global_core.h, shared with the memRegToAln kernel (bpsw_reg2aln.hip).
#include "bpsw_global_core.h"

namespace bpsw {
namespace {

constexpr int WAVES_PER_BLOCK = 4;

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void global_kernel(const GlobalJobsDev jobs, const SwScoring sc,
                                                                        int32_t* __restrict__ out_score,
                                                                        int32_t* __restrict__ out_ncigar,
                                                                        uint32_t* __restrict__ out_cigar,
                                                                        uint8_t* __restrict__ zscratch,
                                                                        const unsigned long long z_per_wave, const int qcap,
                                                                        const int lds_per_wave) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const int slot = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  unsigned char* base = smem + (size_t)wave * lds_per_wave;
  int2* eh = reinterpret_cast<int2*>(base);                                  // qcap + 2 entries
  uint32_t* cig = reinterpret_cast<uint32_t*>(base + 8 * (size_t)(qcap + 2));  // CIG_LDS entries
  int8_t* qp = reinterpret_cast<int8_t*>(cig + CIG_LDS);                      // 5 x qLen profile
  uint8_t* z = zscratch + (size_t)slot * z_per_wave;
  const int stride = gridDim.x * WAVES_PER_BLOCK;

  for (int job = slot; job < jobs.n; job += stride) {
    const int qLen = uni(jobs.q_len[job]), tLen = uni(jobs.t_len[job]), w = uni(jobs.w[job]);
    const uint8_t* __restrict__ q = jobs.q_pool + jobs.q_off[job];
    const uint8_t* __restrict__ tg = jobs.t_pool + jobs.t_off[job];
    const int nCol = qLen < 2 * w + 1 ? qLen : 2 * w + 1;  // SWUtil.scala:248-249

    global_init(lane, qLen, w, sc, q, eh, qp);
    const int score = global_rows(lane, qLen, tLen, w, sc, tg, eh, qp, z, nCol);
    const int n = global_backtrack(lane, qLen, tLen, w, z, nCol, cig);
    // the list was produced from the last operation to the first: write it out reversed (SWUtil.scala:384-394)
    const int cap = min(jobs.max_cigar, CIG_LDS);
    uint32_t* o = out_cigar + (size_t)job * jobs.max_cigar;
    if (n <= cap)
      for (int x = lane; x < n; x += 64) o[x] = cig[n - 1 - x];
    if (lane == 0) {
      out_score[job] = score;
      out_ncigar[job] = n;  // n > max_cigar: the caller must resubmit this job with a larger max_cigar
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void global_prepass_kernel(const GlobalJobsDev jobs, const unsigned long long q_pool_bytes,
                                      const unsigned long long t_pool_bytes, GlobalPrepass* __restrict__ pre) {
  int mq = 0, err = 0;
  unsigned long long mz = 0;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < jobs.n; j += gridDim.x * blockDim.x) {
    const int ql = jobs.q_len[j], tl = jobs.t_len[j], w = jobs.w[j];
    const long long qo = jobs.q_off[j], to = jobs.t_off[j];
    if (ql < 1 || tl < 1 || w < 0 || w > 0x3fffffff || qo < 0 || to < 0 || (unsigned long long)(qo + ql) > q_pool_bytes ||
        (unsigned long long)(to + tl) > t_pool_bytes) {
      err = 1;
      continue;
    }
    const long long ncol = ql < 2ll * w + 1 ? ql : 2ll * w + 1;
    mq = max(mq, ql);
    const unsigned long long zb = (unsigned long long)ncol * (unsigned long long)tl;
    mz = zb > mz ? zb : mz;
  }
  if (mq) atomicMax(&pre->max_qlen, mq);
  if (mz) atomicMax(&pre->max_z, mz);
  if (err) atomicMax(&pre->error, err);
}

}  // namespace

size_t global_lds_per_wave(int qcap) {
  size_t b = 8 * (size_t)(qcap + 2) + 4 * (size_t)CIG_LDS + 5 * (size_t)qcap;
  return (b + 15) & ~(size_t)15;
}
int global_resident_waves(int num_cu, int qcap) {
  const size_t lds = global_lds_per_wave(qcap) * WAVES_PER_BLOCK;
  int per_cu = (int)((160 * 1024) / lds);
  per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
  return num_cu * per_cu * WAVES_PER_BLOCK;
}

void launch_global_prepass(const GlobalJobsDev& jobs, size_t q_pool_bytes, size_t t_pool_bytes, GlobalPrepass* d_pre,
                           hipStream_t s) {
  const int threads = 256;
  int blocks = (jobs.n + threads - 1) / threads;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(global_prepass_kernel, dim3(blocks), dim3(threads), 0, s, jobs, (unsigned long long)q_pool_bytes,
                     (unsigned long long)t_pool_bytes, d_pre);
}

hipError_t launch_global_kernel(const GlobalJobsDev& jobs, const SwScoring& sc, int max_qlen, size_t z_per_wave,
                                int32_t* d_score, int32_t* d_ncigar, uint32_t* d_cigar, uint8_t* d_z, int num_cu,
                                hipStream_t s) {
  if (jobs.n <= 0) return hipSuccess;
  const int qcap = (max_qlen + 31) & ~31;
  const size_t per_wave = global_lds_per_wave(qcap);
  const size_t lds = per_wave * WAVES_PER_BLOCK;
  if (lds > 64 * 1024) return hipErrorInvalidValue;  // qLen <= BPSW_GLOBAL_MAX_QLEN keeps this below 64 KB
  int blocks = (jobs.n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  const int max_blocks = global_resident_waves(num_cu, qcap) / WAVES_PER_BLOCK;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(global_kernel, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), lds, s, jobs, sc, d_score, d_ncigar, d_cigar,
                     d_z, (unsigned long long)z_per_wave, qcap, (int)per_wave);
  return hipGetLastError();
}

}  // namespace bpsw
