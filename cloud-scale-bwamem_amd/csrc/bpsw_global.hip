// bpsw_global.hip -- banded global alignment with backtrack -> CIGAR (SWUtil.SWGlobal, SWUtil.scala:233-397,
// == ksw_global2, native/ksw.c:501-584), gfx950.  SURVEY.md 8(f) item 1: not behind either JNI today
// (bwaGenCigar2 calls it in Scala, MemRegToADAMSAM.scala:804); offered as an additional export.
//
// One job per wavefront.  The band of row i is the static window [max(0,i-w), min(qLen,i+w+1)), swept in 64-column
// chunks.  In the global recurrence the horizontal gap is opened from M (not H):
//     F(i,j+1) = max(F(i,j) - eIns, M(i,j) - oeIns),   M(i,j) = H(i-1,j-1) + S(i,j)
// so F is a pure max-plus prefix scan of M: F(i,j) = max_{k<j}(M(k) - oeIns - (j-1-k)*eIns), one DPP scan per chunk.
// The (H,E) row lives in LDS as int2; the direction byte of every cell (SWUtil.scala:321-339) goes to a per-wave
// scratch matrix z[i*nCol + j-beg] in global memory (row-major, so a chunk's 64 bytes are one coalesced store);
// the backtrack (SWUtil.scala:355-382) walks it and the CIGAR is staged in LDS, then written out reversed.
#include "bpsw_internal.h"
#include "bpsw_wave.h"

namespace bpsw {
namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int MINUS_INF = -0x40000000;  // SWUtil.scala:28
constexpr int CIG_LDS = 512;            // CIGAR operations staged per wave

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
  const int oDel = sc.o_del, eDel = sc.e_del, oIns = sc.o_ins, eIns = sc.e_ins;
  const int oeDel = oDel + eDel, oeIns = oIns + eIns;
  const int stride = gridDim.x * WAVES_PER_BLOCK;

  for (int job = slot; job < jobs.n; job += stride) {
    const int qLen = uni(jobs.q_len[job]), tLen = uni(jobs.t_len[job]), w = uni(jobs.w[job]);
    const uint8_t* __restrict__ q = jobs.q_pool + jobs.q_off[job];
    const uint8_t* __restrict__ tg = jobs.t_pool + jobs.t_off[job];
    const int nCol = qLen < 2 * w + 1 ? qLen : 2 * w + 1;  // SWUtil.scala:248-249

    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < qLen; j += 64) {  // query profile, SWUtil.scala:258-271
      int c = q[j];
      c = c > 4 ? 4 : c;
#pragma unroll
      for (int k = 0; k < 5; ++k) qp[k * qLen + j] = (int8_t)((sc.mat.row[k] >> (8 * c)) & 0xff);
    }
    for (int j = lane; j <= qLen; j += 64) {  // first row, SWUtil.scala:274-288
      const int h = j == 0 ? 0 : (j <= w ? -(oIns + eIns * j) : MINUS_INF);
      eh[j] = make_int2(h, MINUS_INF);
    }
    __builtin_amdgcn_wave_barrier();

    for (int i = 0; i < tLen; ++i) {  // SWUtil.scala:292-349
      int t = uni((int)tg[i]);
      t = t > 4 ? 4 : t;
      const int beg = i > w ? i - w : 0;
      const int end = i + w + 1 < qLen ? i + w + 1 : qLen;
      const int8_t* __restrict__ prof = qp + t * qLen;
      uint8_t* __restrict__ zi = z + (size_t)i * nCol;
      int carry = NEG;                                          // prefix max of g over the columns already swept
      int hleft = beg == 0 ? -(oDel + eDel * (i + 1)) : MINUS_INF;  // h1 before the first column
      for (int j0 = beg; j0 < end; j0 += 64) {
        const int j = j0 + lane;
        const bool act = j < end;
        int2 he = make_int2(0, 0);
        int s = 0;
        if (act) {
          he = eh[j];
          s = prof[j];
        }
        const int M = he.x + s;                                  // M(i,j) = H(i-1,j-1) + S(i,j)
        const int jE = j * eIns - oeIns;
        const int P = max(wave_scan_max(act ? M + jE : NEG), carry);
        const int Pex = wave_shr1(carry, P);
        carry = __builtin_amdgcn_readlane(P, 63);
        const int F = Pex - (jE + oeIns - eIns);                 // F(i,j); "-inf" at j == beg
        int e = he.y;
        int d = M >= e ? 0 : 1;                                  // SWUtil.scala:321-327
        int h = M >= e ? M : e;
        if (h < F) { d = 2; h = F; }
        int tt = M - oeDel;                                      // SWUtil.scala:328-333
        e -= eDel;
        if (e > tt) d |= 1 << 2;
        e = e > tt ? e : tt;
        tt = M - oeIns;                                          // SWUtil.scala:334-338
        if (F - eIns > tt) d |= 2 << 4;
        const int Hprev = wave_shr1(hleft, h);                   // H(i,j-1) -> eh[j].h
        if (act) {
          eh[j] = make_int2(Hprev, e);
          zi[j - beg] = (uint8_t)d;                              // SWUtil.scala:339
        }
        const int nact = min(64, end - j0);
        hleft = __builtin_amdgcn_readlane(h, nact - 1);
      }
      if (lane == 0) eh[end] = make_int2(hleft, MINUS_INF);      // SWUtil.scala:345-346
      __builtin_amdgcn_wave_barrier();
    }
    const int score = uni(eh[qLen].x);                           // SWUtil.scala:351

    // ---- backtrack, SWUtil.scala:355-382 (every lane walks the same cells: uniform loads) -----------------
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");       // this wave's z stores -> its loads
    int n = 0, which = 0, last_op = -1, last_len = 0;
    int i = tLen - 1;
    int k = (i + w + 1 < qLen ? i + w + 1 : qLen) - 1;
    auto push = [&](int op, int len) {                           // pushCigar, SWUtil.scala:401-414
      if (n == 0 || op != last_op) {
        if (n > 0 && n - 1 < CIG_LDS && lane == 0) cig[n - 1] = ((uint32_t)last_len << 4) | (uint32_t)last_op;
        ++n;
        last_op = op;
        last_len = len;
      } else {
        last_len += len;
      }
    };
    while (i >= 0 && k >= 0) {
      const size_t idx = i > w ? (size_t)i * nCol + (k - (i - w)) : (size_t)i * nCol + k;
      which = uni(((int)z[idx] >> (which << 1)) & 3);
      if (which == 0) { push(0, 1); --i; --k; }
      else if (which == 1) { push(2, 1); --i; }
      else { push(1, 1); --k; }
    }
    if (i >= 0) push(2, i + 1);
    if (k >= 0) push(1, k + 1);
    if (n > 0 && n - 1 < CIG_LDS && lane == 0) cig[n - 1] = ((uint32_t)last_len << 4) | (uint32_t)last_op;
    __builtin_amdgcn_wave_barrier();
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
