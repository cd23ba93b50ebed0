// bpsw_global_core.h -- the banded global alignment (SWUtil.SWGlobal, SWUtil.scala:233-397 == ksw_global2,
// native/ksw.c:501-584) as wave-level device routines, shared by global_kernel (bpsw_global.hip: jobs handed over as
// bytes) and reg2aln_kernel (bpsw_reg2aln.hip: memRegToAln on the device, sequences staged from the resident reference).
//
// One job per wavefront.  The band of row i is the static window [max(0,i-w), min(qLen,i+w+1)), swept in 64-column
// chunks.  In the global recurrence the horizontal gap is opened from M (not H):
//     F(i,j+1) = max(F(i,j) - eIns, M(i,j) - oeIns),   M(i,j) = H(i-1,j-1) + S(i,j)
// so F is a pure max-plus prefix scan of M: F(i,j) = max_{k<j}(M(k) - oeIns - (j-1-k)*eIns), one DPP scan per chunk.
// The (H,E) row lives in LDS as int2; the direction byte of every cell (SWUtil.scala:321-339) goes to a per-wave
// scratch matrix z[i*nCol + j-beg] in global memory (row-major, so a chunk's 64 bytes are one coalesced store);
// the backtrack (SWUtil.scala:355-382) walks it and leaves the CIGAR in LDS, last operation first.
#pragma once
#include "bpsw_internal.h"
#include "bpsw_wave.h"

namespace bpsw {
namespace {

constexpr int MINUS_INF = -0x40000000;  // SWUtil.scala:28
constexpr int CIG_LDS = 512;            // CIGAR operations staged per wave

// Query profile (SWUtil.scala:258-271) and first row (:274-288).  q: qLen codes (clamped to 0..4).
template <class QPtr>
__device__ __forceinline__ void global_init(const int lane, const int qLen, const int w, const SwScoring& sc, QPtr q,
                                            int2* __restrict__ eh, int8_t* __restrict__ qp) {
  const int oIns = sc.o_ins, eIns = sc.e_ins;
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < qLen; j += 64) {
    int c = q[j];
    c = c > 4 ? 4 : c;
#pragma unroll
    for (int k = 0; k < 5; ++k) qp[k * qLen + j] = (int8_t)((sc.mat.row[k] >> (8 * c)) & 0xff);
  }
  for (int j = lane; j <= qLen; j += 64) {
    const int h = j == 0 ? 0 : (j <= w ? -(oIns + eIns * j) : MINUS_INF);
    eh[j] = make_int2(h, MINUS_INF);
  }
  __builtin_amdgcn_wave_barrier();
}

// The DP rows, SWUtil.scala:292-349; returns the score (SWUtil.scala:351).  tg: tLen target codes.
template <class TPtr>
__device__ __forceinline__ int global_rows(const int lane, const int qLen, const int tLen, const int w, const SwScoring& sc,
                                           TPtr tg, int2* __restrict__ eh, const int8_t* __restrict__ qp,
                                           uint8_t* __restrict__ z, const int nCol) {
  const int oDel = sc.o_del, eDel = sc.e_del, oIns = sc.o_ins, eIns = sc.e_ins;
  const int oeDel = oDel + eDel, oeIns = oIns + eIns;
  for (int i = 0; i < tLen; ++i) {
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
  return uni(eh[qLen].x);
}

// Backtrack, SWUtil.scala:355-382 (every lane walks the same cells: uniform loads).  Leaves the operations in
// cig[0..n) from the LAST one to the first (len<<4 | op) and returns n (entries beyond CIG_LDS are not stored).
__device__ __forceinline__ int global_backtrack(const int lane, const int qLen, const int tLen, const int w,
                                                const uint8_t* __restrict__ z, const int nCol, uint32_t* __restrict__ cig) {
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
  return n;
}

}  // namespace
}  // namespace bpsw
