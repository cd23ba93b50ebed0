// bpsw_chain2aln.hip -- the memChainToAlnBatched round loop on the device, gfx950.
//
// What it computes: for every read of a batch, the regions memChainToAlnBatched (MemChainToAlignBatched.scala:380-616)
// leaves in regArrays before memSortAndDedup -- chains in order; inside a chain the seeds from the longest down
// (srt, :366-373); testExtension (:680-741) against every region the read has so far; checkOverlapping (:753-787);
// extension() (:789-883) on windows of the reference cut by getMaxSpan (:648-676) + bnsGetSeq; computeSeedCoverage
// (:891-907).  The Scala batches "one seed per read per round" through one JNI call per round; per read that is a
// purely sequential walk, so here ONE WAVEFRONT owns one read and walks all its rounds back to back: the extension
// tasks are never materialised (no wire batch, no nibble packing, no per-round launch), the query comes straight from
// the read bytes and the target straight from the 2-bit reference resident in HBM (bpsw_ref_load).
//
// The DP itself is the same device code as ext_kernel (bpsw_extend_core.h).  Everything else is wave-uniform control:
// the seed loops that are naturally parallel (max span, srt ranks, seed coverage) use the 64 lanes, the sequential tests
// run once per wave.  The read's regions are written to global memory as they are created; a compact copy of the first
// 64 sits in LDS for testExtension.
#include <stdlib.h>

#include "bpsw_extend_core.h"

namespace bpsw {
namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int C2A_TCAP = 768;   // staged target rows per side: qLen + (w << 1) + 2 <= 256 + 508 + 2 (reads <= 256 bases, w <= 254)
constexpr int C2A_RCAP = 64;    // regions of the current read cached in LDS
constexpr int SRT_MARKED = -2;  // MemChainToAlignBatched.scala:51

__device__ __forceinline__ int dtoi_sat(double x) {  // Scala's Double.toInt: truncation, saturating
  if (x >= 2147483647.0) return 2147483647;
  if (x <= -2147483648.0) return (int)0x80000000;
  return (int)x;
}
__device__ __forceinline__ int pac_base(const uint8_t* __restrict__ pac, const long long l_pac, const long long pos) {
  const bool rev = pos >= l_pac;  // bnsGetSeq, util/BNTSeqUtil.scala:56-73
  const long long k = rev ? (l_pac << 1) - 1 - pos : pos;
  const int b = (pac[k >> 2] >> ((~k & 3) << 1)) & 3;
  return rev ? 3 - b : b;
}
__device__ __forceinline__ long long wave_min64(long long v) {
  for (int o = 32; o > 0; o >>= 1) { const long long t = __shfl_xor(v, o); v = t < v ? t : v; }
  return v;
}
__device__ __forceinline__ long long wave_max64(long long v) {
  for (int o = 32; o > 0; o >>= 1) { const long long t = __shfl_xor(v, o); v = t > v ? t : v; }
  return v;
}
__device__ __forceinline__ int wave_sum(int v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// query source: the read bytes, walked forwards (right side) or backwards (left side, MemChainToAlignBatched.scala:505-510)
struct ReadQ {
  const uint8_t* __restrict__ q;
  int start, step;
  __device__ __forceinline__ int operator()(int j) const {
    const int c = q[start + step * j];
    return c > 4 ? 4 : c;
  }
};

struct RegLite {  // what testExtension reads of a region
  long long rb, re;
  int qb, qe;
};

#ifndef BPSW_C2A_WAVES_PER_SIMD
#define BPSW_C2A_WAVES_PER_SIMD 3  // 141 VGPRs without a bound; a budget for 4 spills 8 registers (measured below)
#endif
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, BPSW_C2A_WAVES_PER_SIMD) void chain2aln_kernel(const ChainBatchDev B, const ChainParams P,
                                                                          bpsw_alnreg_t* __restrict__ out_regs,
                                                                          int32_t* __restrict__ out_cnt,
                                                                          int32_t* __restrict__ srt_scratch,
                                                                          const int srt_per_wave, int* __restrict__ next_read) {
  __shared__ __align__(16) uint8_t ts_all[WAVES_PER_BLOCK][C2A_TCAP];
  __shared__ RegLite regs_all[WAVES_PER_BLOCK][C2A_RCAP];
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  uint8_t* ts = ts_all[wave];
  RegLite* rl = regs_all[wave];
  int32_t* srt = srt_scratch + (size_t)uni((int)(blockIdx.x * WAVES_PER_BLOCK + wave)) * srt_per_wave;

  const int a = P.a, oDel = P.o_del, eDel = P.e_del, oIns = P.o_ins, eIns = P.e_ins, w0 = P.w;
  // eIns >= 1 is required by the entry point, so oIns + eIns > 0: the register path of bpsw_extend_core.h always applies
  auto cal_max_gap = [&](int qlen) -> int {  // MemChainToAlignBatched.scala:625-643
    const int ld = dtoi_sat((double)(qlen * a - oDel) / (double)eDel + 1.0);
    const int li = dtoi_sat((double)(qlen * a - oIns) / (double)eIns + 1.0);
    int len = ld > li ? ld : li;
    if (len <= 1) len = 1;
    const int tmp = w0 << 1;
    return len < tmp ? len : tmp;
  };

  for (;;) {
    const int r = dequeue_task(next_read);
    if (r >= B.n_reads) break;
    const int qlen = uni(B.read_len[r]);
    const uint8_t* query = B.read_pool + B.read_off[r];
    bpsw_alnreg_t* regs = out_regs + B.reg_base[r];
    const int chain0 = uni(B.chain_base[r]), nchains = uni(B.chain_cnt[r]);
    int nreg = 0;

    for (int c = chain0; c < chain0 + nchains; ++c) {
      const int ns = uni(B.seed_cnt[c]);
      if (ns == 0) continue;
      const long long s0 = B.seed_base[c];
      const long long* __restrict__ s_rb = B.seed_rbeg + s0;
      const int* __restrict__ s_qb = B.seed_qbeg + s0;
      const int* __restrict__ s_ln = B.seed_len + s0;

      // ---- getMaxSpan, MemChainToAlignBatched.scala:648-676 ----
      long long bmin = B.l_pac << 1, emax = 0;
      for (int i = lane; i < ns; i += 64) {
        const long long rb = s_rb[i];
        const int qb = s_qb[i], ln = s_ln[i];
        const long long b = rb - (qb + cal_max_gap(qb));
        const long long e = rb + ln + (qlen - qb - ln) + cal_max_gap(qlen - qb - ln);
        bmin = b < bmin ? b : bmin;
        emax = e > emax ? e : emax;
      }
      long long rmax0 = wave_min64(bmin), rmax1 = wave_max64(emax);
      if (rmax0 <= 0) rmax0 = 0;
      if (rmax1 >= (B.l_pac << 1)) rmax1 = B.l_pac << 1;
      if (rmax0 < B.l_pac && B.l_pac < rmax1) {  // crossing the strands: keep the side of seed 0
        if (s_rb[0] < B.l_pac) rmax1 = B.l_pac; else rmax0 = B.l_pac;
      }
      // ---- srt = seeds sorted by (len, index), :366-373: every key is unique, so a rank is a position ----
      for (int i = lane; i < ns; i += 64) {
        const int li = s_ln[i];
        int rank = 0;
        for (int j = 0; j < ns; ++j) {
          const int lj = s_ln[j];
          rank += (lj < li || (lj == li && j < i)) ? 1 : 0;
        }
        srt[rank] = i;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");

      for (int k = ns - 1; k >= 0; --k) {  // one "round" per seed, longest first
        const int si = uni(srt[k]);
        const long long srb = s_rb[si];
        const int sqb = uni(s_qb[si]), sl = uni(s_ln[si]);

        // ---- testExtension, :680-741 ----
        int ext = nreg;
        for (int i = 0; i < nreg; ++i) {
          RegLite p;
          if (i < C2A_RCAP) p = rl[i];
          else { p.rb = regs[i].rb; p.re = regs[i].re; p.qb = regs[i].qb; p.qe = regs[i].qe; }
          if (srb >= p.rb && srb + sl <= p.re && sqb >= p.qb && sqb + sl <= p.qe) {
            int qd = sqb - p.qb;
            long long rd = srb - p.rb;
            int mind = qd < rd ? qd : (int)rd;
            int mg = cal_max_gap(mind);
            int w = mg < w0 ? mg : w0;
            if (qd - rd < w && rd - qd < w) { ext = i; break; }
            qd = p.qe - (sqb + sl);
            rd = p.re - (srb + sl);
            mind = qd < rd ? qd : (int)rd;
            mg = cal_max_gap(mind);
            w = mg < w0 ? mg : w0;
            if (qd - rd < w && rd - qd < w) { ext = i; break; }
          }
        }
        ext = uni(ext);
        if (ext < nreg) {  // ---- checkOverlapping, :753-787 ----
          int ovl = ns;
          for (int i = k + 1; i < ns; ++i) {
            const int ti = uni(srt[i]);
            if (ti == SRT_MARKED) continue;
            const int tq = s_qb[ti], tl = s_ln[ti];
            const long long trb = s_rb[ti];
            if ((double)tl >= (double)sl * 0.95) {
              if (sqb <= tq && sqb + sl - tq >= (sl >> 2) && (long long)(tq - sqb) != trb - srb) { ovl = i; break; }
              if (tq <= sqb && tq + tl - sqb >= (sl >> 2) && (long long)(sqb - tq) != srb - trb) { ovl = i; break; }
            }
          }
          if (uni(ovl) == ns) {  // :482-484: contained and nothing overlapping disagrees -> no extension
            srt[k] = SRT_MARKED;  // every lane stores the same value: no lane-dependent branch near the cross-lane code
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            continue;
          }
        }

        // ---- the new region, :486-499, and its extension, :500-599 + extension() :789-883 ----
        int score = sl * a, truesc = sl * a, qb = 0, qe = qlen, width = w0;
        long long rb = srb, re = srb + sl;
        if (sqb > 0 || sqb + sl != qlen) {
          const int lq = sqb, rq = qlen - (sqb + sl);
          int awSide = w0, awMax = w0;  // the band tried last on this side / the widest over both sides (no array: a dynamically indexed one lives in scratch memory)
          int regScore = sl * a;
          int outQBeg = 0, outRBeg = 0, outQEnd = rq, outREnd = 0, trueScore = regScore, sc = -1;
          for (int side = 0; side < 2; ++side) {
            const int qLen = side ? rq : lq;
            if (qLen <= 0) continue;
            const long long rLen64 = side ? rmax1 - (srb + sl) : srb - rmax0;
            const int rLen = (int)(rLen64 < 0 ? 0 : (rLen64 > 0x7fffffff ? 0x7fffffff : rLen64));
            const int penClip = side ? P.pen_clip3 : P.pen_clip5;
            const int maxIns = max(1, dtoi_sat((double)(qLen * P.mat_max + penClip - oIns) / (double)eIns + 1.0));  // SWUtil.scala:110-115
            const int maxDel = max(1, dtoi_sat((double)(qLen * P.mat_max + penClip - oDel) / (double)eDel + 1.0));
            const int hInit = side ? regScore : sl * a;
            const int sc0 = regScore;
            const ReadQ qsrc = {query, side ? sqb + sl : sqb - 1, side ? 1 : -1};
            const long long tpos = side ? srb + sl : srb - 1;  // left target walks backwards, :511-517
            const int tstep = side ? 1 : -1;
            // Row i needs i - w <= qLen, so at most qLen + w + 1 rows of a side are ever swept (the row at
            // i = qLen + w has an empty band and ends the call): stage only those.
            const int tstage = min(rLen, qLen + (w0 << 1) + 2);
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < tstage; i += 64) ts[i] = (uint8_t)(8 * pac_base(B.pac, B.l_pac, tpos + (long long)tstep * i));
            __builtin_amdgcn_wave_barrier();
            ExtRes x = {0, 0, 0, 0, 0, 0};
            // near-exact flank: the DP result is known without running it (bpsw_extend_core.h, flank_closed_form)
            const int oe_min = min(oIns + eIns, oDel + eDel);
            const auto tsrc = [ts](int j) { return (int)(ts[j] >> 3); };
            const bool exact = P.exact_a > 0 && oe_min > 0 && w0 >= 2 &&
                               ((rLen >= qLen && tstage >= qLen &&
                                 flank_closed_form(lane, qLen, min(rLen, tstage), qsrc, tsrc, P.mat, hInit, P.exact_a, oDel, eDel, oIns, eIns,
                                                   P.zdrop, P.certify, &x)) ||
                                (P.certify >= 3 && flank_start_gap_form(lane, qLen, min(rLen, tstage), qsrc, tsrc, P.mat, hInit, P.exact_a, oDel,
                                                                        eDel, oIns, eIns, P.zdrop, w0, &x)));
            if (exact) {
              awSide = w0;
              regScore = x.max;
            }
            for (int i = 0; i < 2 && !exact; ++i) {  // MAX_BAND_TRY
              const int prev = regScore;
              awSide = w0 << i;
              const int w = min(min(awSide, maxIns), maxDel);
              const int tl = min(rLen, qLen + w + 2);
              int oInsT = oIns, eInsT = eIns;  // opaque: keeps the per-lane column constants of every slot count out of long-lived VGPRs
              asm volatile("" : "+s"(oInsT), "+s"(eInsT));
              x = sw_extend_reg_any(lane, qLen, tl, qsrc, ts, P.mat, oDel, eDel, oInsT, eInsT, w, P.zdrop, P.zmode, hInit, P.tail_bound ? P.mat_max : 0);
              regScore = x.max;
              if (regScore == prev || x.max_off < (awSide >> 1) + (awSide >> 2)) break;
            }
            sc = regScore;
            awMax = max(awMax, awSide);
            const bool local = x.gscore <= 0 || x.gscore <= regScore - penClip;
            if (side == 0) {
              outQBeg = local ? sqb - x.qle : 0;
              outRBeg = local ? -x.tle : -x.gtle;
              trueScore = local ? regScore : x.gscore;
            } else {
              outQEnd = local ? x.qle : rq;
              outREnd = local ? x.tle : x.gtle;
              trueScore += (local ? regScore : x.gscore) - sc0;
            }
          }
          qb = outQBeg; rb = outRBeg + srb;  // :590-599
          qe = outQEnd + sqb + sl; re = outREnd + srb + sl;
          score = sc; truesc = trueScore; width = awMax;
        }
        // ---- computeSeedCoverage, :891-907 ----
        int cov = 0;
        for (int i = lane; i < ns; i += 64) {
          const int tq = s_qb[i], tl = s_ln[i];
          const long long trb = s_rb[i];
          if (tq >= qb && tq + tl <= qe && trb >= rb && trb + tl <= re) cov += tl;
        }
        cov = wave_sum(cov);
        {  // written by every lane with identical values (see above)
          bpsw_alnreg_t o;
          o.rb = rb; o.re = re; o.qb = qb; o.qe = qe; o.score = score; o.truesc = truesc; o.sub = 0; o.csub = 0; o.sub_n = 0;
          o.w = width; o.seedcov = cov; o.secondary = 0; o.hash = 0;  // MemAlnRegType.scala:26-38 defaults
          regs[nreg] = o;
          if (nreg < C2A_RCAP) { RegLite t; t.rb = rb; t.re = re; t.qb = qb; t.qe = qe; rl[nreg] = t; }
        }
        ++nreg;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      }
    }
    out_cnt[r] = nreg;
  }
}

}  // namespace

int chain2aln_resident_waves(int num_cu) { return num_cu * 8 * WAVES_PER_BLOCK; }

hipError_t launch_chain2aln_kernel(const ChainBatchDev& B, const ChainParams& P, bpsw_alnreg_t* d_out_regs, int32_t* d_out_cnt,
                                   int32_t* d_srt_scratch, int srt_per_wave, int num_cu, int* d_counter, hipStream_t s) {
  if (B.n_reads <= 0) return hipSuccess;
  int blocks = (B.n_reads + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  static const double cap = getenv("BPSW_C2A_BLOCKS_PER_CU") ? atof(getenv("BPSW_C2A_BLOCKS_PER_CU")) : 8.0;
  int max_blocks = (int)(num_cu * (cap > 8.0 ? 8.0 : cap));
  if (max_blocks < 1) max_blocks = 1;
  if (blocks > max_blocks) blocks = max_blocks;
  hipError_t e = hipMemsetAsync(d_counter, 0, sizeof(int), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(chain2aln_kernel, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, s, B, P, d_out_regs, d_out_cnt, d_srt_scratch,
                     srt_per_wave, d_counter);
  return hipGetLastError();
}

}  // namespace bpsw
