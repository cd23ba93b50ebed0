// bpsw_extend_lane.hip -- "lane-per-task" banded extension: SIXTY-FOUR extension tasks per wavefront, gfx950.
//
// Same computation as ext_kernel (bpsw_extend.hip): extension() of MemChainToAlignBatched.scala:789-883 over
// SWExtend (SWUtil.scala:61-230), bit-exact.  Why a third formulation: the wave-per-task kernels spend ~4 wave
// instructions per DP cell, almost all of it per-row control and cross-lane scans (a row touches only ~44 cells), and
// both are bound by instruction issue.  Here every lane runs the textbook row/column loops of SWExtend on its OWN
// task, so one wave instruction advances up to 64 cells and there is no cross-lane operation at all:
//   * the (h,e) row of a lane lives in LDS as one 32-bit word per column (h<<16 | e; both are >= 0 and small),
//     column-major `eh[j][lane]`, so the 64 lanes of an access hit 64 consecutive words whatever their j;
//   * the query bases of both sides are staged once per task as bytes `qb[k][lane]`; the score is one v_perm_b32 byte
//     select from the two score words of the row's target base (biased by 128, the bias is folded into the add);
//   * the two trimming loops of SWUtil.scala:202-214 are evaluated while the row is swept (last zero column before
//     the running LAST arg-max, first zero column after it), so no second pass over the band;
//   * left/right sides and band retries are a per-lane state machine (as in the quad-task kernel): a lane that ends
//     a call sets up the next one while the others keep sweeping rows.
// Tasks are sorted by left query length so the 64 tasks of a wave have similar loop trip counts
// (ext_lane_sort_kernel).  Tasks that do not fit (side > 255 bp, scores that could exceed 16 bits, negative h0) go to
// ext_kernel through the fallback list.
#include <stdlib.h>

#include <atomic>

#include "bpsw_internal.h"

namespace bpsw {
namespace {

constexpr int LANE_EMAX = 255;       // longest side (columns 0..255 -> 256 eh words per lane)
constexpr int LANE_SCORE_MAX = 60000;  // h and e travel as unsigned 16-bit halves
constexpr int ZNONE = -(1 << 24), ZBIG = 1 << 24;

__device__ __forceinline__ int lo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t v) { return (int)(int16_t)(v >> 16); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

enum { PH_SETUP = 0, PH_ROWS = 1, PH_DONE = 2 };

__global__ __launch_bounds__(64) void ext_lane_kernel(const uint32_t* __restrict__ wire, const int* __restrict__ list,
                                                      const int n_list, int16_t* __restrict__ out, const ExtScoring sc,
                                                      const int e_cols) {
  extern __shared__ __align__(16) uint32_t smem[];
  uint2* tab = reinterpret_cast<uint2*>(smem);                          // [5] biased score words per target base
  uint32_t* eh = smem + 16;                                             // [e_cols][64]
  uint8_t* qb = reinterpret_cast<uint8_t*>(smem + 16 + 64 * e_cols);    // [q_cap][64]
  const int lane = threadIdx.x;
  if (lane < 5) {  // row t of the matrix: byte q = mat[t*5 + q]; +128 == ^0x80 per byte
    const unsigned long long r = sc.mat.row[lane];
    tab[lane] = make_uint2((uint32_t)(r & 0xffffffffull) ^ 0x80808080u, (uint32_t)((r >> 32) & 0xffull) ^ 0x80u);
  }
  __builtin_amdgcn_wave_barrier();

  // header, MemChainToAlignBatched.scala:78-84 (signed bytes)
  const uint32_t hdr0 = wire[0], hdr1 = wire[1];
  const int oDel = (int8_t)(hdr0 & 0xff), eDel = (int8_t)((hdr0 >> 8) & 0xff);
  const int oIns = (int8_t)((hdr0 >> 16) & 0xff), eIns = (int8_t)((hdr0 >> 24) & 0xff);
  const int penClip5 = (int8_t)(hdr1 & 0xff), penClip3 = (int8_t)((hdr1 >> 8) & 0xff);
  const int wBand = (int8_t)((hdr1 >> 16) & 0xff);
  const int oeDel = oDel + eDel, oeIns = oIns + eIns;
  const int zdrop = sc.zdrop, zmode = sc.zdrop_mode;

  const int slot = blockIdx.x * 64 + lane;
  const bool have = slot < n_list;
  const int task = list[have ? slot : 0];
  const uint32_t* rec = wire + 8 + 8 * (size_t)task;  // MemChainToAlignBatched.scala:95-117
  const uint4 ra = *reinterpret_cast<const uint4*>(rec), rb = *reinterpret_cast<const uint4*>(rec + 4);
  const int lq = lo16(ra.x), lr = hi16(ra.x), rq = lo16(ra.y), rr = hi16(ra.y);
  const uint32_t* words = wire + (size_t)(int)ra.z;
  const int regScore0 = lo16(ra.w), qBegT = hi16(ra.w), h0T = lo16(rb.x);
  const int lMaxIns = max(1, lo16(rb.y)), lMaxDel = max(1, hi16(rb.y));  // SWUtil.scala:110-115
  const int rMaxIns = max(1, lo16(rb.z)), rMaxDel = max(1, hi16(rb.z));

  // stage the query bases of both sides (leftQs then rightQs, MemChainToAlignBatched.scala:125-145) as bytes
  {
    const int nq = have ? lq + rq : 0;
    uint8_t* q = qb + lane;
    for (int k = 0; k < nq; k += 8, q += 8 * 64) {
      const uint32_t wv = words[k >> 3];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const uint32_t c = (wv >> (28 - 4 * t)) & 0xFu;
        q[t * 64] = (uint8_t)(c > 4u ? 4u : c);  // codes are 0..4 (LocusEncode); never select outside the score words
      }
    }
  }

  // ---- per-task state ----
  int phase = have ? PH_SETUP : PH_DONE;
  int side = 0, tryi = 0, regScore = regScore0, prev = 0, sc0 = 0, aw0 = wBand, aw1 = wBand;
  int outQBeg = 0, outRBeg = 0, outQEnd = rq, outREnd = 0, trueScore = regScore0, score = -1;  // C2AB:790-807
  int qLen = 0, tLen = 0, qoff = 0, toff = 0, w = 0, h0c = 0, penClip = 0, awCur = 0;
  int i = 0, beg = 0, end = 0, mx = 0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, h1raw = 0;
  uint32_t tw = 0;

  for (;;) {
    if (phase == PH_SETUP) {
      if (side == 0 && lq <= 0) side = 1;  // MemChainToAlignBatched.scala:809 / :844
      if (side == 1 && rq <= 0) side = 2;
      if (side == 2) {  // task complete: 10 int16, MemChainToAlignBatched.scala:181-188, :877-879
        uint32_t* o = reinterpret_cast<uint32_t*>(out + 10 * (size_t)task);
        const int width = aw0 > aw1 ? aw0 : aw1;
        o[0] = rb.w;
        o[1] = ((uint32_t)outQBeg & 0xffffu) | ((uint32_t)outQEnd << 16);
        o[2] = ((uint32_t)outRBeg & 0xffffu) | ((uint32_t)outREnd << 16);
        o[3] = ((uint32_t)score & 0xffffu) | ((uint32_t)trueScore << 16);
        o[4] = (uint32_t)width & 0xffffu;
        phase = PH_DONE;
      } else {
        qLen = side ? rq : lq;
        tLen = side ? rr : lr;
        qoff = side ? lq : 0;
        toff = side ? lq + rq + lr : lq + rq;
        penClip = side ? penClip3 : penClip5;
        if (tryi == 0) sc0 = regScore;      // MemChainToAlignBatched.scala:847
        h0c = side ? sc0 : h0T;             // left: h0; right: the score after the left extension
        prev = regScore;
        awCur = wBand << tryi;
        if (side) aw1 = awCur; else aw0 = awCur;
        w = min(min(awCur, side ? rMaxIns : lMaxIns), side ? rMaxDel : lMaxDel);
        // row -1 (SWUtil.scala:75-78, 97-104): eh[0].h = h0, eh[j].h = max(0, h0 - oeIns - (j-1)*eIns), e = 0
        uint32_t* ep = eh + lane;
        int hv = h0c - oeIns + eIns;
        for (int j = 0; j <= qLen; ++j, ep += 64, hv -= eIns) *ep = (uint32_t)(j == 0 ? h0c : max(0, hv)) << 16;
        i = 0; beg = 0; end = qLen;
        mx = h0c; max_i = -1; max_j = -1; max_ie = -1; gscore = -1; max_off = 0;  // SWUtil.scala:118-125
        h1raw = h0c - oDel;
        phase = PH_ROWS;
      }
    }
    if (phase == PH_ROWS) {
      bool fin = true;
      if (i < tLen) {
        // ---------------------------------------------------------------- one DP row, SWUtil.scala:129-220
        const int tpos = toff + i;
        if ((tpos & 7) == 0 || i == 0) tw = words[tpos >> 3];
        int tc = (int)((tw >> (28 - 4 * (tpos & 7))) & 0xFu);
        tc = tc > 4 ? 4 : tc;
        const uint2 sw = tab[tc];
        h1raw -= eDel;
        int h1 = max(0, h1raw);                 // SWUtil.scala:137-138
        beg = max(beg, i - w);                  // SWUtil.scala:140-142
        end = min(min(end, i + w + 1), qLen);
        int f = 0, mm = 0, mj = -1;
        // trimming bookkeeping: a "zero column" c+1 is a cell c with H(i,c) == 0; column beg is zero when h1 == 0
        int zlast = h1 == 0 ? beg - 1 : ZNONE;  // last zero cell seen so far
        int zl = ZNONE;                          // ... as of the last arg-max update
        int zr = ZBIG;                           // first zero cell after the last arg-max
        {
          uint32_t* ep = eh + beg * 64 + lane;
          const uint8_t* qp = qb + (qoff + beg) * 64 + lane;
          for (int j = beg; j < end; ++j, ep += 64, qp += 64) {  // SWUtil.scala:145-172
            const uint32_t wv = *ep;
            const uint32_t qv = *qp;
            int h = (int)(wv >> 16);
            const int e = (int)(wv & 0xffffu);
            const int sb = (int)__builtin_amdgcn_perm(sw.y, sw.x, qv | 0x0c0c0c00u);  // mat[t*5 + q] + 128
            h = h + sb - 128;
            h = max3i(h, e, f);
            const bool U = mm <= h;  // LAST arg-max, SWUtil.scala:158-161
            const bool Z = h == 0;
            zr = min(zr, Z ? j : ZBIG);
            zr = U ? ZBIG : zr;
            zl = U ? zlast : zl;
            zlast = Z ? j : zlast;
            mj = U ? j : mj;
            mm = max(mm, h);
            const int en = max3i(e - eDel, h - oeDel, 0);
            f = max3i(f - eIns, h - oeIns, 0);
            *ep = ((uint32_t)h1 << 16) | (uint32_t)en;
            h1 = h;
          }
          eh[end * 64 + lane] = (uint32_t)h1 << 16;  // SWUtil.scala:174-175
        }
        // SWUtil.scala:177-182: j after the column loop is end (or beg for an empty band)
        if ((beg < end ? end : beg) == qLen && gscore <= h1) { max_ie = i; gscore = h1; }
        bool stop = mm == 0;  // SWUtil.scala:184-185
        if (!stop) {
          if (mm > mx) {  // SWUtil.scala:187-193
            mx = mm; max_i = i; max_j = mj;
            const int d = mj - i;
            max_off = max3i(max_off, d, -d);
          } else if (zdrop > 0) {  // SWUtil.scala:194-199 (Scala parse) / native/ksw.c:455-461 (BWA parse)
            const int di = i - max_i, dj = mj - max_j;
            const bool A = di > dj;
            const bool B = mx - mm - (di - dj) * eDel > zdrop;
            const bool C = mx - mm - (dj - di) * eIns > zdrop;
            stop = zmode == BPSW_ZDROP_SCALA ? (A && (B || C)) : (A ? B : C);
          }
        }
        if (!stop) {  // band trimming, SWUtil.scala:202-214
          beg = zl > ZNONE / 2 ? zl + 2 : beg;
          end = zr < ZBIG / 2 ? zr + 1 : end + 1;
          i += 1;
          fin = i >= tLen;
        }
      }
      if (fin) {  // ---------------------------------------------------- a call ends: band retry / next side
        regScore = mx;
        const int qle = max_j + 1, tle = max_i + 1, gtle = max_ie + 1;  // SWUtil.scala:222-227
        const bool again = tryi == 0 && !(regScore == prev || max_off < (awCur >> 1) + (awCur >> 2));  // C2AB:821,858
        if (again) {
          tryi = 1;
        } else {
          score = regScore;
          const bool local = gscore <= 0 || gscore <= regScore - penClip;  // C2AB:829, :866
          if (side == 0) {
            outQBeg = local ? qBegT - qle : 0;
            outRBeg = local ? -tle : -gtle;
            trueScore = local ? regScore : gscore;
          } else {
            outQEnd = local ? qle : rq;
            outREnd = local ? tle : gtle;
            trueScore += (local ? regScore : gscore) - sc0;
          }
          side += 1;
          tryi = 0;
        }
        phase = PH_SETUP;
      }
    }
    if (phase == PH_DONE) break;
  }
}

// ---- sort: which tasks the lane kernel takes, ordered by left query length ---------------------------------------
// One workgroup (a 32 k-task table is 1 MB; the sort is ~20 us): LDS histogram over the key, exclusive scan, scatter.
// counts[0] = tasks for the lane kernel, [1] = fallback tasks, [2] = max side among lane tasks, [3] = max lq+rq.
__global__ __launch_bounds__(1024) void ext_lane_sort_kernel(const uint32_t* __restrict__ wire, const int n_tasks,
                                                             const int mat_max, int* __restrict__ lane_list,
                                                             int* __restrict__ fb_list, int* __restrict__ counts) {
  __shared__ int hist[LANE_EMAX + 1];
  __shared__ int nfb, maxe, maxq;
  for (int k = threadIdx.x; k <= LANE_EMAX; k += blockDim.x) hist[k] = 0;
  if (threadIdx.x == 0) { nfb = 0; maxe = 0; maxq = 0; }
  __syncthreads();
  const int pos_max = mat_max > 0 ? mat_max : 0;
  auto key_of = [&](int t) -> int {
    const uint32_t* rec = wire + 8 + 8 * (size_t)t;
    const int lq = lo16(rec[0]), rq = lo16(rec[1]);
    const int reg = lo16(rec[3]), h0 = lo16(rec[4]);
    const int hi = max(h0, reg);
    const bool ok = max(lq, rq) <= LANE_EMAX && h0 >= 0 && reg >= 0 && hi + (lq + rq) * pos_max <= LANE_SCORE_MAX;
    return ok ? max(lq, 0) : -1;
  };
  for (int t = threadIdx.x; t < n_tasks; t += blockDim.x) {
    const int key = key_of(t);
    if (key >= 0) {
      atomicAdd(&hist[key], 1);
      const uint32_t* rec = wire + 8 + 8 * (size_t)t;
      const int lq = max(lo16(rec[0]), 0), rq = max(lo16(rec[1]), 0);
      atomicMax(&maxe, max(lq, rq));
      atomicMax(&maxq, lq + rq);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // exclusive scan, longest left side first
    int run = 0;
    for (int k = LANE_EMAX; k >= 0; --k) { const int c = hist[k]; hist[k] = run; run += c; }
    counts[0] = run;
    counts[2] = maxe;
    counts[3] = maxq;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < n_tasks; t += blockDim.x) {
    const int key = key_of(t);
    if (key >= 0) lane_list[atomicAdd(&hist[key], 1)] = t;
    else fb_list[atomicAdd(&nfb, 1)] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[1] = nfb;
}

}  // namespace

void launch_ext_lane_sort(const uint32_t* d_wire, int n_tasks, int mat_max, int* d_lane_list, int* d_fb_list, int* d_counts,
                          hipStream_t s) {
  hipLaunchKernelGGL(ext_lane_sort_kernel, dim3(1), dim3(1024), 0, s, d_wire, n_tasks, mat_max, d_lane_list, d_fb_list, d_counts);
}

size_t ext_lane_lds_bytes(int max_side, int max_qsum) {
  const size_t e_cols = (size_t)max_side + 1;
  const size_t q_cap = ((size_t)max_qsum + 7) & ~(size_t)7;
  return 64 + 256 * e_cols + 64 * q_cap;
}

hipError_t launch_ext_lane_kernel(const uint32_t* d_wire, const int* d_list, int n_list, int16_t* d_out, const ExtScoring& sc,
                                  int max_side, int max_qsum, hipStream_t s) {
  if (n_list <= 0) return hipSuccess;
  if (max_side > LANE_EMAX || max_qsum > 2 * LANE_EMAX) return hipErrorInvalidValue;
  // a few LDS configurations cover all batches (the dynamic size is part of the dispatch)
  max_side = (max_side + 16) & ~15;  // columns 0..max_side
  max_qsum = (max_qsum + 15) & ~15;
  const size_t lds = ext_lane_lds_bytes(max_side - 1, max_qsum);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  // the opt-in to > 64 KB of dynamic LDS is a property of the function ON A DEVICE: remember the largest size per device
  static std::atomic<size_t> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (lds > 64 * 1024 && lds > attr_set[dev].load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ext_lane_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    size_t seen = attr_set[dev].load(std::memory_order_relaxed);
    while (seen < lds && !attr_set[dev].compare_exchange_weak(seen, lds, std::memory_order_relaxed)) {}
  }
  const int blocks = (n_list + 63) / 64;
  hipLaunchKernelGGL(ext_lane_kernel, dim3(blocks), dim3(64), lds, s, d_wire, d_list, n_list, d_out, sc, max_side);
  return hipGetLastError();
}

}  // namespace bpsw
