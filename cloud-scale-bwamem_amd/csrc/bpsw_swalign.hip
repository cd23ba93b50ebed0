// bpsw_swalign.hip -- unbanded local affine-gap SW for the pair-end mate rescue (boundary 1), gfx950.
//
// What it computes: SWUtil.SWAlign2 (SWUtil.scala:583-601) = SWAlign (SWUtil.scala:417-570) forward
// pass -> (score, tEnd, qEnd, second best outside +-ceil(score/a) rows) and, when asked for (KSW_XSTART),
// a reverse pass over the reversed prefixes that yields (tBeg, qBeg).  Bit-exact with the Scala text,
// including the true-DP second best (the SSE2 C differs there).
//
// How.  The band is static, so the row dependency can be skewed: lane l owns C consecutive query columns and at step t works on
// target row t - l (a systolic anti-diagonal); everything a cell needs from its left neighbour -- F(i,j), the diagonal H(i-1,j-1), the
// running row key and the row's target selector -- arrives with one DPP wave_shr:1 from the previous step: no in-row scan, no LDS
// traffic in the inner loop.  Three forms of it live here:
//   * swp_kernel<C> / swp_resident_kernel<C> (what runs): TWO jobs per wavefront, one in each 16-bit half of every DP register
//     (v_pk_*_u16; exact because a pass stops at the 255 score cap), the query right-aligned, the row keys booked by the lanes
//     behind the last column eight rows at a time, the second best rebuilt from the stored row maxima (see "Packed form" below).
//     The resident kernel is the same job pair taken from the device's submission ring (bpsw_ring.h) instead of one launch's table.
//   * sw_kernel<C>: one job per wavefront in int32, the per-row bookkeeping of SWUtil.scala:517-538 on the scalar unit -- for
//     scorings the packed form cannot take (max(mat) > |b| + 1, scores beyond a byte) and mates above 256 bases.
//   * sw4_kernel: four jobs per wavefront (one per 16-lane DPP row, ten columns per lane) in int32, for such scorings in large batches.
#include <stdlib.h>

#include "bpsw_internal.h"
#include "bpsw_wave.h"

#include "bpsw_ring_dev.h"

#include "bpsw_diag_waves.h"
BPSW_DIAG_WAVES_DEFINE(sw)

namespace bpsw {
namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int TBUF = 2048;           // target bases staged in LDS per wave
constexpr int MINUS_INF = -0x40000000;  // SWUtil.scala:28

struct PassRes {
  int max, max_i, max_j, nb;
};

// Where the target rows of a job come from: a byte-per-base window, or the 2-bit reference resident in HBM
// (bnsGetSeq, util/BNTSeqUtil.scala:37-79: coordinates >= l_pac read the reverse-complement strand).
struct TgSrc {
  const uint8_t* bytes;
  const uint8_t* pac;
  long long l_pac, rb;
  __device__ __forceinline__ int at(int src) const {
    if (bytes) return bytes[src];
    const long long pos = rb + src;
    const bool rev = rb >= l_pac;
    const long long k = rev ? (l_pac << 1) - 1 - pos : pos;
    const int b = (pac[k >> 2] >> ((~k & 3) << 1)) & 3;
    return rev ? 3 - b : b;
  }
};

__device__ __forceinline__ int shr1_zero(int src) {  // lane l <- lane l-1 ; lane 0 <- 0
  return __builtin_amdgcn_update_dpp(0, src, DPP_WAVE_SHR1, 0xf, 0xf, true);
}

// One SWAlign pass (SWUtil.scala:417-570) by a whole wave.
//   pass2 == false: columns 0..qCols-1 are the mate (reverse-complemented on the fly when qrev);
//                   rows are target[0..tLen)
//   pass2 == true : the reversed prefixes of SWUtil.scala:588-590: column j is forward column qEnd-j,
//                   row r is target[tEnd-r] for r <= tEnd and target[r] beyond
template <int C>
__device__ PassRes sw_pass(const int lane, const uint8_t* __restrict__ q, const int qLenRaw, const bool qrev,
                           const int qCols, const bool pass2, const int qEnd, const TgSrc tg,
                           const int tLen, const int tEnd, const SwScoring& sc, const int minScore,
                           const int endScore, const int maxScore, uint8_t* __restrict__ tbuf,
                           uint32_t* __restrict__ list) {
  const int eDel = sc.e_del, eIns = sc.e_ins, oeDel = sc.o_del + sc.e_del, oeIns = sc.o_ins + sc.e_ins;
  int prof_lo[C], prof_hi[C], cmask[C], ckey[C], Hp[C], E[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = lane * C + c;
    const bool valid = j < qCols;
    int code = 4;
    if (valid) {
      const int fc = pass2 ? qEnd - j : j;
      const int raw = qrev ? qLenRaw - 1 - fc : fc;
      code = q[raw];
      if (qrev) code = code < 4 ? 3 - code : 4;  // MemSamPe.scala:1178-1181
      if (code > 4) code = 4;
    }
    const int sh = 8 * code;
    prof_lo[c] = (int)(((sc.mat.row[0] >> sh) & 0xff) | (((sc.mat.row[1] >> sh) & 0xff) << 8) |
                       (((sc.mat.row[2] >> sh) & 0xff) << 16) | (((sc.mat.row[3] >> sh) & 0xff) << 24));
    prof_hi[c] = (int)(((sc.mat.row[4] >> sh) & 0xff) | 0x80808000u);  // row 4 = N, codes 5..7 = "no row": -128
    cmask[c] = valid ? -1 : 0;
    ckey[c] = 1023 - j;
    Hp[c] = 0;
    E[c] = 0;
  }
  int hlast_cur = 0, hlast_old = 0, fout = 0, keyout = 0;
  int tcode = 5;
  const int Lq = (qCols - 1) / C;  // lane that holds the last real column
  const int nsteps = tLen + Lq;
  int mx = MINUS_INF, max_i = -1, max_j = -1, nb = 0, lastScore = 0, lastT = -2;

  for (int t = 0; t < nsteps; ++t) {
    if ((t & (TBUF - 1)) == 0) {  // stage the next window of target bases
      int t0 = t;
      asm volatile("" : "+s"(t0));  // opaque: keeps the staging addresses from becoming per-step 64-bit induction variables (3 of 52 VALU per step)
      __builtin_amdgcn_wave_barrier();
      for (int k = lane; k < TBUF; k += 64) {
        const int r = t0 + k;
        int code = 5;
        if (r < tLen) {
          const int src = (pass2 && r <= tEnd) ? tEnd - r : r;
          code = tg.at(src);
          if (code > 4) code = 4;
        }
        tbuf[k] = (uint8_t)code;
      }
      __builtin_amdgcn_wave_barrier();
    }
    const int ch = tbuf[t & (TBUF - 1)];
    tcode = wave_shr1(ch, tcode);                 // lane 0 starts row t, lane l continues row t-l
    const int din = shr1_zero(hlast_old);         // H(i-1, l*C-1)
    const int fin = shr1_zero(fout);              // F(i, l*C)
    const int kin = shr1_zero(keyout);            // row maximum so far
    hlast_old = hlast_cur;
    const bool lo_sel = tcode < 4;
    const unsigned sh = (unsigned)(tcode & 3) * 8u;
    int diag = din, f = fin, key = kin;
#pragma unroll
    for (int c = 0; c < C; ++c) {  // SWUtil.scala:484-505
      const int pw = lo_sel ? prof_lo[c] : prof_hi[c];
      const int s = __builtin_amdgcn_sbfe(pw, sh, 8u);
      const int h = max3i(diag + s, E[c], f) & cmask[c];
      diag = Hp[c];
      Hp[c] = h;
      key = max(key, (h << 10) | ckey[c]);  // first arg-max wins ties (SWUtil.scala:493)
      E[c] = max3i(E[c] - eDel, h - oeDel, 0);
      f = max3i(f - eIns, h - oeIns, 0);
    }
    hlast_cur = Hp[C - 1];
    fout = f;
    keyout = key;

    const int i = t - Lq;  // the row that has just left the pipe
    if (i >= 0) {
      const int skey = __builtin_amdgcn_readlane(keyout, Lq);
      const int m = skey >> 10;
      if (m >= minScore) {  // SWUtil.scala:517-529
        if (nb == 0 || lastT + 1 != i) {
          if (lane == 0) list[nb] = ((uint32_t)m << 16) | (uint32_t)i;
          ++nb; lastScore = m; lastT = i;
        } else if (lastScore < m) {
          if (lane == 0) list[nb - 1] = ((uint32_t)m << 16) | (uint32_t)i;
          lastScore = m; lastT = i;
        }
      }
      if (m > mx) {  // SWUtil.scala:532-538
        mx = m; max_i = i;
        max_j = m ? 1023 - (skey & 1023) : -1;
        if (mx >= endScore || mx >= maxScore) break;
      }
    }
  }
  PassRes r;
  r.max = mx; r.max_i = max_i; r.max_j = max_j; r.nb = nb;
  return r;
}

template <int C>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void sw_kernel(const SwJobsDev jobs, const SwScoring sc,
                                                                    int32_t* __restrict__ out,
                                                                    uint32_t* __restrict__ scratch,
                                                                    const int scratch_per_wave,
                                                                    const SwPrepass* __restrict__ pre) {
  __shared__ uint8_t tbuf_all[WAVES_PER_BLOCK][TBUF];
  // asynchronous entry: launched for a speculated geometry before the table scan was read back (see launch_ext_kernel)
  if (pre && (pre->error != 0 || pre->max_qlen > 64 * C || 4 * ((pre->max_tlen + 63) & ~63) > 4 * scratch_per_wave)) return;
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const int slot = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  uint8_t* tbuf = tbuf_all[wave];
  uint32_t* list = scratch + (size_t)slot * scratch_per_wave;
  const int maxScore = 255 - abs(sc.b);  // SWUtil.scala:423
  const int xtra = sc.xtra;
  const int stride = gridDim.x * WAVES_PER_BLOCK;

  for (int job = slot; job < jobs.n; job += stride) {
    const int qLen = uni(jobs.q_len[job]), tLen = uni(jobs.t_len[job]);
    const uint8_t* q = jobs.q_pool + jobs.q_off[job];
    const long long toff = jobs.t_off[job];
    const TgSrc tg = {jobs.t_pool ? jobs.t_pool + toff : nullptr, jobs.pac, jobs.l_pac, toff};
    const bool qrev = uni((int)jobs.q_rev[job]) != 0;

    const int minScore = (xtra & BPSW_KSW_XSUBO) ? (xtra & 0xffff) : 0x10000;  // SWUtil.scala:434-437
    const int endScore = (xtra & BPSW_KSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
    const PassRes f = sw_pass<C>(lane, q, qLen, qrev, qLen, false, 0, tg, tLen, 0, sc, minScore, endScore, maxScore,
                                 tbuf, list);
    int score = f.max >= maxScore ? 255 : f.max;  // SWUtil.scala:544
    const int te = f.max_i;
    int qe = -1, score2 = -1, te2 = -1, tb = -1, qb = -1;
    if (score != 255) {  // SWUtil.scala:549-567
      qe = f.max_j;
      if (f.nb > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // lane 0's list writes -> all lanes
        const int tmp = (score + sc.a - 1) / sc.a;
        const int low = te - tmp, high = te + tmp;
        int best = -1;
        for (int k = lane; k < f.nb; k += 64) {
          const uint32_t e = list[k];
          const int tE = (int)(e & 0xffffu);
          if (tE < low || tE > high) best = max(best, (int)((e >> 16) << 16) | (0xffff - k));  // first wins ties
        }
        best = wave_max(best);
        if (best >= 0) {
          const int idx = 0xffff - (best & 0xffff);
          score2 = best >> 16;
          te2 = uni((int)(list[idx] & 0xffffu));
        }
      }
    }
    // SWUtil.scala:586-598
    const bool want_start = (xtra & BPSW_KSW_XSTART) && !((xtra & BPSW_KSW_XSUBO) && score < (xtra & 0xffff));
    if (want_start && qe >= 0 && te >= 0) {
      const PassRes r = sw_pass<C>(lane, q, qLen, qrev, qe + 1, true, qe, tg, tLen, te, sc, 0x10000, score & 0xffff,
                                   maxScore, tbuf, list);
      const int rscore = r.max >= maxScore ? 255 : r.max;
      if (score == rscore) {
        tb = te - r.max_i;
        qb = qe - r.max_j;
      }
    }
    if (lane == 0) {
      int32_t* o = out + 7 * (size_t)job;
      o[0] = score; o[1] = te; o[2] = qe; o[3] = score2; o[4] = te2; o[5] = tb; o[6] = qb;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Quad-job form for mates of up to 160 bases (every 2x150 bp rescue): FOUR jobs per wavefront, one per 16-lane DPP row,
// ten query columns per lane.  Same systolic recurrence as sw_pass; `row_shr:1` keeps the four pipes apart by
// construction (lane 0 of a row receives the boundary value), the four jobs advance in lock step, and the per-row
// bookkeeping of SWUtil.scala:517-538 runs on the scalar unit once per job.  Why: a 150-base mate fills only 50 of the 64
// lanes of sw_kernel<3> and pays the per-step overhead (target fetch, four shifts, the row hand-over) for three cells per
// lane; here 60 of 64 lanes carry ten cells each, the pipe fills in 16 steps instead of 50, and the instruction stream
// serves four jobs: about a third fewer vector instructions per job.
constexpr int Q4C = 10;       // query columns per lane: 16 lanes x 10 = 160 columns per job
constexpr int Q4_TBUF = 512;  // target bases staged in LDS per job

__device__ __forceinline__ int row_shr1(int old, int src) {  // lane l of a 16-lane row <- lane l-1; lane 0 of the row <- old
  return __builtin_amdgcn_update_dpp(old, src, DPP_ROW_SHR1, 0xf, 0xf, false);
}
__device__ __forceinline__ int row_shr1_zero(int src) {
  return __builtin_amdgcn_update_dpp(0, src, DPP_ROW_SHR1, 0xf, 0xf, true);
}
template <class T>
__device__ __forceinline__ T sel4(const int g, const T a, const T b, const T c, const T d) {
  return g == 0 ? a : (g == 1 ? b : (g == 2 ? c : d));
}

struct Quartet {  // wave-uniform: the four jobs a wave works on
  int act[4], qLenRaw[4], qrev[4], tLen[4];
  const uint8_t* q[4];
  const uint8_t* tbytes[4];
  long long rb[4];
};

// One SWAlign pass for the four jobs of a quartet in lock step (see sw_pass for the meaning of the arguments).
__device__ void sw4_pass(const int lane, const Quartet& J, const int (&on)[4], const int (&qCols)[4], const bool pass2,
                         const int (&qEnd)[4], const int (&tEnd)[4], const uint8_t* __restrict__ pac, const long long l_pac,
                         const SwScoring& sc, const int minScore, const int (&endScore)[4], const int maxScore,
                         uint8_t* __restrict__ tbuf_wave, uint32_t* const (&list)[4], PassRes (&res)[4]) {
  constexpr int C = Q4C;
  const int grp = lane >> 4, l = lane & 15;
  const int eDel = sc.e_del, eIns = sc.e_ins, oeDel = sc.o_del + sc.e_del, oeIns = sc.o_ins + sc.e_ins;
  // this lane's job
  const int my_on = sel4(grp, on[0], on[1], on[2], on[3]);
  const int my_qCols = sel4(grp, qCols[0], qCols[1], qCols[2], qCols[3]);
  const int my_qEnd = sel4(grp, qEnd[0], qEnd[1], qEnd[2], qEnd[3]);
  const int my_tEnd = sel4(grp, tEnd[0], tEnd[1], tEnd[2], tEnd[3]);
  const int my_tLen = sel4(grp, J.tLen[0], J.tLen[1], J.tLen[2], J.tLen[3]);
  const int my_qLenRaw = sel4(grp, J.qLenRaw[0], J.qLenRaw[1], J.qLenRaw[2], J.qLenRaw[3]);
  const bool my_qrev = sel4(grp, J.qrev[0], J.qrev[1], J.qrev[2], J.qrev[3]) != 0;
  const uint8_t* my_q = sel4(grp, J.q[0], J.q[1], J.q[2], J.q[3]);
  const TgSrc tg = {sel4(grp, J.tbytes[0], J.tbytes[1], J.tbytes[2], J.tbytes[3]), pac, l_pac,
                    sel4(grp, J.rb[0], J.rb[1], J.rb[2], J.rb[3])};
  uint8_t* tbuf = tbuf_wave + grp * Q4_TBUF;

  int prof_lo[C], prof_hi[C], cmask[C], ckey[C], Hp[C], E[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = l * C + c;
    const bool valid = my_on && j < my_qCols;
    int code = 4;
    if (valid) {
      const int fc = pass2 ? my_qEnd - j : j;
      const int raw = my_qrev ? my_qLenRaw - 1 - fc : fc;
      code = my_q[raw];
      if (my_qrev) code = code < 4 ? 3 - code : 4;  // MemSamPe.scala:1178-1181
      if (code > 4) code = 4;
    }
    const int sh = 8 * code;
    prof_lo[c] = (int)(((sc.mat.row[0] >> sh) & 0xff) | (((sc.mat.row[1] >> sh) & 0xff) << 8) |
                       (((sc.mat.row[2] >> sh) & 0xff) << 16) | (((sc.mat.row[3] >> sh) & 0xff) << 24));
    prof_hi[c] = (int)(((sc.mat.row[4] >> sh) & 0xff) | 0x80808000u);
    cmask[c] = valid ? -1 : 0;
    ckey[c] = 1023 - j;
    Hp[c] = 0;
    E[c] = 0;
  }
  int Lq[4], nsteps[4], stop[4], mx[4], max_i[4], max_j[4], nb[4], lastScore[4], lastT[4];
  int maxsteps = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    Lq[g] = on[g] ? (qCols[g] - 1) / C : 0;          // lane (within the row) that holds the last real column
    nsteps[g] = on[g] ? J.tLen[g] + Lq[g] : 0;
    maxsteps = max(maxsteps, nsteps[g]);
    stop[g] = on[g] ? 0 : 1;
    mx[g] = MINUS_INF; max_i[g] = -1; max_j[g] = -1; nb[g] = 0; lastScore[g] = 0; lastT[g] = -2;
  }
  int hlast_cur = 0, hlast_old = 0, fout = 0, keyout = 0;
  int tcode = 5;

  for (int t = 0; t < maxsteps; ++t) {
    if ((t & (Q4_TBUF - 1)) == 0) {  // stage the next window of target bases of each job
      __builtin_amdgcn_wave_barrier();
      for (int k = l; k < Q4_TBUF; k += 16) {
        const int r = t + k;
        int code = 5;
        if (my_on && r < my_tLen) {
          const int src = (pass2 && r <= my_tEnd) ? my_tEnd - r : r;
          code = tg.at(src);
          if (code > 4) code = 4;
        }
        tbuf[k] = (uint8_t)code;
      }
      __builtin_amdgcn_wave_barrier();
    }
    const int ch = tbuf[t & (Q4_TBUF - 1)];
    tcode = row_shr1(ch, tcode);                     // lane 0 of a row starts target row t, lane l continues row t-l
    const int din = row_shr1_zero(hlast_old);        // H(i-1, l*C-1)
    const int fin = row_shr1_zero(fout);             // F(i, l*C)
    const int kin = row_shr1_zero(keyout);           // row maximum so far
    hlast_old = hlast_cur;
    const bool lo_sel = tcode < 4;
    const unsigned sh = (unsigned)(tcode & 3) * 8u;
    int diag = din, f = fin, key = kin;
#pragma unroll
    for (int c = 0; c < C; ++c) {  // SWUtil.scala:484-505
      const int pw = lo_sel ? prof_lo[c] : prof_hi[c];
      const int s = __builtin_amdgcn_sbfe(pw, sh, 8u);
      const int h = max3i(diag + s, E[c], f) & cmask[c];
      diag = Hp[c];
      Hp[c] = h;
      key = max(key, (h << 10) | ckey[c]);  // first arg-max wins ties (SWUtil.scala:493)
      E[c] = max3i(E[c] - eDel, h - oeDel, 0);
      f = max3i(f - eIns, h - oeIns, 0);
    }
    hlast_cur = Hp[C - 1];
    fout = f;
    keyout = key;

    int all_stop = 1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // the row that has just left job g's pipe: SWUtil.scala:517-538, scalar
      if (!stop[g]) {
        const int i = t - Lq[g];
        if (i >= 0) {
          const int skey = __builtin_amdgcn_readlane(keyout, 16 * g + Lq[g]);
          const int m = skey >> 10;
          if (m >= minScore) {
            if (nb[g] == 0 || lastT[g] + 1 != i) {
              if (lane == 16 * g) list[g][nb[g]] = ((uint32_t)m << 16) | (uint32_t)i;
              ++nb[g]; lastScore[g] = m; lastT[g] = i;
            } else if (lastScore[g] < m) {
              if (lane == 16 * g) list[g][nb[g] - 1] = ((uint32_t)m << 16) | (uint32_t)i;
              lastScore[g] = m; lastT[g] = i;
            }
          }
          if (m > mx[g]) {
            mx[g] = m; max_i[g] = i;
            max_j[g] = m ? 1023 - (skey & 1023) : -1;
            if (mx[g] >= endScore[g] || mx[g] >= maxScore) stop[g] = 1;
          }
        }
        if (t + 1 >= nsteps[g]) stop[g] = 1;
      }
      all_stop &= stop[g];
    }
    if (all_stop) break;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) { res[g].max = mx[g]; res[g].max_i = max_i[g]; res[g].max_j = max_j[g]; res[g].nb = nb[g]; }
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, 4) void sw4_kernel(const SwJobsDev jobs, const SwScoring sc,
                                                                     int32_t* __restrict__ out,
                                                                     uint32_t* __restrict__ scratch,
                                                                     const int scratch_per_job,
                                                                     const SwPrepass* __restrict__ pre) {
  __shared__ uint8_t tbuf_all[WAVES_PER_BLOCK][4 * Q4_TBUF];
  if (pre && (pre->error != 0 || pre->max_qlen > 16 * Q4C || ((pre->max_tlen + 63) & ~63) > scratch_per_job)) return;
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const int slot = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  uint8_t* tbuf = tbuf_all[wave];
  const int maxScore = 255 - abs(sc.b);  // SWUtil.scala:423
  const int xtra = sc.xtra;
  const int stride = gridDim.x * WAVES_PER_BLOCK;
  const int minScore = (xtra & BPSW_KSW_XSUBO) ? (xtra & 0xffff) : 0x10000;  // SWUtil.scala:434-437
  const int endScore0 = (xtra & BPSW_KSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
  const int nquad = (jobs.n + 3) >> 2;

  for (int quad = slot; quad < nquad; quad += stride) {
    Quartet J;
    int job[4], on[4], qCols[4], zero4[4] = {0, 0, 0, 0}, end1[4];
    uint32_t* list[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      job[g] = 4 * quad + g;
      on[g] = job[g] < jobs.n ? 1 : 0;
      const int jj = on[g] ? job[g] : 0;
      J.act[g] = on[g];
      J.qLenRaw[g] = uni(jobs.q_len[jj]);
      J.tLen[g] = uni(jobs.t_len[jj]);
      J.qrev[g] = uni((int)jobs.q_rev[jj]);
      J.q[g] = jobs.q_pool + jobs.q_off[jj];
      const long long toff = jobs.t_off[jj];
      J.tbytes[g] = jobs.t_pool ? jobs.t_pool + toff : nullptr;
      J.rb[g] = toff;
      qCols[g] = J.qLenRaw[g];
      end1[g] = endScore0;
      list[g] = scratch + ((size_t)slot * 4 + g) * (size_t)scratch_per_job;
    }
    PassRes f[4];
    sw4_pass(lane, J, on, qCols, false, zero4, zero4, jobs.pac, jobs.l_pac, sc, minScore, end1, maxScore, tbuf, list, f);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the lists written by single lanes -> all lanes
    int score[4], te[4], qe[4], score2[4], te2[4], tb[4], qb[4], on2[4], qCols2[4], end2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      score[g] = f[g].max >= maxScore ? 255 : f[g].max;  // SWUtil.scala:544
      te[g] = f[g].max_i;
      qe[g] = -1; score2[g] = -1; te2[g] = -1; tb[g] = -1; qb[g] = -1;
      if (on[g] && score[g] != 255) {  // SWUtil.scala:549-567
        qe[g] = f[g].max_j;
        if (f[g].nb > 0) {
          const int tmp = (score[g] + sc.a - 1) / sc.a;
          const int low = te[g] - tmp, high = te[g] + tmp;
          int best = -1;
          for (int k = lane; k < f[g].nb; k += 64) {
            const uint32_t e = list[g][k];
            const int tE = (int)(e & 0xffffu);
            if (tE < low || tE > high) best = max(best, (int)((e >> 16) << 16) | (0xffff - k));  // first wins ties
          }
          best = wave_max(best);
          if (best >= 0) {
            const int idx = 0xffff - (best & 0xffff);
            score2[g] = best >> 16;
            te2[g] = uni((int)(list[g][idx] & 0xffffu));
          }
        }
      }
      // SWUtil.scala:586-598
      const bool want_start = (xtra & BPSW_KSW_XSTART) && !((xtra & BPSW_KSW_XSUBO) && score[g] < (xtra & 0xffff));
      on2[g] = (on[g] && want_start && qe[g] >= 0 && te[g] >= 0) ? 1 : 0;
      qCols2[g] = on2[g] ? qe[g] + 1 : 0;
      end2[g] = score[g] & 0xffff;
    }
    if (on2[0] | on2[1] | on2[2] | on2[3]) {
      PassRes r[4];
      sw4_pass(lane, J, on2, qCols2, true, qe, te, jobs.pac, jobs.l_pac, sc, 0x10000, end2, maxScore, tbuf, list, r);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (on2[g]) {
          const int rscore = r[g].max >= maxScore ? 255 : r[g].max;
          if (score[g] == rscore) { tb[g] = te[g] - r[g].max_i; qb[g] = qe[g] - r[g].max_j; }
        }
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (on[g]) {
          int32_t* o = out + 7 * (size_t)job[g];
          o[0] = score[g]; o[1] = te[g]; o[2] = qe[g]; o[3] = score2[g]; o[4] = te2[g]; o[5] = tb[g]; o[6] = qb[g];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Packed form: TWO jobs per wavefront, one in each 16-bit half of every DP register, and no per-row scalar work.
//
// The kernel is bound by the number of instructions a SIMD can issue, vector or scalar.  Two things cut that count:
//  (1) CDNA's packed 16-bit integer ALU ops (v_pk_add_u16, v_pk_sub_u16 with clamp, v_pk_max_u16) apply one cell update to
//      both halves at once, so the same systolic pipe carries two independent jobs for the instruction count of one.  What
//      makes 16 bits exact: a pass stops at the first row whose maximum reaches maxScore = 255 - |b| (SWUtil.scala:423,537),
//      so every H that is ever consumed is <= 254 + max(mat) <= 255 (launch_sw_kernel checks max(mat) <= |b| + 1); E and F are
//      below H, everything is >= 0, and the row key H<<8 | (255 - j) fits the half exactly for up to 256 columns.  The
//      substitution score of both jobs comes from ONE v_perm_b32 per cell: per column each job keeps its four target-base
//      scores as biased bytes (score + bias >= 0), and the per-row selector word (prepared when the target window is staged,
//      handed down the pipe like the target base in sw_pass) picks job A's byte into the low half and job B's into the high
//      half; `max(diag + s, 0)` is the clamped subtraction of the bias.  A window with an N in it (never the case for
//      windows cut from the 2-bit reference) runs a variant of the step that patches the N rows from a fifth score.
//  (2) The sequential per-row bookkeeping of SWUtil.scala:517-538 leaves the loop.  The query is RIGHT-aligned in the
//      pipe: the unused columns sit at the left, where they compute zeros by themselves (no column mask), the last real
//      column is the last column of lane 63 - PK_TAIL, and the PK_TAIL lanes behind it only hand the finished row keys on.
//      So after every group of PK_TAIL + 1 steps the tail lanes hold the keys of PK_TAIL + 1 consecutive rows of both jobs:
//      each tail lane folds its row into a running best (m, earliest row, column -- one unsigned max, frozen once m reaches
//      the stop score) and stores the packed key; the loop body has no readlane and no branch.  The list of local maxima
//      `b` (SWUtil.scala:517-529), which only feeds the second-best score, is rebuilt afterwards from the stored row maxima
//      in closed form (second_best below).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 pk(int x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ int unpk(u16x2 x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ u16x2 pk_max(u16x2 a, u16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ u16x2 pk_subs(u16x2 a, u16x2 b) { return __builtin_elementwise_sub_sat(a, b); }  // max(a - b, 0)

constexpr int PK_TBUF = 256;   // selector words staged in LDS per wave at a time (1 KB)
constexpr int PK_TAIL = 7;     // lanes behind the last query column; rows are booked PK_TAIL + 1 at a time
constexpr int PK_G = PK_TAIL + 1;
constexpr int PK_LAST = 63 - PK_TAIL;  // lane of the last query column
constexpr int PK_SEL_OFF = 12, PK_SEL_N = 13;  // v_perm_b32 selectors: 12 -> 0x00 (row beyond the window), 13 -> 0xff (row is N)

struct Duo {  // wave-uniform: the two jobs a wave works on
  int qLenRaw[2], qrev[2], tLen[2];
  const uint8_t* q[2];
  const uint8_t* tbytes[2];
  long long rb[2];
};

template <int C>
struct PkCols {  // per lane: C columns of both jobs
  int prof[2][C], pn[C], ckey[C], Hp[C], E[C];
  int sel, hlast_cur, hlast_old, fout, keyout;
};

struct PkConst {
  u16x2 eDel, eIns, oeDel, oeIns, biasv, k256;
  bool tail;
  bool same_oe;  // oDel + eDel == oIns + eIns (the default): E and F share `H - oe`
};

// one anti-diagonal of both jobs (SWUtil.scala:484-505)
template <int C, bool HASN, bool SAME_OE>
__device__ __forceinline__ void swp_step(PkCols<C>& S, const PkConst& K, const int ch) {
  S.sel = wave_shr1(ch, S.sel);                    // lane 0 starts a row of each job, lane l continues the row lane l-1 had
  const int din = shr1_zero(S.hlast_old);          // H(i-1, l*C-1)
  const int fin = shr1_zero(S.fout);               // F(i, l*C)
  const int kin = shr1_zero(S.keyout);             // row maxima so far
  S.hlast_old = S.hlast_cur;
  int s[C];
#pragma unroll
  for (int c = 0; c < C; ++c) s[c] = (int)__builtin_amdgcn_perm((unsigned)S.prof[1][c], (unsigned)S.prof[0][c], (unsigned)S.sel);
  if (HASN) {
    const int nm = ((S.sel & 0xff) == PK_SEL_N ? 0xffff : 0) | (((S.sel >> 16) & 0xff) == PK_SEL_N ? (int)0xffff0000 : 0);
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = (s[c] & ~nm) | (S.pn[c] & nm);
  }
  u16x2 diag = pk(din), f = pk(fin), key = pk(kin);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    // (one 32-bit add for both halves: H <= 255 and a biased score byte <= 255 never carry into the other half, and a plain
    // v_add_u32 issues at twice the rate of v_pk_add_u16)
    const u16x2 m = pk_subs(pk((int)((unsigned)unpk(diag) + (unsigned)s[c])), K.biasv);   // max(H(i-1,j-1) + s, 0)
    const u16x2 h = pk_max(pk_max(m, pk(S.E[c])), f);
    diag = pk(S.Hp[c]);
    S.Hp[c] = unpk(h);
    key = pk_max(key, h * K.k256 + pk(S.ckey[c]));                     // first arg-max wins ties (SWUtil.scala:493)
    const u16x2 hd = pk_subs(h, K.oeDel);
    const u16x2 hi = SAME_OE ? hd : pk_subs(h, K.oeIns);
    S.E[c] = unpk(pk_max(pk_subs(pk(S.E[c]), K.eDel), hd));
    f = pk_max(pk_subs(f, K.eIns), hi);
  }
  S.hlast_cur = S.Hp[C - 1];
  S.fout = unpk(f);
  S.keyout = K.tail ? kin : unpk(key);
}

// The second-best score of SWUtil.scala:549-567 from the row maxima m[0..n_rows) of one pass.  The list logic of :517-529
// over the rows with m >= minScore ("hot"): a hot row is WRITTEN (appended, or replaces the last entry) iff the row before was
// not written or m grew, w(r) = hot(r) & (!w(r-1) | m(r) > m(r-1)); an entry's final value is a written row that the next
// row does not replace.  w is a parity chain between the rows where it is forced (not hot -> 0, hot and grown -> 1), so with
// k(r) the last forced row at or before r:  w(r) = hot(r) & (w(k) ^ ((r - k) & 1)).  The best entry outside +-tmp rows of
// te, the earliest on ties, is then one wave maximum.
__device__ void second_best(const int lane, const uint32_t* __restrict__ keys, const int shift, const int base, const int n_rows,
                            const int minScore, const int low, const int high, int& score2, int& te2) {
  int best = -1, carry = -1;
  for (int r0 = 0; r0 < n_rows; r0 += 64) {
    const int r = r0 + lane;
    const bool in = r < n_rows;
    const int mc = in ? (int)((keys[base + r] >> shift) & 0xffffu) >> 8 : 0;
    const int mp = (in && r > 0) ? (int)((keys[base + r - 1] >> shift) & 0xffffu) >> 8 : 0;
    const int mn = (r + 1 < n_rows) ? (int)((keys[base + r + 1] >> shift) & 0xffffu) >> 8 : 0;
    const bool hot = in && mc >= minScore;
    const bool grown = r == 0 || mc > mp;
    const bool forced = !hot || grown;
    const int val = (hot && grown) ? 1 : 0;
    int k = forced ? ((r << 1) | val) : -1;
    k = max(wave_scan_max(k), carry);
    carry = __builtin_amdgcn_readlane(k, 63);
    const bool w = hot && (((k & 1) ^ ((r - (k >> 1)) & 1)) != 0);
    const bool fin = w && !(r + 1 < n_rows && mn >= minScore && mn > mc);
    if (fin && (r < low || r > high)) best = max(best, (mc << 16) | (0xffff - r));
  }
  best = wave_max(best);
  if (best >= 0) { score2 = best >> 16; te2 = 0xffff - (best & 0xffff); }
}

struct PkRes {
  int max, max_i, max_j, n_rows;
};

template <int C>
__device__ void swp_pass(const int lane, const Duo& J, const int (&on)[2], const int (&qCols)[2], const bool pass2,
                         const int (&qEnd)[2], const int (&tEnd)[2], const uint8_t* __restrict__ pac, const long long l_pac,
                         const SwScoring& sc, const int bias, const int (&stopScore)[2], uint32_t* __restrict__ tbuf,
                         uint32_t* __restrict__ keys, int (&D)[2], PkRes (&res)[2]) {
  const auto sat16 = [](int v) { return v > 0xffff ? 0xffff : v; };
  PkConst K;
  K.eDel = pk(sat16(sc.e_del) * 0x10001); K.eIns = pk(sat16(sc.e_ins) * 0x10001);
  K.oeDel = pk(sat16(sc.o_del + sc.e_del) * 0x10001); K.oeIns = pk(sat16(sc.o_ins + sc.e_ins) * 0x10001);
  K.biasv = pk(bias * 0x10001);
  int k256 = 256 * 0x10001;
  asm volatile("" : "+v"(k256));  // opaque: a multiply-add (v_pk_mad_u16), not a shift and an add
  K.k256 = pk(k256);
  K.tail = lane > PK_LAST;
  K.same_oe = sc.o_del + sc.e_del == sc.o_ins + sc.e_ins;
  PkCols<C> S;
  int L0[2], nsteps[2], maxsteps = 0;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int pad = (PK_LAST + 1) * C - qCols[g];   // unused columns, at the left
    L0[g] = on[g] ? pad / C : 0;                    // first lane with a query column: where row t enters at step t
    D[g] = PK_LAST - L0[g];                         // row i leaves lane PK_LAST at step i + D
    nsteps[g] = on[g] ? J.tLen[g] + D[g] : 0;
    maxsteps = max(maxsteps, nsteps[g]);
  }
  maxsteps = (maxsteps + PK_G - 1) / PK_G * PK_G;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    int nrow[2];
    S.ckey[c] = 0;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int j = lane * C + c - ((PK_LAST + 1) * C - qCols[g]);
      const bool valid = on[g] && j >= 0 && j < qCols[g];
      int w = 0;
      nrow[g] = 0;
      if (valid) {
        const int fc = pass2 ? qEnd[g] - j : j;
        const int raw = J.qrev[g] ? J.qLenRaw[g] - 1 - fc : fc;
        int code = J.q[g][raw];
        if (J.qrev[g]) code = code < 4 ? 3 - code : 4;  // MemSamPe.scala:1178-1181
        if (code > 4) code = 4;
        const int sh = 8 * code;
#pragma unroll
        for (int r = 0; r < 4; ++r) w |= (((int)(int8_t)(sc.mat.row[r] >> sh) + bias) & 0xff) << (8 * r);
        nrow[g] = ((int)(int8_t)(sc.mat.row[4] >> sh) + bias) & 0xff;
        S.ckey[c] |= (255 - j) << (16 * g);
      }
      S.prof[g][c] = w;   // an unused column scores -bias against everything: it stays 0
    }
    S.pn[c] = nrow[0] | (nrow[1] << 16);
    S.Hp[c] = 0;
    S.E[c] = 0;
  }
  S.hlast_cur = S.hlast_old = S.fout = S.keyout = 0;
  const TgSrc tgA = {J.tbytes[0], pac, l_pac, J.rb[0]}, tgB = {J.tbytes[1], pac, l_pac, J.rb[1]};
  const auto selector = [&](const int ra, const int rb) {  // the v_perm_b32 selector word of row ra of job A and row rb of job B
    int sa = PK_SEL_OFF, sb = PK_SEL_OFF;
    if (on[0] && ra >= 0 && ra < J.tLen[0]) {
      const int code = tgA.at((pass2 && ra <= tEnd[0]) ? tEnd[0] - ra : ra);
      sa = code < 4 ? code : PK_SEL_N;
    }
    if (on[1] && rb >= 0 && rb < J.tLen[1]) {
      const int code = tgB.at((pass2 && rb <= tEnd[1]) ? tEnd[1] - rb : rb);
      sb = code < 4 ? 4 + code : PK_SEL_N;
    }
    return sa | (PK_SEL_OFF << 8) | (sb << 16) | (PK_SEL_OFF << 24);
  };
  const auto has_n = [](const int w) { return (w & 0xff) == PK_SEL_N || ((w >> 16) & 0xff) == PK_SEL_N; };
  // Row t of a job enters its first query lane L0 at step t: lane 0 is fed row t + L0 at step t, and the L0 rows that would
  // have had to be fed before step 0 start out in the lanes below L0, already on their way.
  S.sel = selector(L0[0] - 1 - lane, L0[1] - 1 - lane);
  bool nPrev = __builtin_amdgcn_ballot_w64(has_n(S.sel)) != 0;

  // per tail lane: the row it will hold at the end of the first group, and its running best m<<24 | (0xffff - row)<<8 | (255 - j)
  const int x = lane - PK_LAST;  // 0..PK_TAIL on the booking lanes
  const bool booker = x >= 0;
  int row[2] = {PK_TAIL - x - D[0], PK_TAIL - x - D[1]};
  int idx = PK_TAIL - x;         // step at which the key this lane holds at a group end left lane PK_LAST: its slot in keys[]
  unsigned best[2] = {0u, 0u};
  const unsigned thr[2] = {(unsigned)min(stopScore[0], 255) << 24, (unsigned)min(stopScore[1], 255) << 24};
  bool hasN = false;
  int stop[2] = {on[0] ? 0 : 1, on[1] ? 0 : 1};

  for (int t = 0; t < maxsteps; t += PK_G) {
    if ((t & (PK_TBUF - 1)) == 0) {  // stage the selector words of the next window of target rows
      int t0 = t;
      asm volatile("" : "+s"(t0));
      __builtin_amdgcn_wave_barrier();
      bool anyN = false;
      for (int k = lane; k < PK_TBUF; k += 64) {
        const int w = selector(t0 + k + L0[0], t0 + k + L0[1]);
        anyN |= has_n(w);
        tbuf[k] = (uint32_t)w;
      }
      const bool nCur = __builtin_amdgcn_ballot_w64(anyN) != 0;
      hasN = nCur || nPrev;  // rows of the previous window are still in the pipe for 63 steps
      nPrev = nCur;
      __builtin_amdgcn_wave_barrier();
    }
    const uint32_t* w8 = tbuf + (t & (PK_TBUF - 1));
    int ch[PK_G];
#pragma unroll
    for (int u = 0; u < PK_G; ++u) ch[u] = (int)w8[u];
    if (hasN) {
#pragma unroll
      for (int u = 0; u < PK_G; ++u) swp_step<C, true, false>(S, K, ch[u]);
    } else if (K.same_oe) {
#pragma unroll
      for (int u = 0; u < PK_G; ++u) swp_step<C, false, true>(S, K, ch[u]);
    } else {
#pragma unroll
      for (int u = 0; u < PK_G; ++u) swp_step<C, false, false>(S, K, ch[u]);
    }
    // the rows that have just left the pipe: SWUtil.scala:532-538 per tail lane
    if (booker) keys[idx] = (uint32_t)S.keyout;
    idx += PK_G;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const unsigned k16 = g ? (unsigned)S.keyout >> 16 : (unsigned)S.keyout & 0xffffu;
      const unsigned cand = ((k16 & 0xff00u) << 16) | ((unsigned)(0xffff - row[g]) << 8) | (k16 & 0xffu);
      // a lane's best is frozen once it holds a row that reached the stop score (0: nothing held yet)
      if (booker && row[g] >= 0 && row[g] < J.tLen[g] && (best[g] == 0u || best[g] < thr[g])) best[g] = max(best[g], cand);
      row[g] += PK_G;
      if (!stop[g] && (__builtin_amdgcn_ballot_w64(best[g] != 0u && best[g] >= thr[g]) != 0 || t + PK_G >= nsteps[g])) stop[g] = 1;
    }
    if (stop[0] & stop[1]) break;
  }
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    // the first row that reached the stop score, if any; else the best row (SWUtil.scala:532-538)
    const bool frozen = best[g] != 0u && best[g] >= thr[g];
    const unsigned long long fz = __builtin_amdgcn_ballot_w64(frozen);
    unsigned pick = fz ? (frozen ? ((best[g] >> 8) & 0xffffu) | 0x10000u : 0u) : 0u;   // (0xffff - row): the earliest row wins
    int src;
    if (fz) {
      const int top = wave_max((int)pick);
      src = __builtin_ctzll(__builtin_amdgcn_ballot_w64((int)pick == top));
    } else {
      const int hi = wave_max((int)(best[g] >> 1));   // unsigned order, without the lowest bit
      const unsigned long long c1 = __builtin_amdgcn_ballot_w64((int)(best[g] >> 1) == hi);
      // rows differ between lanes, so at most the lowest bit (of the column) cannot decide; candidates are the same row -> same lane
      src = __builtin_ctzll(c1);
    }
    const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)best[g], src);
    const bool any = on[g] && J.tLen[g] > 0;
    const int m = (int)(b >> 24);
    res[g].max = any ? m : MINUS_INF;
    res[g].max_i = any ? 0xffff - (int)((b >> 8) & 0xffffu) : -1;
    res[g].max_j = (any && m) ? 255 - (int)(b & 0xffu) : -1;
    res[g].n_rows = any ? (fz ? res[g].max_i + 1 : J.tLen[g]) : 0;
  }
}

constexpr int PK_MATE_LDS = 320;       // bytes per staged mate: (PK_LAST + 1) * C <= 285 columns for C <= 5
constexpr int PK_KEY_PAD = 128;        // steps past the last target row: pipe depth (<= 56) + group rounding + the tail lanes
constexpr int PK_KEYS_LDS_MAX = 1536;  // rows; 4 waves x (1536 + 128) words = 26 KB per workgroup

struct DuoDiag { unsigned d0 = 0u, d1 = 0u, d2 = 0u, d3 = 0u; };  // what the wave log keeps of a job pair (diagnostics builds)

// One pair of jobs (2 duo, 2 duo + 1) of a job table by one wavefront: both passes, the second best, the result records.
// Shared by the per-call launch (swp_kernel) and the resident kernel behind the submission ring (swp_resident_kernel).
template <int C>
__device__ __forceinline__ void swp_do_duo(const SwJobsDev& jobs, const SwScoring& sc, const int bias, int32_t* __restrict__ out,
                                           const int duo, const int lane, uint32_t* __restrict__ tbuf, uint32_t* __restrict__ keys,
                                           uint8_t (*mate_lds)[PK_MATE_LDS], DuoDiag& diag) {
  const int maxScore = 255 - abs(sc.b);  // SWUtil.scala:423
  const int xtra = sc.xtra;
  const int minScore = (xtra & BPSW_KSW_XSUBO) ? (xtra & 0xffff) : 0x10000;  // SWUtil.scala:434-437
  const int endScore0 = (xtra & BPSW_KSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
  Duo J;
  int job[2], on[2], qCols[2], zero2[2] = {0, 0}, stop1[2], D[2];
  // both job records with one load when the table came as records (in the host path it sits in pinned host memory, where
  // every load instruction is a PCIe request of its own): lanes 0-7 hold job 2*duo, lanes 8-15 job 2*duo + 1
  uint32_t recw = 0;
  if (jobs.packed && lane < 16 && 2 * duo + (lane >> 3) < jobs.n) recw = jobs.packed[16 * (size_t)duo + lane];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    job[g] = 2 * duo + g;
    on[g] = job[g] < jobs.n ? 1 : 0;
    const int jj = on[g] ? job[g] : 0;
    long long qoff, toff;
    if (jobs.packed) {
      const auto field = [&](const int k) { return (unsigned)__builtin_amdgcn_readlane((int)recw, 8 * g + k); };
      qoff = (long long)(((unsigned long long)field(1) << 32) | field(0));
      toff = (long long)(((unsigned long long)field(3) << 32) | field(2));
      J.qLenRaw[g] = (int)field(4); J.tLen[g] = (int)field(5); J.qrev[g] = (int)field(6);
    } else {
      J.qLenRaw[g] = uni(jobs.q_len[jj]);
      J.tLen[g] = uni(jobs.t_len[jj]);
      J.qrev[g] = uni((int)jobs.q_rev[jj]);
      qoff = jobs.q_off[jj];
      toff = jobs.t_off[jj];
    }
    // the mate goes to LDS once, 64 consecutive bytes per load: both passes build their column profiles from it (they used
    // to read it from the pool byte by strided byte, C loads per pass over the same lines)
    uint8_t* mate = mate_lds[g];
    const uint8_t* src = jobs.q_pool + qoff;
    const int ncopy = on[g] ? min(J.qLenRaw[g], PK_MATE_LDS) : 0;
    for (int k = lane; k < ncopy; k += 64) mate[k] = src[k];
    J.q[g] = mate;
    J.tbytes[g] = jobs.t_pool ? jobs.t_pool + toff : nullptr;
    J.rb[g] = toff;
    qCols[g] = J.qLenRaw[g];
    stop1[g] = min(endScore0, maxScore);  // SWUtil.scala:537
  }
  __builtin_amdgcn_wave_barrier();
  PkRes f[2];
  swp_pass<C>(lane, J, on, qCols, false, zero2, zero2, jobs.pac, jobs.l_pac, sc, bias, stop1, tbuf, keys, D, f);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the keys written by the tail lanes -> all lanes
  int score[2], te[2], qe[2], score2[2], te2[2], tb[2], qb[2], on2[2], qCols2[2], stop2[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    score[g] = f[g].max >= maxScore ? 255 : f[g].max;  // SWUtil.scala:544
    te[g] = f[g].max_i;
    qe[g] = -1; score2[g] = -1; te2[g] = -1; tb[g] = -1; qb[g] = -1;
    if (on[g] && score[g] != 255) {  // SWUtil.scala:549-567
      qe[g] = f[g].max_j;
      const int tmp = (score[g] + sc.a - 1) / sc.a;
      second_best(lane, keys, 16 * g, D[g], f[g].n_rows, minScore, te[g] - tmp, te[g] + tmp, score2[g], te2[g]);
    }
    // SWUtil.scala:586-598
    const bool want_start = (xtra & BPSW_KSW_XSTART) && !((xtra & BPSW_KSW_XSUBO) && score[g] < (xtra & 0xffff));
    on2[g] = (on[g] && want_start && qe[g] >= 0 && te[g] >= 0) ? 1 : 0;
    qCols2[g] = on2[g] ? qe[g] + 1 : 0;
    stop2[g] = min(score[g] & 0xffff, maxScore);
  }
  diag.d0 = (unsigned)((on[0] ? J.tLen[0] : 0) | ((on[1] ? J.tLen[1] : 0) << 16));
  diag.d1 = (unsigned)((on[0] ? f[0].n_rows : 0) | ((on[1] ? f[1].n_rows : 0) << 16));
  diag.d2 = (unsigned)((on2[0] ? te[0] + 1 : 0) | ((on2[1] ? te[1] + 1 : 0) << 16));
  diag.d3 = (unsigned)((on[0] ? score[0] : 0) | ((on[1] ? score[1] : 0) << 16));
  if (on2[0] | on2[1]) {
    PkRes r[2];
    __builtin_amdgcn_wave_barrier();
    swp_pass<C>(lane, J, on2, qCols2, true, qe, te, jobs.pac, jobs.l_pac, sc, bias, stop2, tbuf, keys, D, r);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (on2[g]) {
        const int rscore = r[g].max >= maxScore ? 255 : r[g].max;
        if (score[g] == rscore) { tb[g] = te[g] - r[g].max_i; qb[g] = qe[g] - r[g].max_j; }
      }
    }
  }
  {
    // both result records (jobs 2 duo and 2 duo + 1 are neighbours in `out`) with ONE store instruction, a field per lane: one
    // 56-byte write instead of fourteen four-byte ones -- the records go to the caller's pinned block over PCIe, where every store
    // instruction of a lane is a packet of its own
    int v = 0;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int f[7] = {score[g], te[g], qe[g], score2[g], te2[g], tb[g], qb[g]};
#pragma unroll
      for (int k = 0; k < 7; ++k) v = lane == 7 * g + k ? f[k] : v;
    }
    // (a system-scope store: written through to host memory now, not when the kernel ends -- the resident form publishes a unit
    // after waiting for its stores, bpsw_ring_dev.h; a plain store may sit in the L2 until a write-back)
    if (lane < 7 * (on[0] + on[1])) __hip_atomic_store(out + 7 * (size_t)job[0] + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __builtin_amdgcn_wave_barrier();
}

// KL: the packed row keys (one word per step and wave, written by the booking lanes, read once by second_best) stay in LDS --
// scratch_per_job + PK_KEY_PAD words per wave of dynamic shared memory -- instead of making a round trip through HBM, which was
// 4.5x the kernel's compulsory traffic (profiles/pmc_traffic.json, round 1).  The launcher picks KL for windows up to
// PK_KEYS_LDS_MAX rows (every 2x150 / 2x250 bp rescue window); longer windows keep the scratch rows in HBM.
template <int C, bool KL>
// (five waves per SIMD for up to three columns per lane: 91 VGPRs without a spill instead of 97; more columns keep four)
#ifndef BPSW_SWP_WAVES
#define BPSW_SWP_WAVES 5
#endif
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, C <= 3 ? BPSW_SWP_WAVES : 4) void swp_kernel(const SwJobsDev jobs, const SwScoring sc, const int bias,
                                                                     int32_t* __restrict__ out,
                                                                     uint32_t* __restrict__ scratch,
                                                                     const int scratch_per_job,
                                                                     const SwPrepass* __restrict__ pre) {
  __shared__ uint32_t tbuf_all[WAVES_PER_BLOCK][PK_TBUF + PK_G];
  __shared__ uint8_t mate_all[WAVES_PER_BLOCK][2][PK_MATE_LDS];
  extern __shared__ uint32_t key_rows[];
  BPSW_DIAG_WAVE_BEGIN();
  BPSW_DIAG_DUO_DECL();
  // Shortest launches first: a rescue launch is a quarter of an extension call's work and there are four times as many, and
  // the host's threads wait for each -- its waves go before the extension kernel's (priority 0) on the SIMDs they share, the sift
  // kernel's (priority 2: the short first launch of an extension call) before both.  Measured on the bench step: the device phase
  // of a rescue call 0.554 -> 0.512 ms, of an extension call 1.05 -> 1.09 ms, the step +3 % (1.99 -> 2.05 x 10^8 reads/s; priorities
  // 2 / 3 and a raised priority for the extension kernel's last tickets: the same).  -DBPSW_SWP_PRIO=0 / -DBPSW_SIFT_PRIO=0: off.
#ifndef BPSW_SWP_PRIO
#define BPSW_SWP_PRIO 1
#endif
  if (BPSW_SWP_PRIO) __builtin_amdgcn_s_setprio(BPSW_SWP_PRIO);
  if (pre && (pre->error != 0 || pre->max_qlen > (PK_LAST + 1) * C || ((pre->max_tlen + 63) & ~63) > scratch_per_job)) return;
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const int slot = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  uint32_t* tbuf = tbuf_all[wave];
  // packed row keys, one word per step: <= max_tlen + 70 words
  uint32_t* keys = KL ? key_rows + (size_t)wave * (size_t)(scratch_per_job + PK_KEY_PAD) : scratch + (size_t)slot * 4 * (size_t)scratch_per_job;
  const int stride = gridDim.x * WAVES_PER_BLOCK;
  const int nduo = (jobs.n + 1) >> 1;
  DuoDiag diag;
  for (int duo = slot; duo < nduo; duo += stride) swp_do_duo<C>(jobs, sc, bias, out, duo, lane, tbuf, keys, mate_all[wave], diag);
  BPSW_DIAG_DUO_SET(diag.d0, diag.d1, diag.d2, diag.d3);
  BPSW_DIAG_WAVE_END_DUO(3, out, lane);
}

// The resident form (bpsw_ring.h): the same job pairs, taken one at a time from the descriptors task threads append to the device's
// submission ring instead of from one launch's table.  Wavefront 0 of workgroup 0 is the ring's poller; every other wavefront is a
// worker.  The row keys always stay in LDS: `key_rows_cap` rows per wave, fixed for the epoch -- the host sends a call whose longest
// window has more rows through a launch of its own (swp_kernel).
template <int C>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, C <= 3 ? BPSW_SWP_WAVES : 4) void swp_resident_kernel(const RingArgs A, const int key_rows_cap) {
  __shared__ uint32_t tbuf_all[WAVES_PER_BLOCK][PK_TBUF + PK_G];
  __shared__ uint8_t mate_all[WAVES_PER_BLOCK][2][PK_MATE_LDS];
  extern __shared__ uint32_t key_rows[];
#ifndef BPSW_RING_PRIO
#define BPSW_RING_PRIO BPSW_SWP_PRIO
#endif
  if (BPSW_RING_PRIO) __builtin_amdgcn_s_setprio(BPSW_RING_PRIO);
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  if (blockIdx.x == 0 && wave == 0) {
    ring_poller(A, lane);
    return;
  }
  uint32_t* tbuf = tbuf_all[wave];
  uint32_t* keys = key_rows + (size_t)wave * (size_t)(key_rows_cap + PK_KEY_PAD);
  RingWorker W;
  DuoDiag diag;
  for (;;) {
    uint32_t unit = 0, word = 0;
    if (!ring_next_unit(A, lane, W, unit, word)) break;
    const auto f32 = [&](const int k) { return (uint32_t)__builtin_amdgcn_readlane((int)word, k); };
    const auto f64 = [&](const int k) { return ((unsigned long long)f32(k + 1) << 32) | (unsigned long long)f32(k); };
    // the payload (SwRingPayload, words 8..) as the arguments a launch would have got
    SwJobsDev jobs;
    jobs.n = (int)f32(20);
    jobs.q_len = nullptr; jobs.t_len = nullptr; jobs.q_off = nullptr; jobs.t_off = nullptr; jobs.q_rev = nullptr;
    jobs.packed = (const uint32_t*)f64(8);
    jobs.q_pool = (const uint8_t*)f64(10);
    jobs.t_pool = (const uint8_t*)f64(12);
    jobs.pac = (const uint8_t*)f64(14);
    jobs.l_pac = (long long)f64(16);
    int32_t* out = (int32_t*)f64(18);
    const int bias = (int)f32(21);
    SwScoring sc;
#pragma unroll
    for (int r = 0; r < 5; ++r) sc.mat.row[r] = f64(22 + 2 * r);
    sc.a = (int)f32(32); sc.b = (int)f32(33); sc.o_del = (int)f32(34); sc.e_del = (int)f32(35);
    sc.o_ins = (int)f32(36); sc.e_ins = (int)f32(37); sc.xtra = (int)f32(38);
    swp_do_duo<C>(jobs, sc, bias, out, (int)unit, lane, tbuf, keys, mate_all[wave], diag);
    ring_unit_done(A, lane, W, word);
  }
}

// validates the job table and finds the longest mate / window
__global__ void sw_prepass_kernel(const SwJobsDev jobs, const unsigned long long q_pool_bytes,
                                  const unsigned long long t_pool_bytes, SwPrepass* __restrict__ pre) {
  int mq = 0, mt = 0, err = 0;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < jobs.n; j += gridDim.x * blockDim.x) {
    const int ql = jobs.q_len[j], tl = jobs.t_len[j];
    const long long qo = jobs.q_off[j], to = jobs.t_off[j];
    // a window named by coordinates must lie inside one strand of the loaded reference
    const bool t_ok = jobs.t_pool ? (unsigned long long)(to + tl) <= t_pool_bytes
                                  : (to + tl <= (jobs.l_pac << 1) && (to >= jobs.l_pac || to + tl <= jobs.l_pac));
    if (ql < 1 || tl < 0 || qo < 0 || to < 0 || (unsigned long long)(qo + ql) > q_pool_bytes || !t_ok) {
      err = 1;
      continue;
    }
    mq = max(mq, ql);
    mt = max(mt, tl);
  }
  mq = wave_max(mq); mt = wave_max(mt); err = wave_max(err);  // one atomic per wavefront and word
  if ((threadIdx.x & 63) == 0) {
    if (mq) atomicMax(&pre->max_qlen, mq);
    if (mt) atomicMax(&pre->max_tlen, mt);
    if (err) atomicMax(&pre->error, err);
  }
}

// bnsGetSeq for n windows: swap / clamp / strand rules of util/BNTSeqUtil.scala:37-59, bases by TgSrc::at
__global__ void ref_fetch_kernel(const uint8_t* __restrict__ pac, const long long l_pac, const int n,
                                 const long long* __restrict__ beg, const long long* __restrict__ end,
                                 uint8_t* __restrict__ out_pool, const unsigned long long out_pool_bytes,
                                 const long long* __restrict__ out_off, long long* __restrict__ out_len,
                                 int* __restrict__ error) {
  for (int t = blockIdx.x; t < n; t += gridDim.x) {
    long long b = beg[t], e = end[t];
    if (e < b) { const long long x = b; b = e; e = x; }
    if (e > (l_pac << 1)) e = l_pac << 1;
    if (b < 0) b = 0;
    long long len = e - b;
    if (len < 0) len = 0;                        // both ends beyond 2*l_pac
    if (!(b >= l_pac || e <= l_pac)) len = 0;    // bridging the forward-reverse boundary: nothing
    const long long off = out_off[t];
    if (off < 0 || (unsigned long long)(off + len) > out_pool_bytes) {
      if (threadIdx.x == 0) { out_len[t] = len; atomicMax(error, 1); }
      continue;
    }
    if (threadIdx.x == 0) out_len[t] = len;
    const TgSrc src = {nullptr, pac, l_pac, b};
    for (long long k = threadIdx.x; k < len; k += blockDim.x) out_pool[off + k] = (uint8_t)src.at((int)k);
  }
}

template <int C>
hipError_t launch_pk(const SwJobsDev& jobs, const SwScoring& sc, int bias, int32_t* d_out, uint32_t* d_scratch, int per_job,
                     int blocks, hipStream_t s, const SwPrepass* pre, KernelEvents kev) {
  static const bool keys_hbm = getenv("BPSW_SW_KEYS_LDS") && atoi(getenv("BPSW_SW_KEYS_LDS")) == 0;  // A/B switch
  if (per_job <= PK_KEYS_LDS_MAX && !keys_hbm) {
    const size_t lds = sizeof(uint32_t) * WAVES_PER_BLOCK * (size_t)(per_job + PK_KEY_PAD);
    BPSW_LAUNCH(kev, (swp_kernel<C, true>), dim3(blocks), dim3(64 * WAVES_PER_BLOCK), lds, s, jobs, sc, bias, d_out, d_scratch, per_job, pre);
  } else {
    BPSW_LAUNCH(kev, (swp_kernel<C, false>), dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, s, jobs, sc, bias, d_out, d_scratch, per_job, pre);
  }
  return hipGetLastError();
}

template <int C>
hipError_t launch_c(const SwJobsDev& jobs, const SwScoring& sc, int32_t* d_out, uint32_t* d_scratch, int per_wave,
                    int blocks, hipStream_t s, const SwPrepass* pre, KernelEvents kev) {
  BPSW_LAUNCH(kev, sw_kernel<C>, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, s, jobs, sc, d_out, d_scratch, per_wave, pre);
  return hipGetLastError();
}

}  // namespace

void launch_sw_prepass(const SwJobsDev& jobs, size_t q_pool_bytes, size_t t_pool_bytes, SwPrepass* d_pre, hipStream_t s) {
  const int threads = 256;
  int blocks = (jobs.n + threads - 1) / threads;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(sw_prepass_kernel, dim3(blocks), dim3(threads), 0, s, jobs, (unsigned long long)q_pool_bytes,
                     (unsigned long long)t_pool_bytes, d_pre);
}

void launch_ref_fetch(const uint8_t* d_pac, long long l_pac, int n, const long long* d_beg, const long long* d_end,
                      uint8_t* d_out_pool, size_t out_pool_bytes, const long long* d_out_off, long long* d_out_len,
                      int* d_error, hipStream_t s) {
  int blocks = n < 1 ? 1 : (n > 8192 ? 8192 : n);
  hipLaunchKernelGGL(ref_fetch_kernel, dim3(blocks), dim3(256), 0, s, d_pac, l_pac, n, d_beg, d_end, d_out_pool,
                     (unsigned long long)out_pool_bytes, d_out_off, d_out_len, d_error);
}

// one uint32 list entry per target row and job: four jobs per resident wave in the quad-job kernel
size_t sw_scratch_bytes_per_wave(int max_tlen) { return 16 * (((size_t)max_tlen + 63) & ~(size_t)63); }
// The quad-job kernel serves scorings the packed kernel cannot take (sw_pack_bias).  BPSW_SW_QUAD: 0 never, 1 always (mates
// <= 160 bases), unset: for batches large enough to keep >= 6 four-job waves on every SIMD.  Measured on MI355X (tools/sw_kernel_time.py): 57 664 jobs 16.6 vs 13.6 M jobs/s, 28 832 jobs 14.6 vs 13.6,
// 7 208 jobs (the bench step) 10.0 vs 11.5 -- with four times fewer, longer waves a small batch leaves the chip half empty.
static int sw_quad_mode() {
  static const int m = getenv("BPSW_SW_QUAD") ? atoi(getenv("BPSW_SW_QUAD")) : -1;
  return m;
}
bool sw_quad_enabled() { return sw_quad_mode() != 0; }
static bool sw_quad_for(int n_jobs, int num_cu) {
  const int m = sw_quad_mode();
  if (m == 0) return false;
  if (m > 0) return true;
  return (long long)n_jobs >= 24ll * 4 * num_cu * 4 / 4 * 1;  // >= 6 waves x 4 jobs per SIMD (4 SIMDs per CU)
}
// BPSW_SW_PACK=0 turns the two-jobs-per-wave packed kernel off.  It needs what makes 16-bit halves exact (see swp_pass):
// every H that is consumed <= 255, i.e. max(mat) <= |b| + 1; biased scores in a byte; gap penalties >= 0.  Returns the bias
// or -1.
static int sw_pack_bias(const SwScoring& sc) {
  static const int off = getenv("BPSW_SW_PACK") ? atoi(getenv("BPSW_SW_PACK")) == 0 : 0;
  if (off) return -1;
  int lo = 0, hi = 0;
  for (int r = 0; r < 5; ++r)
    for (int c = 0; c < 5; ++c) {
      const int v = (int)(int8_t)(sc.mat.row[r] >> (8 * c));
      lo = v < lo ? v : lo;
      hi = v > hi ? v : hi;
    }
  const int bias = -lo;
  if (hi > abs(sc.b) + 1 || hi + bias > 254 || abs(sc.b) > 254 || sc.a < 1) return -1;
  if (sc.e_del < 0 || sc.e_ins < 0 || sc.o_del < 0 || sc.o_ins < 0) return -1;
  return bias;
}
bool sw_pack_enabled(const SwScoring& sc) { return sw_pack_bias(sc) >= 0; }
int sw_resident_waves(int num_cu) { return num_cu * 8 * WAVES_PER_BLOCK; }

hipError_t launch_sw_kernel(const SwJobsDev& jobs, const SwScoring& sc, int max_qlen, int max_tlen, int32_t* d_out,
                            uint32_t* d_scratch, int num_cu, hipStream_t s, const SwPrepass* d_pre_check, KernelEvents kev) {
  if (jobs.n <= 0) return hipSuccess;
  int blocks = (jobs.n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  // at most 8: the row scratch is sized for sw_resident_waves() = 8 workgroups per CU (a larger grid would index past it)
  static const double sw_cap = [] {
    const double v = getenv("BPSW_SW_BLOCKS_PER_CU") ? atof(getenv("BPSW_SW_BLOCKS_PER_CU")) : 8.0;
    return v > 8.0 ? 8.0 : (v > 0.0 ? v : 8.0);
  }();
  const int max_blocks = (int)(num_cu * sw_cap) > 0 ? (int)(num_cu * sw_cap) : 1;
  if (blocks > max_blocks) blocks = max_blocks;
  const int per_job = (int)(sw_scratch_bytes_per_wave(max_tlen) / 16);
  const int c = (max_qlen + 63) / 64;
  const int bias = sw_pack_bias(sc);
  if (bias >= 0 && max_qlen <= 256 && max_tlen < 65536 && jobs.n > 1) {  // two jobs per wavefront, packed 16-bit
    int pblocks = ((jobs.n + 1) / 2 + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (pblocks > max_blocks) pblocks = max_blocks;
    const int pc = (max_qlen + PK_LAST) / (PK_LAST + 1);  // columns per lane over the 64 - PK_TAIL lanes that hold the query
    if (pc <= 1) return launch_pk<1>(jobs, sc, bias, d_out, d_scratch, per_job, pblocks, s, d_pre_check, kev);
    if (pc == 2) return launch_pk<2>(jobs, sc, bias, d_out, d_scratch, per_job, pblocks, s, d_pre_check, kev);
    if (pc == 3) return launch_pk<3>(jobs, sc, bias, d_out, d_scratch, per_job, pblocks, s, d_pre_check, kev);
    if (pc == 4) return launch_pk<4>(jobs, sc, bias, d_out, d_scratch, per_job, pblocks, s, d_pre_check, kev);
    return launch_pk<5>(jobs, sc, bias, d_out, d_scratch, per_job, pblocks, s, d_pre_check, kev);
  }
  if (max_qlen <= 16 * Q4C && sw_quad_for(jobs.n, num_cu)) {  // four jobs per wavefront
    int qblocks = (jobs.n + 4 * WAVES_PER_BLOCK - 1) / (4 * WAVES_PER_BLOCK);
    if (qblocks > max_blocks) qblocks = max_blocks;
    BPSW_LAUNCH(kev, sw4_kernel, dim3(qblocks), dim3(64 * WAVES_PER_BLOCK), 0, s, jobs, sc, d_out, d_scratch, per_job, d_pre_check);
    return hipGetLastError();
  }
  const int per_wave = per_job;  // one job per wave: the first quarter of the wave's scratch
  if (c <= 1) return launch_c<1>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check, kev);
  if (c == 2) return launch_c<2>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check, kev);
  if (c == 3) return launch_c<3>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check, kev);
  if (c == 4) return launch_c<4>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check, kev);
  if (c <= 6) return launch_c<6>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check, kev);
  if (c <= 8) return launch_c<8>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check, kev);
  return hipErrorInvalidValue;
}


// ---- the resident form behind the submission ring (bpsw_ring.h) -------------------------------------------------------
// Rows of packed keys a worker wave keeps in LDS, fixed for an epoch of class c (columns per lane): the windows of 2x150 bp pairs
// have 500-700 rows, those of 2x250 bp pairs up to ~1200; a call with a longer window takes a launch of its own.
int swp_resident_key_rows(int c_class) { return c_class <= 3 ? 1024 : PK_KEYS_LDS_MAX; }

// The ring class of a batch (= the packed kernel's columns per lane, 1..5), or 0 when it has to be launched on its own: a scoring the
// packed kernel cannot take, mates above 256 bases, windows longer than the resident kernel's key rows.
int sw_ring_class(const SwScoring& sc, int max_qlen, int max_tlen, int* bias_out) {
  const int bias = sw_pack_bias(sc);
  if (bias < 0 || max_qlen > 256 || max_qlen < 1) return 0;
  const int pc = (max_qlen + PK_LAST) / (PK_LAST + 1);
  const int c_class = pc <= 3 ? 3 : 5;  // two resident kernels at most per device: mates up to 171 bases, and up to 256
  const int per_job = (int)(sw_scratch_bytes_per_wave(max_tlen) / 16);
  if (per_job > swp_resident_key_rows(c_class)) return 0;
  *bias_out = bias;
  return c_class;
}

hipError_t launch_swp_resident(int c_class, const RingArgs& A, int blocks, hipStream_t s) {
  const int cap = swp_resident_key_rows(c_class);
  const size_t lds = sizeof(uint32_t) * WAVES_PER_BLOCK * (size_t)(cap + PK_KEY_PAD);
  if (c_class == 3) hipLaunchKernelGGL((swp_resident_kernel<3>), dim3(blocks), dim3(64 * WAVES_PER_BLOCK), lds, s, A, cap);
  else if (c_class == 5) hipLaunchKernelGGL((swp_resident_kernel<5>), dim3(blocks), dim3(64 * WAVES_PER_BLOCK), lds, s, A, cap);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace bpsw
