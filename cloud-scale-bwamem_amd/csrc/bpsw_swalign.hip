// bpsw_swalign.hip -- unbanded local affine-gap SW for the pair-end mate rescue (boundary 1), gfx950.
//
// What it computes: SWUtil.SWAlign2 (SWUtil.scala:583-601) = SWAlign (SWUtil.scala:417-570) forward
// pass -> (score, tEnd, qEnd, second best outside +-ceil(score/a) rows) and, when asked for (KSW_XSTART),
// a reverse pass over the reversed prefixes that yields (tBeg, qBeg).  Bit-exact with the Scala text,
// including the true-DP second best (the SSE2 C differs there).
//
// How: one job per 64-lane wavefront, DP state entirely in registers.  The band is static, so the
// row dependency can be skewed: lane l owns the C consecutive query columns [l*C, l*C+C) and at
// step t works on target row t-l (a systolic anti-diagonal).  Everything a cell needs from its left
// neighbour -- F(i,j), the diagonal H(i-1,j-1), the running row maximum and the row's target base --
// arrives with one DPP wave_shr:1 from the previous step, so there is no in-row scan and no LDS
// traffic in the inner loop (LDS only holds a 2 KB window of target bases).  Row i leaves the pipe at
// lane (qLen-1)/C as key = H_max<<10 | (1023 - first argmax); the sequential per-row bookkeeping of
// SWUtil.scala:517-538 runs on the scalar unit.
#include <stdlib.h>

#include "bpsw_internal.h"
#include "bpsw_wave.h"

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
  if (mq) atomicMax(&pre->max_qlen, mq);
  if (mt) atomicMax(&pre->max_tlen, mt);
  if (err) atomicMax(&pre->error, err);
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
hipError_t launch_c(const SwJobsDev& jobs, const SwScoring& sc, int32_t* d_out, uint32_t* d_scratch, int per_wave,
                    int blocks, hipStream_t s, const SwPrepass* pre) {
  hipLaunchKernelGGL(sw_kernel<C>, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, s, jobs, sc, d_out, d_scratch, per_wave, pre);
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
// BPSW_SW_QUAD: 0 never, 1 always (mates <= 160 bases), unset: for batches large enough to keep >= 6 four-job waves on
// every SIMD.  Measured on MI355X (tools/sw_kernel_time.py): 57 664 jobs 16.6 vs 13.6 M jobs/s, 28 832 jobs 14.6 vs 13.6,
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
int sw_resident_waves(int num_cu) { return num_cu * 8 * WAVES_PER_BLOCK; }

hipError_t launch_sw_kernel(const SwJobsDev& jobs, const SwScoring& sc, int max_qlen, int max_tlen, int32_t* d_out,
                            uint32_t* d_scratch, int num_cu, hipStream_t s, const SwPrepass* d_pre_check) {
  if (jobs.n <= 0) return hipSuccess;
  int blocks = (jobs.n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  const int max_blocks = num_cu * 8;
  if (blocks > max_blocks) blocks = max_blocks;
  const int per_job = (int)(sw_scratch_bytes_per_wave(max_tlen) / 16);
  if (max_qlen <= 16 * Q4C && sw_quad_for(jobs.n, num_cu)) {  // four jobs per wavefront
    int qblocks = (jobs.n + 4 * WAVES_PER_BLOCK - 1) / (4 * WAVES_PER_BLOCK);
    if (qblocks > max_blocks) qblocks = max_blocks;
    hipLaunchKernelGGL(sw4_kernel, dim3(qblocks), dim3(64 * WAVES_PER_BLOCK), 0, s, jobs, sc, d_out, d_scratch, per_job, d_pre_check);
    return hipGetLastError();
  }
  const int per_wave = per_job;  // one job per wave: the first quarter of the wave's scratch
  const int c = (max_qlen + 63) / 64;
  if (c <= 1) return launch_c<1>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check);
  if (c == 2) return launch_c<2>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check);
  if (c == 3) return launch_c<3>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check);
  if (c == 4) return launch_c<4>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check);
  if (c <= 6) return launch_c<6>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check);
  if (c <= 8) return launch_c<8>(jobs, sc, d_out, d_scratch, per_wave, blocks, s, d_pre_check);
  return hipErrorInvalidValue;
}

}  // namespace bpsw
