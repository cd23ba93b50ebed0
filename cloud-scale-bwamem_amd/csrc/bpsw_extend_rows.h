// bpsw_extend_rows.h -- the SWExtend row sweep of the short extension kernels (ext_kernel<.., SHORT>) in its third form: an
// ADAPTIVE window, and a hand-written row loop.
//
// What a DP row costs is the pipe time of its instructions: ~4.5 cycles of a SIMD for a half-rate vector
// instruction (max / min / compare / select / DPP -- most of this kernel), ~2.3 for a full-rate one, ~3.7 for a scalar one, and
// the sum is what a row takes.  Two things follow.
//   * One column per lane is much cheaper than two (44 against 75 vector instructions per row), and the band of a row -- the
//     positive cells around the best one -- fits 64 columns for three rows in four even of 2x250 bp reads at 8 % / 2 % error.  So the
//     sweep keeps the (H,E) row in a 64-column window with ONE column per lane while the band fits, switches to a 128-column
//     window with two columns per lane while it does not, to a 256-column window with FOUR columns per lane when even that is too
//     narrow (round 5: rows_cpp4; 256 columns hold every band of a flank of up to 255 bases), and back (ds_bpermute moves the state
//     between the layouts; a column that enters a window is never read before the band has written it, see sw_extend_lean2).
//   * Half of a row's instructions were scalar control that the compiler builds around a loop with four exits (boolean flags in
//     SGPR pairs, s_and_b64 vcc / exec before every uniform branch, re-materialised constants).  The one-column loop -- the one that
//     serves most rows -- is written in GCN assembly here (rows1_asm): every rare case (an N row, a window that has to move, an
//     empty band) leaves the loop with the row untouched and is served by the C++ form of the same row (rows_cpp<1>), so the
//     assembly holds only the common path: 32 vector + ~45 scalar instructions per row against 44 + 65.  (Round 4 added the
//     two-column loop in assembly, rows2_asm, and "fast" forms of both for rows whose left band end cannot move: 58 / ~75 per row.)
//   * Every piece derives its per-lane constants from an opaque copy of the lane number at its own entry (rows_opaque, round 5):
//     hoisted to the top of the kernel they were a dozen VGPRs more than the short kernel's 64 hold.
// Same arithmetic, same order of evaluation as sw_extend_lean1 / lean2 / leanS (bpsw_extend_core.h), which stay for the full
// kernel and chain2aln_kernel; BPSW_EXT_ADAPTIVE=0 at build time puts the short kernels back on them.
#pragma once
#include "bpsw_extend_core.h"

namespace bpsw {
namespace {

#ifndef BPSW_EXT_ADAPTIVE
#define BPSW_EXT_ADAPTIVE 1
#endif
#ifndef BPSW_EXT_ROWS_FAST
#define BPSW_EXT_ROWS_FAST 1  // 0: every row of the assembly loops on their general form (A/B runs)
#endif
#ifndef BPSW_EXT_ROWS_ASM
#define BPSW_EXT_ROWS_ASM 1  // 0: the C++ form of the one-column loop everywhere (A/B runs, and the reference the tests compare with)
#endif

// the call's state between the pieces of the sweep: wave-uniform scalars, and per lane the (H,E) row and the profile
struct RowState {
  int i, beg, end, h1raw, mx, max_i, max_j, max_ie, gscore, max_off;
  int base;            // first column of the window
  int H0, E0, H1, E1;  // one column per lane: column base + lane in H0 / E0; two: columns base + 2 lane, base + 2 lane + 1
  int plo0, plo1;      // the lane's profile words: the scores of its column(s) against target A, C, G, T, one byte each
};
// four columns per lane (rows_cpp4: bands wider than 127 columns): columns base + 4 lane + 0..3.  Kept apart from RowState so that
// it is live only inside the wide phase of a call -- RowState rides through the assembly loops in registers
// (named scalars, not arrays: an int[4] that is copied as a whole becomes a <4 x i32> value -- four CONSECUTIVE, aligned VGPRs -- and the
// row loop's state in such tuples was part of what round 5's first build of rows_cpp4 spilled)
struct Row4 {
  int H0, H1, H2, H3, E0, E1, E2, E3, plo0, plo1, plo2, plo3;
};
#define ROWS4_EACH(M) M(0) M(1) M(2) M(3)
enum { ROWS_DONE = 0, ROWS_MORE = 1, ROWS_OTHER_MODE = 2, ROWS_SLOW = 3, ROWS_OVERFLOW = 4 };
// The lane number as a value the optimiser cannot see through.  Every piece of the sweep derives its per-lane constants (2 lane,
// lane * eIns, NEG, ...) from one of these at its own entry: a handful of instructions per piece -- per 64 rows at most -- instead
// of a dozen VGPRs that, hoisted to the top of the kernel, live (or are spilled) through every piece that does not use them.
__device__ __forceinline__ int rows_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
constexpr int ROWS_NARROW = 52;  // a band of at most this many columns goes (back) to one column per lane; wider than 63 must leave it

// The query profile of a call lives in LDS (round 4): prof[j] = the scores of query base j against target A, C, G, T (one byte
// each: the word a lane holds for its column), profn[j] = its score against a target N; entries qLen and qLen + 1 stand for every column
// past the query end (code 4; two of them, so that a lane may read the words of two neighbouring columns from any clamped index).  A window move -- in rows_cpp or inside the assembly loops -- is then two ds_bpermute and one LDS read
// instead of a nibble fetch from the wire batch per lane.
struct ProfLds {
  int* prof;        // qLen + 2 words
  int8_t* profn;    // qLen + 2 bytes
  unsigned addr;    // LDS byte address of prof (for the assembly loops' ds_read)
};
template <class QC>
__device__ __forceinline__ void rows_build_profile(const ProfLds& pl, const QC& qcode, const MatRows& mat, const int qLen, const int lane) {
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j <= qLen + 1; j += 64) {
    const int code = j < qLen ? qcode(j) : 4;
    const int sh = 8 * code;
    pl.prof[j] = (int)(((mat.row[0] >> sh) & 0xff) | (((mat.row[1] >> sh) & 0xff) << 8) | (((mat.row[2] >> sh) & 0xff) << 16) |
                       (((mat.row[3] >> sh) & 0xff) << 24));
    pl.profn[j] = (int8_t)((mat.row[4] >> sh) & 0xff);
  }
  __builtin_amdgcn_wave_barrier();
}
template <int COLS>
__device__ __forceinline__ void rows_load_profile(RowState& st, const ProfLds& pl, const int qLen, const int lane) {
  const int j0 = min(st.base + COLS * lane, qLen);
  st.plo0 = pl.prof[j0];
  if (COLS == 2) st.plo1 = pl.prof[min(j0 + 1, qLen)];
}

// The C++ form of the sweep over a window, COLS columns per lane (1: 64 columns, 2: 128), for at most max_rows rows.  It serves
// every row the assembly loop declines, every row of the two-column layout, and is the reference the assembly is tested against.
//   ROWS_DONE        the call is over (m == 0, z-drop, the tail-row bound, the last target row)
//   ROWS_MORE        max_rows rows swept
//   ROWS_OTHER_MODE  COLS == 1: the next row's band does not fit 64 columns; COLS == 2: it fits ROWS_NARROW columns again
//   ROWS_OVERFLOW    COLS == 2: the next row's band does not fit 128 columns (the task goes to the full kernel)
// In the last two cases the row has not been touched.
template <int COLS>
__device__ int rows_cpp(RowState& st, const int lane_arg, const int qLen, const int tLen, const ProfLds& pl, const uint8_t* __restrict__ ts,
                        const int oDel, const int eDel, const int oIns, const int eIns, const int w, const int zdrop,
                        const int zmode, const int h0, const int amax, int max_rows) {
  const int lane = rows_opaque(lane_arg);
  const int oeDel = oDel + eDel, oeIns = oIns + eIns;
  const int jE0 = COLS * lane * eIns - oeIns;  // j*eIns - oeIns of the lane's first column, j counted from the window's origin
  const int kC = oeIns - eIns;
  const int col0 = COLS * lane, col1 = COLS * lane + 1;
  const int WIN = 64 * COLS;
  int i = st.i, beg = st.beg, end = st.end, h1raw = st.h1raw, mx = st.mx, max_i = st.max_i, max_j = st.max_j, max_ie = st.max_ie;
  int gscore = st.gscore, max_off = st.max_off, base = st.base;
  int Hs0 = st.H0, Es0 = st.E0, Hs1 = st.H1, Es1 = st.E1;
  const int i_tail = amax > 0 ? qLen : 0x7fffffff;
  const int w1 = w + 1;
  int ret = ROWS_DONE;
  const auto save = [&]() {
    st.i = i; st.beg = beg; st.end = end; st.h1raw = h1raw; st.mx = mx; st.max_i = max_i; st.max_j = max_j; st.max_ie = max_ie;
    st.gscore = gscore; st.max_off = max_off; st.base = base; st.H0 = Hs0; st.E0 = Es0; st.H1 = Hs1; st.E1 = Es1;
  };
  for (; i < tLen; ++i) {
    if (max_rows-- <= 0) { ret = ROWS_MORE; break; }
    if (i >= i_tail) {  // nothing past this row can change the result (tail_row_bound)
      const int U = tail_row_bound(qLen, i, h0, amax, oDel, eDel);
      const int stop = (U <= mx ? 1 : 0) & (U < gscore ? 1 : 0);
      if (stop) break;
    }
    const int nbeg = smax2(beg, i - w);           // SWUtil.scala:140-142 (idempotent: an untouched row may be clamped again)
    const int nend = smin2(smin2(end, i + w1), qLen);
    if (COLS == 2 && nend - nbeg <= ROWS_NARROW && nend > nbeg) { beg = nbeg; end = nend; ret = ROWS_OTHER_MODE; break; }
    if (nend - base > WIN - 1) {  // column `end` (written this row) lies beyond the window: move the window up
      const int nb = COLS == 2 ? (nbeg & ~1) : nbeg;
      if (nend - nb > WIN - 1) { beg = nbeg; end = nend; ret = COLS == 1 ? ROWS_OTHER_MODE : ROWS_OVERFLOW; break; }
      const int from = (lane + ((nb - base) / COLS)) << 2;  // byte address of the source lane; lanes past 63 wrap and fetch
      Hs0 = __builtin_amdgcn_ds_bpermute(from, Hs0);        // columns the band has not reached yet (never read before written)
      Es0 = __builtin_amdgcn_ds_bpermute(from, Es0);
      if (COLS == 2) {
        Hs1 = __builtin_amdgcn_ds_bpermute(from, Hs1);
        Es1 = __builtin_amdgcn_ds_bpermute(from, Es1);
      }
      base = nb;
      st.base = base;
      rows_load_profile<COLS>(st, pl, qLen, lane);
    }
    beg = nbeg; end = nend;
    const int tsv = __builtin_amdgcn_readfirstlane((int)ts[i]);  // 8 * target base
    h1raw -= eDel;
    const int h1 = smax2(0, h1raw);    // SWUtil.scala:137-138
    const int rbeg = beg - base;       // the band in window coordinates
    const int span = end - beg;
    const unsigned spanA = (unsigned)smax2(span, 0);
    int scv0, scv1 = 0;
    if (__builtin_expect(tsv == 32, 0)) {  // an N row
      scv0 = (int)pl.profn[min(base + col0, qLen)];
      if (COLS == 2) scv1 = (int)pl.profn[min(base + col1, qLen)];
      asm volatile("" : "+v"(scv0), "+v"(scv1));  // keeps the branch: two selects per row otherwise
    } else {
      scv0 = __builtin_amdgcn_sbfe(st.plo0, (unsigned)tsv, 8u);
      if (COLS == 2) scv1 = __builtin_amdgcn_sbfe(st.plo1, (unsigned)tsv, 8u);
    }
    const unsigned rel0 = (unsigned)(col0 - rbeg), rel1 = (unsigned)(col1 - rbeg);
    const bool act0 = rel0 < spanA, act1 = COLS == 2 && rel1 < spanA;
    const int a0 = act0 ? max(Hs0 + scv0, Es0) : NEG_A;
    const int a1 = act1 ? max(Hs1 + scv1, Es1) : NEG_A;
    const int Pg0 = a0 + jE0, Pg1 = a1 + jE0 + eIns;
    int Pl = COLS == 2 ? max(Pg0, Pg1) : Pg0;
    int scan_a = COLS == 2 ? max((a0 << 7) | col0, (a1 << 7) | col1) : ((a0 << 7) | col0);  // the row maximum and its LAST column
    dual_scan_max(Pl, scan_a);
    const int Pprev = wave_shr1(NEG, Pl);
    const int H0 = max3i(a0, Pprev - kC - jE0, 0);
    const int H1 = COLS == 2 ? max3i(a1, max(Pprev, Pg0) - kC - jE0 - eIns, 0) : 0;
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(act0), m1 = COLS == 2 ? __builtin_amdgcn_ballot_w64(act1) : 0ull;
    const unsigned long long z0 = __builtin_amdgcn_ballot_w64(H0 < 1) & m0;
    const unsigned long long z1 = COLS == 2 ? (__builtin_amdgcn_ballot_w64(H1 < 1) & m1) : 0ull;
    const int En0 = act0 ? max3i(Es0 - eDel, H0 - oeDel, 0) : 0;
    const int En1 = act1 ? max3i(Es1 - eDel, H1 - oeDel, 0) : 0;
    if (COLS == 2) {
      const int hs0 = wave_shr1(h1, H1);  // H(i,j-1)
      Hs0 = rel0 == 0u ? h1 : hs0;       // eh[beg].h = h1, SWUtil.scala:153 (written in every lane: see sw_extend_lean2)
      Hs1 = rel1 == 0u ? h1 : H0;
      Es1 = En1;
    } else {
      const int hs = wave_shr1(h1, H0);
      Hs0 = rel0 == 0u ? h1 : hs;
    }
    Es0 = En0;
    const int mkey = smax2(0, __builtin_amdgcn_readlane(scan_a, 63));
    const int m = mkey >> 7, mjr = mkey & 127, mj = base + mjr;

    const int jlast = span > 0 ? end : beg;  // SWUtil.scala:177-182
    if (jlast == qLen) {
      int hlast = h1;
      if (span > 0) {
        const int e = end - base;
        if (COLS == 2) {
          const int he = __builtin_amdgcn_readlane(Hs0, e >> 1), ho = __builtin_amdgcn_readlane(Hs1, e >> 1);
          hlast = (e & 1) ? ho : he;
        } else {
          hlast = __builtin_amdgcn_readlane(Hs0, e);
        }
      }
      const bool better = gscore <= hlast;
      max_ie = better ? i : max_ie;
      gscore = better ? hlast : gscore;
    }
    if (m == 0) break;  // SWUtil.scala:184-185
    if (m > mx) {       // SWUtil.scala:187-193
      const int d = mj - i;
      max_off = smax2(max_off, smax2(d, -d));
      mx = m; max_i = i; max_j = mj;
    } else if (zdrop > 0) {  // SWUtil.scala:194-199 (Scala parse) / native/ksw.c:455-461 (BWA parse)
      const int stop = zdrop_stop((i - max_i) - (mj - max_j), mx - m, eDel, eIns, zdrop, zmode) ? 1 : 0;
      if (stop) break;
    }
    // band trimming, SWUtil.scala:202-214: last zero of H left of mj, first zero right of mj (window coordinates + base)
    const int nb0 = beg + (h1 == 0 ? 1 : 0);
    if ((z0 | z1) == 0ull) {
      beg = nb0;
      end = end + 1;
    } else if (COLS == 2) {
      const int ze_l = s_lead_zeros(z0 & s_below_mask((mjr + 1) >> 1));
      const int zo_l = s_lead_zeros(z1 & s_below_mask(mjr >> 1));
      const int cl = smax2(ze_l >= 0 ? 2 * (63 - ze_l) : -1, zo_l >= 0 ? 2 * (63 - zo_l) + 1 : -1);
      const int se = (mjr + 2) >> 1, so = (mjr + 1) >> 1;
      const int fe = s_first_one((z0 >> ((mjr + 1) >> 1)) >> ((mjr + 1) & 1));
      const int fo = s_first_one(z1 >> so);
      const int cr = smin2(fe >= 0 ? 2 * (se + fe) : 1 << 20, fo >= 0 ? 2 * (so + fo) + 1 : 1 << 20);
      beg = cl >= 0 ? base + cl + 2 : nb0;
      end = cr < (1 << 20) ? base + cr + 1 : end + 1;
    } else {
      const int lzc = s_lead_zeros(z0 & s_below_mask(mjr));
      const int fo = s_first_one((z0 >> mjr) >> 1);
      beg = lzc >= 0 ? base + 65 - lzc : nb0;
      end = fo >= 0 ? mj + 2 + fo : end + 1;
    }
  }
  save();
  return ret;
}

// ---- four columns per lane: a window of 256 columns (round 5) -----------------------------------------------------------------------
// A band wider than 127 columns used to leave the window sweeps altogether: the short kernel deferred the task to the full kernel or
// (2x250 bp batches) swept the WHOLE side again with the slot sweep (sw_extend_reg<3>/<4>: ~410 instructions per row, measured).
// That is not a rare path: the right-hand extension of a 250-base read starts from the score the left-hand one reached, and with
// h0 = 120-250 the first hundred rows are "live" -- h1 > 0, the band keeps its left end at column 0 and grows by a column per row --
// so a 130-231-base flank outgrows 128 columns within thirty rows.  On configs[4] a quarter of all rows went through the slot sweep,
// and they were 70 % of the extension's instructions (tools/pmc_cfg5_instr.sh, tools/pmc_rowcost.sh with H0=120).  Here the same row
// as rows_cpp, four columns per lane: one dual scan per row whatever the width, the four columns' terms folded before and after it;
// 256 columns hold every band of a flank the register path takes (qLen <= 255), so this layout never overflows.
//   ROWS_DONE        the call is over
//   ROWS_OTHER_MODE  the next row's band fits ROWS4_NARROW columns again (the row has not been touched): back to two columns per lane
constexpr int ROWS4_NARROW = 100;
__device__ __forceinline__ void rows_load_profile4(Row4& q, const int base, const ProfLds& pl, const int qLen, const int lane) {
  // (opaque: the four LDS addresses take two instructions each to form; hoisted out of the row loops they were four VGPRs that lived --
  // in scratch memory -- through the whole side)
  int l4 = 4 * lane;
  asm volatile("" : "+v"(l4));
  const int j0 = base + l4;
#define ROWS4_M(c) q.plo##c = pl.prof[min(j0 + c, qLen)];
  ROWS4_EACH(ROWS4_M)
#undef ROWS4_M
}
__device__ __forceinline__ int rows_cpp4(RowState& st, Row4& q, const int lane_arg, const int qLen, const int tLen, const ProfLds& pl, const uint8_t* __restrict__ ts,
                         const int oDel, const int eDel, const int oIns, const int eIns, const int w, const int zdrop,
                         const int zmode, const int h0, const int amax) {
  constexpr int C = 4, WIN = 256;
  const int oeDel = oDel + eDel, oeIns = oIns + eIns;
  const int lane = rows_opaque(lane_arg);
  const int lane4 = C * lane;
  const int jE0 = lane4 * eIns - oeIns;  // j*eIns - oeIns of the lane's first column, j counted from the window's origin
  const int kC = oeIns - eIns;
  int i = st.i, beg = st.beg, end = st.end, h1raw = st.h1raw, mx = st.mx, max_i = st.max_i, max_j = st.max_j, max_ie = st.max_ie;
  int gscore = st.gscore, max_off = st.max_off, base = st.base;
  int Hs0 = q.H0, Hs1 = q.H1, Hs2 = q.H2, Hs3 = q.H3, Es0 = q.E0, Es1 = q.E1, Es2 = q.E2, Es3 = q.E3;
  const int i_tail = amax > 0 ? qLen : 0x7fffffff;
  const int w1 = w + 1;
  int ret = ROWS_DONE;
  for (; i < tLen; ++i) {
    if (i >= i_tail) {  // nothing past this row can change the result (tail_row_bound)
      const int U = tail_row_bound(qLen, i, h0, amax, oDel, eDel);
      const int stop = (U <= mx ? 1 : 0) & (U < gscore ? 1 : 0);
      if (stop) break;
    }
    const int nbeg = smax2(beg, i - w);  // SWUtil.scala:140-142
    const int nend = smin2(smin2(end, i + w1), qLen);
    if (nend - nbeg <= ROWS4_NARROW && nend > nbeg) { beg = nbeg; end = nend; ret = ROWS_OTHER_MODE; break; }
    if (nend - base > WIN - 1) {  // column `end` lies beyond the window: move the window up to the band's left end (always fits: qLen <= 255)
      const int nb = nbeg & ~3;
      const int from = (lane + ((nb - base) >> 2)) << 2;
#define ROWS4_M(c) Hs##c = __builtin_amdgcn_ds_bpermute(from, Hs##c); Es##c = __builtin_amdgcn_ds_bpermute(from, Es##c);
      ROWS4_EACH(ROWS4_M)
#undef ROWS4_M
      base = nb;
      rows_load_profile4(q, base, pl, qLen, lane);
    }
    beg = nbeg; end = nend;
    const int tsv = __builtin_amdgcn_readfirstlane((int)ts[i]);  // 8 * target base
    h1raw -= eDel;
    const int h1 = smax2(0, h1raw);  // SWUtil.scala:137-138
    const int rbeg = beg - base;
    const unsigned spanA = (unsigned)smax2(end - beg, 0);
    const int span = end - beg;
    int scv0, scv1, scv2, scv3;
    if (__builtin_expect(tsv == 32, 0)) {  // an N row
      int l4 = C * lane;
      asm volatile("" : "+v"(l4));  // (as in rows_load_profile4)
#define ROWS4_M(c) scv##c = (int)pl.profn[min(base + l4 + c, qLen)];
      ROWS4_EACH(ROWS4_M)
#undef ROWS4_M
    } else {
#define ROWS4_M(c) scv##c = __builtin_amdgcn_sbfe(q.plo##c, (unsigned)tsv, 8u);
      ROWS4_EACH(ROWS4_M)
#undef ROWS4_M
    }
    // Everything per column is kept RELATIVE to the lane's first column (the j*eIns terms as c*eIns, the column number as c): the only
    // lane-dependent constants of the loop are lane4 and jE0.  (With the absolute forms the compiler kept a dozen of them -- j*eIns -
    // oeIns, kC + ..., 4 lane + c per c -- in VGPRs it had to spill around every side: 84-104 bytes of scratch per lane, 12 MB of
    // write-backs per launch, profiles/pmc_traffic.json of round 5's first build.)
    const int rel0 = lane4 - rbeg;  // the lane's first column, counted from the band's left end
#define ROWS4_M(c)                                                \
    const bool act##c = (unsigned)(rel0 + c) < spanA;             \
    const int a##c = act##c ? max(Hs##c + scv##c, Es##c) : NEG_A; \
    const int Pg##c = a##c + c * eIns;
    ROWS4_EACH(ROWS4_M)
#undef ROWS4_M
    // the lane's maximum and its LAST column
    const int k4 = max(max3i((a0 << 2) | 0, (a1 << 2) | 1, (a2 << 2) | 2), (a3 << 2) | 3);
    int Pl = max(max3i(Pg0, Pg1, Pg2), Pg3) + jE0;
    int scan_a = ((k4 & ~3) << 6) | (k4 & 3) | lane4;  // a << 8 | column: the row maximum and its LAST column in one scan (columns 0..255)
    dual_scan_max(Pl, scan_a);
    int pre = wave_shr1(NEG, Pl) - jE0;  // the F prefix of the columns left of this lane (+ j*eIns - oeIns, j from the lane's first column)
#define ROWS4_M(c)                                                                                                \
    const int H##c = max3i(a##c, pre - (kC + c * eIns), 0);                                                       \
    pre = max(pre, Pg##c);                                                                                        \
    const unsigned long long z##c = __builtin_amdgcn_ballot_w64(H##c < 1) & __builtin_amdgcn_ballot_w64(act##c); \
    Es##c = act##c ? max3i(Es##c - eDel, H##c - oeDel, 0) : 0;
    ROWS4_EACH(ROWS4_M)
#undef ROWS4_M
    {  // the row shifted by one column: Hs(col) = H(i, col - 1); eh[beg].h = h1 (SWUtil.scala:153), written in every lane
      const int hs = wave_shr1(h1, H3);
      Hs0 = rel0 == 0 ? h1 : hs;
      Hs1 = rel0 == -1 ? h1 : H0;
      Hs2 = rel0 == -2 ? h1 : H1;
      Hs3 = rel0 == -3 ? h1 : H2;
    }
    const int mkey = smax2(0, __builtin_amdgcn_readlane(scan_a, 63));
    const int m = mkey >> 8, mjr = mkey & 255, mj = base + mjr;

    const int jlast = span > 0 ? end : beg;  // SWUtil.scala:177-182
    if (jlast == qLen) {
      int hlast = h1;
      if (span > 0) {
        const int e = end - base;
        const int v0 = __builtin_amdgcn_readlane(Hs0, e >> 2), v1 = __builtin_amdgcn_readlane(Hs1, e >> 2);
        const int v2 = __builtin_amdgcn_readlane(Hs2, e >> 2), v3 = __builtin_amdgcn_readlane(Hs3, e >> 2);
        hlast = (e & 2) ? ((e & 1) ? v3 : v2) : ((e & 1) ? v1 : v0);
      }
      const bool better = gscore <= hlast;
      max_ie = better ? i : max_ie;
      gscore = better ? hlast : gscore;
    }
    if (m == 0) break;  // SWUtil.scala:184-185
    if (m > mx) {       // SWUtil.scala:187-193
      const int d = mj - i;
      max_off = smax2(max_off, smax2(d, -d));
      mx = m; max_i = i; max_j = mj;
    } else if (zdrop > 0) {  // SWUtil.scala:194-199 (Scala parse) / native/ksw.c:455-461 (BWA parse)
      const int stop = zdrop_stop((i - max_i) - (mj - max_j), mx - m, eDel, eIns, zdrop, zmode) ? 1 : 0;
      if (stop) break;
    }
    // band trimming, SWUtil.scala:202-214: the last zero column left of mj, the first zero column right of mj (columns of class c are
    // 4 lane + c: the zero masks are four interleaved bit sets)
    const int nb0 = beg + (h1 == 0 ? 1 : 0);
    if ((z0 | z1 | z2 | z3) == 0ull) {
      beg = nb0;
      end = end + 1;
    } else {
      int cl = -1, cr = 1 << 20;
      // (plain C, not the s_bfm / s_flbit / s_ff1 helpers of the two-column form: their "s" operands want values the compiler has
      // PROVEN uniform, and it does not prove it for everything this loop derives them from)
#define ROWS4_M(c)                                                                                                        \
      {                                                                                                                     \
        const int cnt = (mjr + 3 - c) >> 2; /* lanes whose column of class c lies left of mj: 4 lane + c <= mjr - 1 */      \
        const unsigned long long below = z##c & (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull));                               \
        const int top = below ? 63 - (int)__builtin_clzll(below) : -1; /* the highest such lane with a zero cell */          \
        cl = smax2(cl, top >= 0 ? C * top + c : -1);                                                                        \
        const int sc = (mjr + 4 - c) >> 2; /* first lane whose column of class c lies right of mj: 4 lane + c >= mjr + 1 */ \
        const unsigned long long above = sc >= 64 ? 0ull : (z##c >> sc);                                                    \
        const int f = above ? (int)__builtin_ctzll(above) : -1;                                                             \
        cr = smin2(cr, f >= 0 ? C * (sc + f) + c : 1 << 20);                                                                \
      }
      ROWS4_EACH(ROWS4_M)
#undef ROWS4_M
      beg = cl >= 0 ? base + cl + 2 : nb0;
      end = cr < (1 << 20) ? base + cr + 1 : end + 1;
    }
  }
  st.i = i; st.beg = beg; st.end = end; st.h1raw = h1raw; st.mx = mx; st.max_i = max_i; st.max_j = max_j; st.max_ie = max_ie;
  st.gscore = gscore; st.max_off = max_off; st.base = base;
  q.H0 = Hs0; q.H1 = Hs1; q.H2 = Hs2; q.H3 = Hs3; q.E0 = Es0; q.E1 = Es1; q.E2 = Es2; q.E3 = Es3;
  return ret;
}

// top of a row at or past the query end: the tail-row test (the rows before it run the instantiation without)
#define ROWS_TAIL_TOP "s_cmp_ge_i32 %[i], %[itail]\n\ts_cbranch_scc1 L_tail_t_%=\n\t"

// ---- the one-column loop in assembly ---------------------------------------------------------------------------------------
// Sweeps rows i .. row_end-1 of the window layout COLS == 1 (lane l holds column base + l), the common path only.  Leaves with
//   ROWS_DONE  the call is over            ROWS_MORE  i == row_end (the caller reloads the target chunk or ends the call)
//   ROWS_SLOW  row i is one the loop does not serve (column `end` beyond the window, an empty band; an N row is kept out of
//              [i, row_end) by the caller, rows_asm_end): the row has not
//              been touched (the band clamp, which is idempotent, may have been applied) -- rows_cpp<1> sweeps it.
// vTS: 8 * target base of rows (i & ~63) + lane.  Register use: see the operand list.  DPP reads need two wait states behind the
// VALU write of their operand, v_readlane / v_writelane with a scalar lane select none when a SALU instruction wrote it.
#define ROWS1_TEXT(TOP, SFX) \
      "L_row" SFX "_%=:\n\t" TOP "L_rowb" SFX "_%=:\n\t" \
      "v_readlane_b32 %[t], %[vTS], %[i]\n\t"  /* 8 * target base of row i (lane i & 63) */ \
      "s_sub_i32 %[t1], %[i], %[w]\n\t" \
      "s_max_i32 %[beg], %[beg], %[t1]\n\t"  /* beg = max(beg, i - w)            SWUtil.scala:140-142 */ \
      "s_add_i32 %[t1], %[i], %[w1]\n\t" \
      "s_min_i32 %[end], %[end], %[t1]\n\t" \
      "s_min_i32 %[end], %[end], %[qlen]\n\t"  /* end = min(end, i + w + 1, qLen) */ \
      "s_sub_i32 %[t2], %[end], %[base]\n\t" \
      "s_cmp_gt_i32 %[t2], 63\n\t" \
      "s_cbranch_scc1 L_slow" SFX "_%=\n\t"  /* column `end` beyond the window */ \
      "s_sub_i32 %[span], %[end], %[beg]\n\t" \
      "s_cmp_lt_i32 %[span], 1\n\t" \
      "s_cbranch_scc1 L_slow" SFX "_%=\n\t"  /* an empty band */ \
      "s_sub_i32 m0, %[beg], %[base]\n\t"  /* rbeg, the band's left end in window coordinates (in M0: v_writelane takes one SGPR + M0) */ \
      "v_bfe_i32 %[vS], %[vP], %[t], 8\n\t" \
      "v_subrev_u32 %[vT0], m0, %[vLane]\n\t"  /* rel = lane - rbeg */ \
      "v_cmp_gt_u32 %[act], %[span], %[vT0]\n\t"  /* act = rel < span (unsigned) */ \
      "v_add_u32 %[vA], %[vH], %[vS]\n\t" \
      "v_max_i32 %[vA], %[vA], %[vE]\n\t" \
      "v_cndmask_b32 %[vA], %[vNEG], %[vA], %[act]\n\t"  /* a = max(H(i-1,j-1) + s, E) or "no cell" */ \
      "v_sub_u32 %[vG], %[vA], %[vNegC]\n\t"  /* g = a + j*eIns */ \
      "v_lshl_or_b32 %[vK], %[vA], 7, %[vLane]\n\t"  /* a << 7 | column: the row maximum and its LAST column in one scan */ \
      "s_sub_i32 %[h1raw], %[h1raw], %[edel]\n\t"  /* (scalar work in the wait states of the scans) */ \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
      "s_max_i32 %[h1], %[h1raw], 0\n\t"  /* h1 = max(0, h0 - oDel - eDel*(i+1))   SWUtil.scala:137-138 */ \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
      "s_cmp_eq_u32 %[h1], 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
      "s_addc_u32 %[t3], %[beg], 0\n\t"  /* nb0 = beg + (h1 == 0) */ \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
      "s_nop 1\n\t" \
      "v_mov_b32_dpp %[vPp], %[vG] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"  /* exclusive prefix of g */ \
      "v_readlane_b32 %[mkey], %[vK], 63\n\t" \
      "v_add3_u32 %[vS], %[vPp], %[vNegC], %[nkc]\n\t"  /* F = Pex - (j-1)*eIns - oeIns */ \
      "v_max_i32 %[vT0], %[vA], %[vS]\n\t"  /* H (>= 0 wherever the cell is in the band: E never goes below 0) */ \
      "v_cmp_gt_i32 vcc, 1, %[vT0]\n\t"  /* H == 0 */ \
      "v_subrev_u32 %[vE], %[edel], %[vE]\n\t" \
      "v_subrev_u32 %[vS], %[oedel], %[vT0]\n\t" \
      "v_max3_i32 %[vE], %[vE], %[vS], 0\n\t"  /* E(i+1,j) = max(E - eDel, H - oeDel, 0) */ \
      "v_cndmask_b32 %[vE], 0, %[vE], %[act]\n\t"  /* eh[end].e = 0 */ \
      "v_mov_b32_dpp %[vH], %[vT0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"  /* eh[j].h = H(i,j-1) */ \
      "s_and_b64 %[z], vcc, %[act]\n\t"  /* the zero cells of the band */ \
      "v_writelane_b32 %[vH], %[h1], m0\n\t"  /* eh[beg].h = h1                        SWUtil.scala:153 */ \
      /* SWUtil.scala:177-182: j after the column loop is end (the band is not empty); h1 there is eh[end].h */ \
      "s_cmp_lg_u32 %[end], %[qlen]\n\t" \
      "s_cbranch_scc1 L_nogs" SFX "_%=\n\t" \
      "s_nop 0\n\t" \
      "v_readlane_b32 %[t1], %[vH], %[t2]\n\t"  /* lane end - base */ \
      "s_cmp_le_i32 %[gs], %[t1]\n\t" \
      "s_cselect_b32 %[maxie], %[i], %[maxie]\n\t" \
      "s_max_i32 %[gs], %[gs], %[t1]\n\t" \
      "L_nogs" SFX "_%=:\n\t" \
      "s_cmp_lt_i32 %[mkey], 128\n\t" \
      "s_cbranch_scc1 L_done" SFX "_%=\n\t"  /* m == 0                                SWUtil.scala:184-185 */ \
      "s_lshr_b32 %[m], %[mkey], 7\n\t" \
      "s_and_b32 %[mj], %[mkey], 127\n\t" \
      "s_add_i32 %[mja], %[mj], %[base]\n\t" \
      "s_cmp_gt_i32 %[m], %[mx]\n\t" \
      "s_cbranch_scc0 L_noimp" SFX "_%=\n\t" \
      "s_sub_i32 %[t1], %[mja], %[i]\n\t"  /* SWUtil.scala:187-193 */ \
      "s_abs_i32 %[t1], %[t1]\n\t" \
      "s_max_i32 %[moff], %[moff], %[t1]\n\t" \
      "s_mov_b32 %[mx], %[m]\n\t" \
      "s_mov_b32 %[maxi], %[i]\n\t" \
      "s_mov_b32 %[maxj], %[mja]\n\t" \
      "L_trim" SFX "_%=:\n\t"  /* band trimming, SWUtil.scala:202-214 */ \
      "s_cmp_eq_u64 %[z], 0\n\t" \
      "s_cbranch_scc1 L_nozero" SFX "_%=\n\t" \
      "s_bfm_b64 %[u64], %[mj], 0\n\t" \
      "s_and_b64 %[u64], %[u64], %[z]\n\t" \
      "s_flbit_i32_b64 %[t1], %[u64]\n\t"  /* last zero left of mj (leading-zero count, -1: none) */ \
      "s_lshr_b64 %[u64], %[z], %[mj]\n\t" \
      "s_lshr_b64 %[u64], %[u64], 1\n\t" \
      "s_ff1_i32_b64 %[t2], %[u64]\n\t"  /* first zero right of mj (-1: none) */ \
      "s_sub_i32 %[t4], %[b65], %[t1]\n\t" \
      "s_cmp_lt_i32 %[t1], 0\n\t" \
      "s_cselect_b32 %[beg], %[t3], %[t4]\n\t" \
      "s_add_i32 %[t4], %[mja], %[t2]\n\t" \
      "s_add_i32 %[t4], %[t4], 2\n\t" \
      "s_add_i32 %[t1], %[end], 1\n\t" \
      "s_cmp_lt_i32 %[t2], 0\n\t" \
      "s_cselect_b32 %[end], %[t1], %[t4]\n\t" \
      "s_branch L_next" SFX "_%=\n\t" \
      "L_nozero" SFX "_%=:\n\t" \
      "s_mov_b32 %[beg], %[t3]\n\t" \
      "s_add_i32 %[end], %[end], 1\n\t" \
      "L_next" SFX "_%=:\n\t" \
      "s_add_i32 %[i], %[i], 1\n\t" \
      "s_cmp_lt_i32 %[i], %[rowend]\n\t" \
      "s_cbranch_scc1 L_row" SFX "_%=\n\t" \
      "s_mov_b32 %[reason], 1\n\t"  /* ROWS_MORE */ \
      "s_branch L_out" SFX "_%=\n\t" \
      "L_noimp" SFX "_%=:\n\t"  /* SWUtil.scala:194-199 (Scala parse) / native/ksw.c:455-461 (BWA parse) */ \
      "s_cmp_lt_i32 %[zdrop], 1\n\t" \
      "s_cbranch_scc1 L_trim" SFX "_%=\n\t" \
      "s_sub_i32 %[t1], %[i], %[maxi]\n\t" \
      "s_sub_i32 %[t2], %[mja], %[maxj]\n\t" \
      "s_sub_i32 %[t1], %[t1], %[t2]\n\t"  /* k */ \
      "s_sub_i32 %[t2], %[mx], %[m]\n\t"  /* X */ \
      "s_cmp_gt_i32 %[t1], 0\n\t" \
      "s_cbranch_scc0 L_zneg" SFX "_%=\n\t" \
      "s_mul_i32 %[t4], %[t1], %[zpos]\n\t" \
      "s_add_i32 %[t4], %[t4], %[t2]\n\t" \
      "s_cmp_gt_i32 %[t4], %[zdrop]\n\t" \
      "s_cbranch_scc1 L_done" SFX "_%=\n\t" \
      "s_branch L_trim" SFX "_%=\n\t" \
      "L_zneg" SFX "_%=:\n\t" \
      "s_cmp_eq_u32 %[zneg], 0\n\t" \
      "s_cbranch_scc1 L_trim" SFX "_%=\n\t" \
      "s_mul_i32 %[t4], %[t1], %[eins]\n\t" \
      "s_add_i32 %[t4], %[t4], %[t2]\n\t" \
      "s_cmp_gt_i32 %[t4], %[zdrop]\n\t" \
      "s_cbranch_scc1 L_done" SFX "_%=\n\t" \
      "s_branch L_trim" SFX "_%=\n\t" \
      "L_tail" SFX "_%=:\n\t"  /* tail_row_bound: U = max(u0 - i*eDel, qa); over once U <= max and U < gscore */ \
      "s_mul_i32 %[t1], %[i], %[edel]\n\t" \
      "s_sub_i32 %[t1], %[u0], %[t1]\n\t" \
      "s_max_i32 %[t1], %[t1], %[qa]\n\t" \
      "s_cmp_le_i32 %[t1], %[mx]\n\t" \
      "s_cbranch_scc0 L_rowb" SFX "_%=\n\t" \
      "s_cmp_lt_i32 %[t1], %[gs]\n\t" \
      "s_cbranch_scc0 L_rowb" SFX "_%=\n\t" \
      "L_done" SFX "_%=:\n\t" \
      "s_mov_b32 %[reason], 0\n\t"  /* ROWS_DONE */ \
      "s_branch L_out" SFX "_%=\n\t" \
      "L_slow" SFX "_%=:\n\t" \
      "s_mov_b32 %[reason], 3\n\t"  /* ROWS_SLOW */ \
      "L_out" SFX "_%=:\n\t"
// ---- the one-column loop, FAST form (round 4) ---------------------------------------------------------------------------------
// What a row costs is the NUMBER of its instructions -- vector, scalar or branch alike ~2.2 cycles of the SIMD once three or more
// waves share it, s_nop free (tools/ubench/gen_rowloop_price.py, profiles/r04_rowloop_price_*.txt) -- so the common rows run a loop
// that holds only what they need, under preconditions the caller establishes per entry instead of tests per row:
//   * the window has room: end - base <= 63 on entry and row_end <= i + (base + 63 - end) + 1 (end grows by at most one per row);
//   * the band clamp cannot bind: end <= min(i + w + 1, qLen) on entry (then it never binds again: end grows by at most one per
//     row) and row_end <= beg + w + 1 (beg never decreases), so beg = max(beg, i - w) is the identity;
//   * the rows are all above the query end (no tail-row test) and all in ONE phase of h1 = max(0, h0 - oDel - eDel (i + 1)):
//     LIVE (h1 > 0 on every row: h1 = h1raw after the decrement, nb0 = beg) or DEAD (h1 == 0: no h1 at all -- H is clamped at 0
//     so that the lane left of the band hands the band's first column its eh[beg].h = 0 through the shift, nb0 = beg + 1).
// Also: the band mask is one s_bfm_b64; "improved" is one compare of the scan key against mx << 7 | 127 (mxhi) and comes before
// the m == 0 test, which only a row that did not improve needs; gscore / max_ie live in one key (H << 16 | i, signed max: a later
// row wins a tie, SWUtil.scala:178-181); the zero test of the trimming is the SCC of the s_and_b64 that builds the mask.
// 58 instructions on the common path (a row that improves, no zero cell in the band) against 87.

// ---- how a fast row learns its maximum (round 6) -----------------------------------------------------------------------------
// Rounds 3-5: a second wave-wide DPP scan beside the one F needs, on the key a << 7 | column (row maximum and its LAST column at lane
// 63): a v_lshl_or, six half-rate DPP steps and a v_readlane per row.  But a row's maximum is only ever USED when (1) it beats the
// call's maximum so far (SWUtil.scala:187-193), (2) the row did not and the z-drop has to be looked at (:194-199), or (3) the band has
// a zero cell and the trimming starts from the maximum's column (:202-214).  And in case (1) the cell that beats the old maximum is
// nearly always ALONE in its row (the diagonal's): then it IS the row's maximum and its last column -- one compare of the row against
// the old maximum (a ballot), a population count, s_ff1 and a v_readlane find it without any scan.  In case (2) a stop needs
// X + k zc1 > zlim with X = max - m and k = (i - max_i) - (mj - max_j) <= kmax = (i - max_i) - (beg - max_j): if ANY cell of the row
// holds a >= max - zlim + max(kmax, 0) zc1, the row's maximum does too and no form of the test can fire (both parses: the detailed
// tests only take more off X) -- one more compare instead of the scan; a row with no zero cell then needs nothing else.  Everything
// else -- two cells beat the old maximum in one row, a zero cell in a row that did not improve, a row that may stop -- computes the key
// scan after all, out of line (on the row's H, whose maximal cells are the maximal cells of a: F < max a), and goes through the
// unchanged code of rounds 3-5.  The common one-column row loses eight vector instructions (six of them DPP) for two and gains five
// scalar ones (58 -> 55 instructions); the two-column row ten vector for four, its new scalar work in the wait states the DPP steps need
// anyway.  Measured (one MI355X, 24 contexts of 30 k-task batches, kernels alone): 2x250 bp at 8 % / 2 % 28.35 -> 27.42 ms per round
// (-3.3 %), configs[4] 2.41 -> 2.53 x 10^7 reads/s; 2x150 bp at 1 % unchanged within the noise (2.28 ms) -- a row's time follows its
// instruction COUNT (about 2.2 cycles each whatever the kind), and that fell by a twentieth, not by the fifth the vector
// pipe's share of it did.  BPSW_ROWS_UNIQ=0 builds the old form (tools/build_variant.sh old -DBPSW_ROWS_UNIQ=0 for an A/B).
#ifndef BPSW_ROWS_UNIQ
#define BPSW_ROWS_UNIQ 1
#endif
// ... unless the band already ends at the query end: `end` never exceeds qLen (a zero cell can pull it back, it then grows by one a row up
// to qLen again) and qLen - base fits the window, so the run is bounded by its other limits only.  Without this a flank whose band
// sat near the window's top left the loop every few rows for a dispatcher pass that moved nothing (round 6).
#ifndef BPSW_ROWS_ATQ
#define BPSW_ROWS_ATQ 1
#endif
#if BPSW_ROWS_ATQ
#define ROWSF_ROOM_ATQ "s_cmp_eq_u32 %[end], %[qlen]\n\ts_cselect_b32 %[fastend], %[hardend], %[fastend]\n\t"
#else
#define ROWSF_ROOM_ATQ
#endif
#define ROWS_GSCAN_(V, NOP) \
      "v_max_i32_dpp " V ", " V ", " V " row_shr:1 row_mask:0xf bank_mask:0xf\n\t" NOP \
      "v_max_i32_dpp " V ", " V ", " V " row_shr:2 row_mask:0xf bank_mask:0xf\n\t" NOP \
      "v_max_i32_dpp " V ", " V ", " V " row_shr:4 row_mask:0xf bank_mask:0xf\n\t" NOP \
      "v_max_i32_dpp " V ", " V ", " V " row_shr:8 row_mask:0xf bank_mask:0xf\n\t" NOP \
      "v_max_i32_dpp " V ", " V ", " V " row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" NOP \
      "v_max_i32_dpp " V ", " V ", " V " row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
#if BPSW_ROWS_UNIQ
// (the maximum so far lives in its plain form, mx -- one v_readlane gives the new one; its key form mxhi = mx << 7 | 127 is made where the
// key scan's result is compared with it)
#define ROWSF_TAIL_CMP_MAX "s_cmp_le_i32 %[t1], %[mx]\n\t"
#define ROWSF_MX_FROM_KEY
#define ROWSF_MX_OUT(s_mx, s_mxhi) (s_mx)
#else
#define ROWSF_TAIL_CMP_MAX "s_lshl_b32 %[t2], %[t1], 7\n\ts_or_b32 %[t2], %[t2], 127\n\ts_cmp_le_i32 %[t2], %[mxhi]\n\t"
#define ROWSF_MX_FROM_KEY "s_lshr_b32 %[mx], %[mxhi], 7\n\t"
#define ROWSF_MX_OUT(s_mx, s_mxhi) ((s_mxhi) >> 7)
#endif
#if BPSW_ROWS_UNIQ
#define ROWS1F_KEY_INIT \
      "v_cmp_lt_i32_e64 %[u64], %[mx], %[vA]\n\t"  /* the cells that beat the call's maximum so far (mx: its plain form is the state here) */
#define ROWS1F_SCANS \
      ROWS_GSCAN_("%[vG]", "s_nop 1\n\t") \
      "s_nop 1\n\t" \
      "v_mov_b32_dpp %[vPp], %[vG] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"  /* exclusive prefix of g */
#define ROWS1F_DECIDE(SFX) \
      "s_bcnt1_i32_b64 %[h1], %[u64]\n\t"  /* SCC = some cell does */ \
      "s_cbranch_scc0 L_fni" SFX "_%=\n\t"  /* no cell beats the maximum */ \
      "s_cmp_eq_u32 %[h1], 1\n\t" \
      "s_cbranch_scc0 L_fks" SFX "_%=\n\t"  /* several do: the key scan */ \
      "s_ff1_i32_b64 %[mj], %[u64]\n\t"  /* the one cell that does: the row's maximum, and its only column */ \
      "v_readlane_b32 %[mx], %[vA], %[mj]\n\t" \
      "s_mov_b32 %[maxi], %[i]\n\t" \
      "s_add_i32 %[maxj], %[mj], %[base]\n\t" \
      "s_sub_i32 %[t1], %[maxj], %[i]\n\t" \
      "s_abs_i32 %[t1], %[t1]\n\t" \
      "s_max_i32 %[moff], %[moff], %[t1]\n\t"
#define ROWS1F_OUTOFLINE(SFX) \
      "L_fni" SFX "_%=:\n\t"  /* the row did not improve: can it stop, does the trimming need its maximum? */ \
      "s_and_b64 %[z], vcc, %[act]\n\t" \
      "s_cbranch_scc1 L_fks" SFX "_%=\n\t"  /* a zero cell: the trimming starts from the maximum's column */ \
      "s_sub_i32 %[t1], %[i], %[maxi]\n\t" \
      "s_sub_i32 %[t2], %[beg], %[maxj]\n\t" \
      "s_sub_i32 %[t1], %[t1], %[t2]\n\t"  /* kmax */ \
      "s_max_i32 %[t1], %[t1], 0\n\t" \
      "s_mul_i32 %[t1], %[t1], %[zc1]\n\t" \
      "s_add_i32 %[t1], %[t1], %[mx]\n\t" \
      "s_sub_i32 %[t1], %[t1], %[zlim]\n\t"  /* a cell this high rules every stop out */ \
      "v_cmp_le_i32_e64 %[u64], %[t1], %[vA]\n\t" \
      "s_cmp_lg_u64 %[u64], 0\n\t" \
      "s_cbranch_scc1 L_fnz" SFX "_%=\n\t"  /* (no zero cell, no stop: the next row's band) */ \
      "L_fks" SFX "_%=:\n\t"  /* the key scan after all, on H (vcc: its zero cells, as L_ftrim expects them) */ \
      "v_lshl_or_b32 %[vK], %[vT0], 7, %[vLane]\n\t" \
      "v_cmp_gt_i32 vcc, 1, %[vT0]\n\t" \
      "v_cndmask_b32 %[vK], %[vNEG], %[vK], %[act]\n\t"  /* (H of a lane outside the band is whatever F left there) */ \
      "s_nop 1\n\t" \
      ROWS_GSCAN_("%[vK]", "s_nop 1\n\t") \
      "s_nop 1\n\t" \
      "v_readlane_b32 %[mkey], %[vK], 63\n\t" \
      "s_lshl_b32 %[mxhi], %[mx], 7\n\t"  /* (the key form of the maximum: only this path and L_fnoimp behind it read it) */ \
      "s_or_b32 %[mxhi], %[mxhi], 127\n\t" \
      "s_cmp_gt_i32 %[mkey], %[mxhi]\n\t"  /* m > max                              SWUtil.scala:187-193 */ \
      "s_cbranch_scc0 L_fnoimp" SFX "_%=\n\t" \
      "s_lshr_b32 %[mx], %[mkey], 7\n\t" \
      "s_mov_b32 %[maxi], %[i]\n\t" \
      "s_and_b32 %[mj], %[mkey], 127\n\t" \
      "s_add_i32 %[maxj], %[mj], %[base]\n\t" \
      "s_sub_i32 %[t1], %[maxj], %[i]\n\t" \
      "s_abs_i32 %[t1], %[t1]\n\t" \
      "s_max_i32 %[moff], %[moff], %[t1]\n\t" \
      "s_branch L_ftrim" SFX "_%=\n\t"
#else
#define ROWS1F_KEY_INIT \
      "v_lshl_or_b32 %[vK], %[vA], 7, %[vLane]\n\t"  /* a << 7 | column */
#define ROWS1F_SCANS \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
      "s_nop 1\n\t" \
      "v_mov_b32_dpp %[vPp], %[vG] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"  /* exclusive prefix of g */ \
      "v_readlane_b32 %[mkey], %[vK], 63\n\t"
#define ROWS1F_DECIDE(SFX) \
      "s_cmp_gt_i32 %[mkey], %[mxhi]\n\t"  /* m > max                              SWUtil.scala:187-193 */ \
      "s_cbranch_scc0 L_fnoimp" SFX "_%=\n\t" \
      "s_or_b32 %[mxhi], %[mkey], 127\n\t" \
      "s_mov_b32 %[maxi], %[i]\n\t" \
      "s_and_b32 %[mj], %[mkey], 127\n\t" \
      "s_add_i32 %[maxj], %[mj], %[base]\n\t" \
      "s_sub_i32 %[t1], %[maxj], %[i]\n\t" \
      "s_abs_i32 %[t1], %[t1]\n\t" \
      "s_max_i32 %[moff], %[moff], %[t1]\n\t"
#define ROWS1F_OUTOFLINE(SFX)
#endif

// ---- the same for the two-column loop (ROWS2F_TEXT): the cells that beat the maximum so far are two ballots (even columns, odd columns);
// their counts, and the column of the one cell when there is one, are scalar work that fills the wait states between the DPP steps of
// the one scan that is left
#if BPSW_ROWS_UNIQ
#define ROWS2F_KEY_AND_SCANS(H1STEP) \
      "v_cmp_lt_i32_e64 %[u64], %[mx], %[vA0]\n\t"  /* the even / odd cells that beat the call's maximum so far */ \
      "v_cmp_lt_i32 vcc, %[mx], %[vA1]\n\t" \
      "v_max_i32 %[vG], %[vG], %[vG0]\n\t"  /* the lane's two columns folded */ \
      H1STEP \
      "s_or_b64 %[z0], %[u64], vcc\n\t"  /* (z0, z1: free until the zero masks are taken) */ \
      "s_bcnt1_i32_b64 %[h1], %[z0]\n\t"  /* lanes with such a cell */ \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
      "s_and_b64 %[z1], %[u64], vcc\n\t"  /* SCC = a lane whose two cells both do */ \
      "s_addc_u32 %[h1], %[h1], 0\n\t"  /* 1 exactly when ONE cell of the row does */ \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
      "s_ff1_i32_b64 %[mj], %[z0]\n\t" \
      "s_lshl_b32 %[mj], %[mj], 1\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
      "s_cmp_lg_u64 vcc, 0\n\t" \
      "s_addc_u32 %[mj], %[mj], 0\n\t"  /* then its column: 2 lane, + 1 when it is the odd one */ \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 1\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
      "s_nop 1\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
      "s_nop 1\n\t" \
      "v_mov_b32_dpp %[vPp], %[vG] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"  /* prefix over the columns of the lanes below */
#define ROWS2F_DECIDE(SFX) \
      "s_cmp_eq_u32 %[h1], 1\n\t" \
      "s_cbranch_scc0 L_g2nu" SFX "_%=\n\t"  /* none, or several */ \
      "v_max_i32 %[vS0], %[vA0], %[vA1]\n\t"  /* (the one cell is the row's maximum, so also its lane's; H of that cell is its a: F is below the maximum) */ \
      "s_lshr_b32 %[t1], %[mj], 1\n\t" \
      "s_mov_b32 %[maxi], %[i]\n\t" \
      "v_readlane_b32 %[mx], %[vS0], %[t1]\n\t" \
      "s_add_i32 %[maxj], %[mj], %[base]\n\t" \
      "s_sub_i32 %[t1], %[maxj], %[i]\n\t" \
      "s_abs_i32 %[t1], %[t1]\n\t" \
      "s_max_i32 %[moff], %[moff], %[t1]\n\t"
#define ROWS2F_OUTOFLINE(SFX) \
      "L_g2nu" SFX "_%=:\n\t" \
      "s_cmp_eq_u32 %[h1], 0\n\t" \
      "s_cbranch_scc0 L_g2ks" SFX "_%=\n\t"  /* several cells beat the maximum: the key scan */ \
      "s_or_b64 %[u64], %[z0], %[z1]\n\t" \
      "s_cbranch_scc1 L_g2ks" SFX "_%=\n\t"  /* not improved and a zero cell: the trimming starts from the maximum's column */ \
      "s_sub_i32 %[t1], %[i], %[maxi]\n\t" \
      "s_sub_i32 %[t2], %[beg], %[maxj]\n\t" \
      "s_sub_i32 %[t1], %[t1], %[t2]\n\t"  /* kmax */ \
      "s_max_i32 %[t1], %[t1], 0\n\t" \
      "s_mul_i32 %[t1], %[t1], %[zc1]\n\t" \
      "s_add_i32 %[t1], %[t1], %[mx]\n\t" \
      "s_sub_i32 %[t1], %[t1], %[zlim]\n\t"  /* a cell this high rules every stop out (H of a lane outside the band is below the row's maximum) */ \
      "v_cmp_le_i32_e64 %[u64], %[t1], %[vA0]\n\t" \
      "v_cmp_le_i32 vcc, %[t1], %[vA1]\n\t" \
      "s_or_b64 %[u64], %[u64], vcc\n\t" \
      "s_cbranch_scc1 L_g2nz" SFX "_%=\n\t" \
      "L_g2ks" SFX "_%=:\n\t"  /* the key scan after all, on H */ \
      "v_lshl_or_b32 %[vK], %[vA0], 7, %[vL2]\n\t" \
      "v_lshl_or_b32 %[vS0], %[vA1], 7, %[vL2p1]\n\t" \
      "v_cndmask_b32 %[vK], %[vNEG], %[vK], %[act0]\n\t" \
      "v_cndmask_b32 %[vS0], %[vNEG], %[vS0], %[act1]\n\t" \
      "v_max_i32 %[vK], %[vK], %[vS0]\n\t" \
      "s_nop 1\n\t" \
      ROWS_GSCAN_("%[vK]", "s_nop 1\n\t") \
      "s_nop 1\n\t" \
      "v_readlane_b32 %[mkey], %[vK], 63\n\t" \
      "s_lshl_b32 %[mxhi], %[mx], 7\n\t" \
      "s_or_b32 %[mxhi], %[mxhi], 127\n\t" \
      "s_cmp_gt_i32 %[mkey], %[mxhi]\n\t"  /* m > max                              SWUtil.scala:187-193 */ \
      "s_cbranch_scc0 L_g2noimp" SFX "_%=\n\t" \
      "s_lshr_b32 %[mx], %[mkey], 7\n\t" \
      "s_mov_b32 %[maxi], %[i]\n\t" \
      "s_and_b32 %[mj], %[mkey], 127\n\t" \
      "s_add_i32 %[maxj], %[mj], %[base]\n\t" \
      "s_sub_i32 %[t1], %[maxj], %[i]\n\t" \
      "s_abs_i32 %[t1], %[t1]\n\t" \
      "s_max_i32 %[moff], %[moff], %[t1]\n\t" \
      "s_branch L_g2trim" SFX "_%=\n\t"
#else
#define ROWS2F_KEY_AND_SCANS(H1STEP) \
      "v_lshl_or_b32 %[vK], %[vA0], 7, %[vL2]\n\t" \
      "v_lshl_or_b32 %[vS0], %[vA1], 7, %[vL2p1]\n\t" \
      "v_max_i32 %[vG], %[vG], %[vG0]\n\t"  /* the lane's two columns folded */ \
      "v_max_i32 %[vK], %[vK], %[vS0]\n\t"  /* a << 7 | column: row maximum and its LAST column */ \
      H1STEP \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
      "s_nop 1\n\t" \
      "v_mov_b32_dpp %[vPp], %[vG] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"  /* prefix over the columns of the lanes below */ \
      "v_readlane_b32 %[mkey], %[vK], 63\n\t"
#define ROWS2F_DECIDE(SFX) \
      "s_cmp_gt_i32 %[mkey], %[mxhi]\n\t"  /* m > max                              SWUtil.scala:187-193 */ \
      "s_cbranch_scc0 L_g2noimp" SFX "_%=\n\t" \
      "s_or_b32 %[mxhi], %[mkey], 127\n\t" \
      "s_mov_b32 %[maxi], %[i]\n\t" \
      "s_and_b32 %[mj], %[mkey], 127\n\t" \
      "s_add_i32 %[maxj], %[mj], %[base]\n\t" \
      "s_sub_i32 %[t1], %[maxj], %[i]\n\t" \
      "s_abs_i32 %[t1], %[t1]\n\t" \
      "s_max_i32 %[moff], %[moff], %[t1]\n\t"
#define ROWS2F_OUTOFLINE(SFX)
#endif
#define ROWSF_TAILTEST(SFX) \
      /* a row at or past the query end: tail_row_bound -- U = max(u0 - i eDel, qa); the call is over once U <= max and U < gscore */ \
      /* (in the loop's own forms: U << 7 | 127 <= mxhi; (U + 1) << 16 <= gskey) */ \
      "s_mul_i32 %[t1], %[i], %[edel]\n\t" \
      "s_sub_i32 %[t1], %[u0], %[t1]\n\t" \
      "s_max_i32 %[t1], %[t1], %[qa]\n\t" \
      ROWSF_TAIL_CMP_MAX \
      "s_cbranch_scc0 L_ttgo" SFX "_%=\n\t" \
      "s_add_i32 %[t1], %[t1], 1\n\t" \
      "s_lshl_b32 %[t1], %[t1], 16\n\t" \
      "s_cmp_le_i32 %[t1], %[gskey]\n\t" \
      "s_cbranch_scc1 L_fdone_%=\n\t" \
      "L_ttgo" SFX "_%=:\n\t"
// The row itself (everything between a body's label and its SWUtil.scala:177-182 step), shared by the two bodies of a fast loop
#define ROWS1F_ROW(H1STEP, HMAX, HSHIFT) \
      "v_readlane_b32 %[t], %[vTS], %[i]\n\t"  /* 8 * target base of row i */ \
      "s_nop 1\n\t" \
      "v_bfe_i32 %[vS], %[vP], %[t], 8\n\t" \
      "v_add_u32 %[vA], %[vH], %[vS]\n\t" \
      "v_max_i32 %[vA], %[vA], %[vE]\n\t" \
      "v_cndmask_b32 %[vA], %[vNEG], %[vA], %[act]\n\t"  /* a = max(H(i-1,j-1) + s, E) or "no cell" */ \
      "v_sub_u32 %[vG], %[vA], %[vNegC]\n\t"  /* g = a + j*eIns */ \
      ROWS1F_KEY_INIT \
      H1STEP \
      ROWS1F_SCANS \
      "v_add3_u32 %[vS], %[vPp], %[vNegC], %[nkc]\n\t"  /* F = Pex - (j-1)*eIns - oeIns */ \
      HMAX  /* H */ \
      "v_cmp_gt_i32 vcc, 1, %[vT0]\n\t"  /* H == 0 */ \
      "v_subrev_u32 %[vE], %[edel], %[vE]\n\t" \
      "v_subrev_u32 %[vS], %[oedel], %[vT0]\n\t" \
      "v_max3_i32 %[vE], %[vE], %[vS], 0\n\t"  /* E(i+1,j) */ \
      "v_cndmask_b32 %[vE], 0, %[vE], %[act]\n\t"  /* eh[end].e = 0 */ \
      HSHIFT  /* eh[j].h = H(i,j-1), eh[beg].h = h1 */
#define ROWS1F_GSCORE \
      "v_lshl_or_b32 %[vS], %[vH], 16, %[i]\n\t"  /* H(i,j-1) << 16 | i */ \
      "s_sub_i32 %[t2], %[end], %[base]\n\t" \
      "v_readlane_b32 %[t1], %[vS], %[t2]\n\t"  /* lane end - base: H(i, qLen-1) */ \
      "s_max_i32 %[gskey], %[gskey], %[t1]\n\t"
// The second body of a fast loop (round 6, BPSW_ROWS_ATQ): rows whose band ENDS AT THE QUERY END and is last row's (LIVE) or last row's
// moved up by one lane (DEAD) -- three rows in four of a 2x150 bp batch.  Such a row is only ever entered from a row that has just
// established end == qLen, so it needs neither the test in front of the gscore step nor the one that decides whether `end` grows:
// four instructions fewer.  Everything out of line (the key scan, the z-drop tests, the trimming on a zero cell) is shared with the
// first body and returns to IT, which makes no assumption.
#define ROWS1F_QBODY(SFX, H1STEP, HMAX, HSHIFT, NB0Q) \
      "L_fbodyq" SFX "_%=:\n\t" \
      ROWS1F_ROW(H1STEP, HMAX, HSHIFT) \
      ROWS1F_GSCORE \
      ROWS1F_DECIDE(SFX) \
      "s_and_b64 %[z], vcc, %[act]\n\t"  /* the zero cells of the band; SCC = there are some */ \
      "s_cbranch_scc1 L_fzero" SFX "_%=\n\t" \
      NB0Q
#define ROWS1F_TEXT(SFX, H1STEP, HMAX, HSHIFT, NB0_NOZERO, NB0_ZERO, PHASE_MIN, PHASE_SWITCH, TAIL_MIN, TAIL_SWITCH, TAILTOP, QBODY) \
      /* the band's set-up, only when the band is not simply last row's (LIVE: unchanged at the query end) or last row's moved up by */ \
      /* one lane (DEAD: the mask is shifted in place) */ \
      "L_frow" SFX "_%=:\n\t" \
      "s_sub_i32 %[span], %[end], %[beg]\n\t" \
      "s_cmp_lt_i32 %[span], 1\n\t" \
      "s_cbranch_scc1 L_fslow_%=\n\t"  /* an empty band */ \
      "s_sub_i32 m0, %[beg], %[base]\n\t"  /* rbeg */ \
      "s_bfm_b64 %[act], %[span], m0\n\t"  /* the lanes of the band */ \
      "L_fbody" SFX "_%=:\n\t" \
      TAILTOP \
      ROWS1F_ROW(H1STEP, HMAX, HSHIFT) \
      /* SWUtil.scala:177-182 */ \
      "s_cmp_lg_u32 %[end], %[qlen]\n\t" \
      "s_cbranch_scc1 L_fnogs" SFX "_%=\n\t" \
      ROWS1F_GSCORE \
      "L_fnogs" SFX "_%=:\n\t" \
      ROWS1F_DECIDE(SFX) \
      "L_ftrim" SFX "_%=:\n\t"  /* band trimming, SWUtil.scala:202-214 */ \
      "s_and_b64 %[z], vcc, %[act]\n\t"  /* the zero cells of the band; SCC = there are some */ \
      "s_cbranch_scc1 L_fzero" SFX "_%=\n\t" \
      "L_fnz" SFX "_%=:\n\t" \
      NB0_NOZERO  /* beg, end and the band mask of the next row; falls through when the set-up has to run */ \
      "L_fnext" SFX "_%=:\n\t" \
      "s_add_i32 %[i], %[i], 1\n\t" \
      "s_cmp_lt_i32 %[i], %[fastend]\n\t" \
      "s_cbranch_scc1 L_frow" SFX "_%=\n\t" \
      /* row i is the first one this run may not sweep as it is: out of rows (the caller looks), or out of window */ \
      "L_fbound" SFX "_%=:\n\t" \
      "s_add_i32 %[t1], %[beg], %[w1]\n\t"  /* rows up to beg + w: the left clamp cannot bind */ \
      "s_min_i32 %[hardend], %[rowend], %[t1]\n\t" \
      TAIL_MIN \
      PHASE_MIN \
      "s_cmp_lt_i32 %[i], %[hardend]\n\t" \
      "s_cbranch_scc1 L_fwin" SFX "_%=\n\t" \
      /* no row to run: the end of the target chunk, the query end, the end of the phase, or the clamp */ \
      "s_cmp_ge_i32 %[i], %[rowend]\n\t" \
      "s_cbranch_scc1 L_fchunk" SFX "_%=\n\t" \
      TAIL_SWITCH \
      PHASE_SWITCH \
      "s_branch L_ftogen_%=\n\t" \
      "L_fchunk" SFX "_%=:\n\t"  /* the next 64 target rows, when row i starts a chunk that holds no N */ \
      "s_cmp_ge_i32 %[i], %[tlen]\n\t" \
      "s_cbranch_scc1 L_fmore_%=\n\t" \
      "s_and_b32 %[t1], %[i], 63\n\t" \
      "s_cmp_lg_u32 %[t1], 0\n\t" \
      "s_cbranch_scc1 L_fmore_%=\n\t"  /* an N row ahead: the caller's */ \
      "s_add_i32 %[t1], %[tsaddr], %[i]\n\t" \
      "v_add_u32 %[vT0], %[t1], %[vLane]\n\t" \
      "ds_read_u8 %[vTS], %[vT0]\n\t" \
      "s_waitcnt lgkmcnt(0)\n\t" \
      "v_cmp_eq_u32 vcc, 32, %[vTS]\n\t" \
      "s_nop 4\n\t" \
      "s_cmp_lg_u64 vcc, 0\n\t" \
      "s_cbranch_scc1 L_fmore_%=\n\t"  /* an N row in the chunk (or stale bytes past the target's end that look like one): the caller's */ \
      "s_add_i32 %[rowend], %[i], 64\n\t" \
      "s_min_i32 %[rowend], %[rowend], %[tlen]\n\t" \
      "s_branch L_fbound" SFX "_%=\n\t" \
      "L_fwin" SFX "_%=:\n\t" \
      "s_sub_i32 %[t1], %[end], %[base]\n\t" \
      "s_cmp_lt_i32 %[t1], 64\n\t" \
      "s_cbranch_scc1 L_froom" SFX "_%=\n\t" \
      /* column `end` would fall outside the window: move the window up to the band's left end (as rows_cpp does) */ \
      "s_sub_i32 %[t2], %[beg], %[base]\n\t"  /* lanes to move down */ \
      "s_sub_i32 %[t1], %[end], %[beg]\n\t" \
      "s_cmp_gt_i32 %[t1], 63\n\t" \
      "s_cbranch_scc1 L_fslow_%=\n\t"  /* a band of 64 columns: not for this layout */ \
      "v_add_lshl_u32 %[vT0], %[vLane], %[t2], 2\n\t"  /* byte address of the source lane; lanes past 63 wrap (don't-cares) */ \
      "v_add_u32 %[vS], %[beg], %[vLane]\n\t" \
      "v_min_i32 %[vS], %[vS], %[qlen]\n\t" \
      "v_lshl_add_u32 %[vS], %[vS], 2, %[profaddr]\n\t"  /* the profile word of column beg + lane (entry qLen: past the query) */ \
      "ds_bpermute_b32 %[vH], %[vT0], %[vH]\n\t" \
      "ds_bpermute_b32 %[vE], %[vT0], %[vE]\n\t" \
      "ds_read_b32 %[vP], %[vS]\n\t" \
      "s_mov_b32 %[base], %[beg]\n\t" \
      "s_add_i32 %[b65], %[beg], 65\n\t" \
      "s_waitcnt lgkmcnt(0)\n\t" \
      "L_froom" SFX "_%=:\n\t"  /* t1 = end - base <= 63: 64 - t1 rows can run before `end` can leave the window ... */ \
      "s_sub_i32 %[t1], 64, %[t1]\n\t" \
      "s_add_i32 %[fastend], %[i], %[t1]\n\t" \
      "s_min_i32 %[fastend], %[fastend], %[hardend]\n\t" \
      ROWSF_ROOM_ATQ \
      "s_branch L_frow" SFX "_%=\n\t" \
      "L_fzero" SFX "_%=:\n\t" \
      "s_add_i32 %[mja], %[mj], %[base]\n\t" \
      "s_bfm_b64 %[u64], %[mj], 0\n\t" \
      "s_and_b64 %[u64], %[u64], %[z]\n\t" \
      "s_flbit_i32_b64 %[t1], %[u64]\n\t"  /* last zero left of mj */ \
      "s_lshr_b64 %[u64], %[z], %[mj]\n\t" \
      "s_lshr_b64 %[u64], %[u64], 1\n\t" \
      "s_ff1_i32_b64 %[t2], %[u64]\n\t"  /* first zero right of mj */ \
      "s_sub_i32 %[t4], %[b65], %[t1]\n\t" \
      NB0_ZERO \
      "s_cmp_lt_i32 %[t1], 0\n\t" \
      "s_cselect_b32 %[beg], %[t3], %[t4]\n\t" \
      "s_add_i32 %[t4], %[mja], %[t2]\n\t" \
      "s_add_i32 %[t4], %[t4], 2\n\t" \
      "s_add_i32 %[t1], %[end], 1\n\t" \
      "s_min_i32 %[t1], %[t1], %[qlen]\n\t" \
      "s_cmp_lt_i32 %[t2], 0\n\t" \
      "s_cselect_b32 %[end], %[t1], %[t4]\n\t" \
      "s_branch L_fnext" SFX "_%=\n\t" \
      ROWS1F_OUTOFLINE(SFX) \
      "L_fnoimp" SFX "_%=:\n\t" \
      "s_cmp_lt_i32 %[mkey], 128\n\t" \
      "s_cbranch_scc1 L_fdone_%=\n\t"  /* m == 0                                SWUtil.scala:184-185 */ \
      "s_and_b32 %[mj], %[mkey], 127\n\t" \
      /* SWUtil.scala:194-199 (Scala parse) / native/ksw.c:455-461 (BWA parse): k = (i - max_i) - (mj - max_j), X = max - m.  First a */ \
      /* test every stop passes -- X + k * eIns > zdrop (Scala: that IS its test once k > 0) / X > zdrop (BWA: both its tests take */ \
      /* something off X) -- which nearly every row fails; zlim = zdrop, or a value no score reaches when the z-drop is off */ \
      "s_add_i32 %[mja], %[mj], %[base]\n\t" \
      "s_sub_i32 %[t1], %[i], %[maxi]\n\t" \
      "s_sub_i32 %[t2], %[mja], %[maxj]\n\t" \
      "s_sub_i32 %[t1], %[t1], %[t2]\n\t"  /* k */ \
      "s_sub_i32 %[t2], %[mxhi], %[mkey]\n\t" \
      "s_lshr_b32 %[t2], %[t2], 7\n\t"  /* X (the low 7 bits of mxhi are all ones: no borrow from the column) */ \
      "s_mul_i32 %[t4], %[t1], %[zc1]\n\t" \
      "s_add_i32 %[t4], %[t4], %[t2]\n\t" \
      "s_cmp_gt_i32 %[t4], %[zlim]\n\t" \
      "s_cbranch_scc0 L_ftrim" SFX "_%=\n\t" \
      "s_cmp_gt_i32 %[t1], 0\n\t" \
      "s_cbranch_scc0 L_fzneg" SFX "_%=\n\t" \
      "s_mul_i32 %[t4], %[t1], %[zpos]\n\t" \
      "s_add_i32 %[t4], %[t4], %[t2]\n\t" \
      "s_cmp_gt_i32 %[t4], %[zlim]\n\t" \
      "s_cbranch_scc1 L_fdone_%=\n\t" \
      "s_branch L_ftrim" SFX "_%=\n\t" \
      "L_fzneg" SFX "_%=:\n\t" \
      "s_cmp_eq_u32 %[zneg], 0\n\t" \
      "s_cbranch_scc1 L_ftrim" SFX "_%=\n\t" \
      "s_mul_i32 %[t4], %[t1], %[eins]\n\t" \
      "s_add_i32 %[t4], %[t4], %[t2]\n\t" \
      "s_cmp_gt_i32 %[t4], %[zlim]\n\t" \
      "s_cbranch_scc1 L_fdone_%=\n\t" \
      "s_branch L_ftrim" SFX "_%=\n\t" \
      QBODY
#define ROWS1F_LIVE_H1STEP "s_sub_i32 %[h1raw], %[h1raw], %[edel]\n\t"
#define ROWS1F_LIVE_HMAX "v_max_i32 %[vT0], %[vA], %[vS]\n\t"
#define ROWS1F_LIVE_HSHIFT "v_mov_b32_dpp %[vH], %[vT0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_writelane_b32 %[vH], %[h1raw], m0\n\t"
// QT: where a row goes on whose band is this row's and ends at the query end ("L_fbodyq" with the second body, "L_fbody" without)
#define ROWS1F_LIVE_(SFX, DEADSFX, TAIL_MIN, TAIL_SWITCH, TAILTOP, QT, QBODY) \
  ROWS1F_TEXT(SFX, ROWS1F_LIVE_H1STEP, ROWS1F_LIVE_HMAX, ROWS1F_LIVE_HSHIFT, \
              /* beg stays; end at the query end: the next row's band is this row's */ \
              "s_cmp_lt_i32 %[end], %[qlen]\n\t" \
              "s_cbranch_scc1 L_fgrow" SFX "_%=\n\t" \
              "s_add_i32 %[i], %[i], 1\n\t" \
              "s_cmp_lt_i32 %[i], %[fastend]\n\t" \
              "s_cbranch_scc1 " QT SFX "_%=\n\t" \
              "s_branch L_fbound" SFX "_%=\n\t" \
              "L_fgrow" SFX "_%=:\n\t" \
              "s_add_i32 %[end], %[end], 1\n\t", \
              "s_mov_b32 %[t3], %[beg]\n\t", "s_min_i32 %[hardend], %[hardend], %[ih1z]\n\t", \
              "s_cmp_ge_i32 %[i], %[ih1z]\n\ts_cbranch_scc1 L_fbound" DEADSFX "_%=\n\t", TAIL_MIN, TAIL_SWITCH, TAILTOP, QBODY)
#define ROWS1F_LIVE_QBODY(SFX) \
  ROWS1F_QBODY(SFX, ROWS1F_LIVE_H1STEP, ROWS1F_LIVE_HMAX, ROWS1F_LIVE_HSHIFT, \
               "s_add_i32 %[i], %[i], 1\n\t" \
               "s_cmp_lt_i32 %[i], %[fastend]\n\t" \
               "s_cbranch_scc1 L_fbodyq" SFX "_%=\n\t" \
               "s_branch L_fbound" SFX "_%=\n\t")
#define ROWS1F_DEAD_H1STEP "s_nop 0\n\t"
#define ROWS1F_DEAD_HMAX "v_max3_i32 %[vT0], %[vA], %[vS], 0\n\t"
#define ROWS1F_DEAD_HSHIFT "v_mov_b32_dpp %[vH], %[vT0] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
// QSH: where a row goes on whose band ends at the query end after its mask has moved up ("L_fshq": the second body's row step; "L_fsh")
#define ROWS1F_DEAD_(SFX, TAIL_MIN, TAIL_SWITCH, TAILTOP, QSH, QBODY) \
  ROWS1F_TEXT(SFX, ROWS1F_DEAD_H1STEP, ROWS1F_DEAD_HMAX, ROWS1F_DEAD_HSHIFT, \
              /* beg + 1, and end + 1 below the query end: the band moves up by one lane (or loses its first column) -- the mask follows */ \
              "s_add_i32 %[beg], %[beg], 1\n\t" \
              "s_cmp_lt_i32 %[end], %[qlen]\n\t" \
              "s_cbranch_scc0 L_fatq" SFX "_%=\n\t" \
              "s_add_i32 %[end], %[end], 1\n\t" \
              "s_lshl_b64 %[act], %[act], 1\n\t" \
              "L_fsh" SFX "_%=:\n\t" \
              "s_add_i32 %[i], %[i], 1\n\t" \
              "s_cmp_lt_i32 %[i], %[fastend]\n\t" \
              "s_cbranch_scc1 L_fbody" SFX "_%=\n\t" \
              "s_branch L_fbound" SFX "_%=\n\t" \
              "L_fatq" SFX "_%=:\n\t" \
              "s_lshl_b64 %[u64], %[act], 1\n\t" \
              "s_and_b64 %[act], %[act], %[u64]\n\t" \
              "s_cbranch_scc1 " QSH SFX "_%=\n\t",  /* (an empty band: through the set-up, which leaves the loop) */ \
              "s_add_i32 %[t3], %[beg], 1\n\t", "", "", TAIL_MIN, TAIL_SWITCH, TAILTOP, QBODY)
#define ROWS1F_DEAD_QBODY(SFX) \
  ROWS1F_QBODY(SFX, ROWS1F_DEAD_H1STEP, ROWS1F_DEAD_HMAX, ROWS1F_DEAD_HSHIFT, \
               "s_add_i32 %[beg], %[beg], 1\n\t" \
               "s_lshl_b64 %[u64], %[act], 1\n\t" \
               "s_and_b64 %[act], %[act], %[u64]\n\t" \
               "s_cbranch_scc0 L_fnext" SFX "_%=\n\t"  /* (an empty band: through the set-up, which leaves the loop) */ \
               "L_fshq" SFX "_%=:\n\t" \
               "s_add_i32 %[i], %[i], 1\n\t" \
               "s_cmp_lt_i32 %[i], %[fastend]\n\t" \
               "s_cbranch_scc1 L_fbodyq" SFX "_%=\n\t" \
               "s_branch L_fbound" SFX "_%=\n\t")
#define ROWSF_TAILMIN "s_min_i32 %[hardend], %[hardend], %[itail]\n\t"
#ifndef BPSW_ROWS_ATQ
#define BPSW_ROWS_ATQ 1
#endif
#if BPSW_ROWS_ATQ
#define ROWS1F_ALL \
  ROWS1F_LIVE_("_l", "_d", ROWSF_TAILMIN, "s_cmp_ge_i32 %[i], %[itail]\n\ts_cbranch_scc1 L_fbound_lt_%=\n\t", "", "L_fbodyq", ROWS1F_LIVE_QBODY("_l")) \
  ROWS1F_DEAD_("_d", ROWSF_TAILMIN, "s_cmp_ge_i32 %[i], %[itail]\n\ts_cbranch_scc1 L_fbound_dt_%=\n\t", "", "L_fshq", ROWS1F_DEAD_QBODY("_d")) \
  ROWS1F_LIVE_("_lt", "_dt", "", "", ROWSF_TAILTEST("_lt"), "L_fbody", "") \
  ROWS1F_DEAD_("_dt", "", "", ROWSF_TAILTEST("_dt"), "L_fsh", "")
#else
#define ROWS1F_ALL \
  ROWS1F_LIVE_("_l", "_d", ROWSF_TAILMIN, "s_cmp_ge_i32 %[i], %[itail]\n\ts_cbranch_scc1 L_fbound_lt_%=\n\t", "", "L_fbody", "") \
  ROWS1F_DEAD_("_d", ROWSF_TAILMIN, "s_cmp_ge_i32 %[i], %[itail]\n\ts_cbranch_scc1 L_fbound_dt_%=\n\t", "", "L_fsh", "") \
  ROWS1F_LIVE_("_lt", "_dt", "", "", ROWSF_TAILTEST("_lt"), "L_fbody", "") \
  ROWS1F_DEAD_("_dt", "", "", ROWSF_TAILTEST("_dt"), "L_fsh", "")
#endif
// all the instantiations in one statement (two statements under a branch make the compiler route the scalar state through VGPRs):
// sel 0 = the general loop (ROWS1_TEXT, with or without the tail-row test), 1 = the fast loop in its LIVE phase, 2 = DEAD.
// Every read-write operand is early-clobber: an input that happens to hold the same value (end and qLen on row 0) must not share
// its register.
#define ROWS1_ASM \
  asm volatile( \
      "s_mov_b32 %[form], %[sel]\n\t" \
      "s_cmp_eq_u32 %[sel], 1\n\t" \
      "s_cbranch_scc1 L_fbound_l_%=\n\t" \
      "s_cmp_eq_u32 %[sel], 2\n\t" \
      "s_cbranch_scc1 L_fbound_d_%=\n\t" \
      "s_cmp_eq_u32 %[sel], 3\n\t" \
      "s_cbranch_scc1 L_fbound_lt_%=\n\t" \
      "s_cmp_eq_u32 %[sel], 4\n\t" \
      "s_cbranch_scc1 L_fbound_dt_%=\n\t" \
      "s_cmp_eq_u32 %[tailrows], 0\n\t" \
      "s_cbranch_scc1 L_rowb_n_%=\n\t" \
      ROWS1_TEXT(ROWS_TAIL_TOP, "_t") \
      "s_branch L_end_%=\n\t" \
      ROWS1_TEXT("", "_n") \
      "s_branch L_end_%=\n\t" \
      ROWS1F_ALL \
      "L_ftotail_%=:\n\t"  /* the rows at and past the query end: the general loop with the tail-row test, in its own form of the state */ \
      ROWSF_MX_FROM_KEY \
      "s_ashr_i32 %[gs], %[gskey], 16\n\t" \
      "s_sext_i32_i16 %[maxie], %[gskey]\n\t" \
      "s_mov_b32 %[form], 0\n\t" \
      "s_branch L_row_t_%=\n\t" \
      "L_ftogen_%=:\n\t"  /* the left clamp may bind from here on: the general loop */ \
      ROWSF_MX_FROM_KEY \
      "s_ashr_i32 %[gs], %[gskey], 16\n\t" \
      "s_sext_i32_i16 %[maxie], %[gskey]\n\t" \
      "s_mov_b32 %[form], 0\n\t" \
      "s_branch L_row_t_%=\n\t"  /* (the instantiation with the tail-row test: it may run past the query end) */ \
      "L_fmore_%=:\n\t" \
      "s_mov_b32 %[reason], 1\n\t"  /* ROWS_MORE */ \
      "s_branch L_out_%=\n\t" \
      "L_fdone_%=:\n\t" \
      "s_mov_b32 %[reason], 0\n\t"  /* ROWS_DONE */ \
      "s_branch L_out_%=\n\t" \
      "L_fslow_%=:\n\t" \
      "s_mov_b32 %[reason], 3\n\t"  /* ROWS_SLOW */ \
      "L_out_%=:\n\t" \
      "L_end_%=:\n\t" \
      : [vH] "+&v"(vH), [vE] "+&v"(vE), [vPp] "+&v"(vPp), [i] "+&s"(s_i), [beg] "+&s"(s_beg), [end] "+&s"(s_end), [h1raw] "+&s"(s_h1raw), \
        [mx] "+&s"(s_mx), [maxi] "+&s"(s_maxi), [maxj] "+&s"(s_maxj), [maxie] "+&s"(s_maxie), [gs] "+&s"(s_gs), [moff] "+&s"(s_moff), \
        [mxhi] "+&s"(s_mxhi), [gskey] "+&s"(s_gskey), [base] "+&s"(s_base), [b65] "+&s"(s_b65), [vP] "+&v"(vP), \
        [rowend] "+&s"(s_rowend), [vTS] "+&v"(vTS), [reason] "=&s"(reason), [fastend] "=&s"(s_fastend), [hardend] "=&s"(s_hardend), [form] "=&s"(s_form), [vS] "=&v"(vS), [vA] "=&v"(vA), [vG] "=&v"(vG), [vK] "=&v"(vK), [vT0] "=&v"(vT0), [t] "=&s"(t), \
        [h1] "=&s"(h1), [span] "=&s"(span), [mkey] "=&s"(mkey), [m] "=&s"(m), [mj] "=&s"(mj), [mja] "=&s"(mja), \
        [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3), [t4] "=&s"(t4), [act] "=&s"(act), [z] "=&s"(z), [u64] "=&s"(u64) \
      : [vLane] "v"(lane), [vNegC] "v"(vNegC), [vNEG] "v"(vNEG), [qlen] "s"(qLen), [tlen] "s"(s_tlen), [tsaddr] "s"(s_tsaddr), [ih1z] "s"(s_ih1z), \
        [w] "s"(w), [w1] "s"(s_w1), [edel] "s"(eDel), [oedel] "s"(oeDel), [nkc] "s"(s_nkc), \
        [zdrop] "s"(zdrop), [zpos] "s"(s_zpos), [zneg] "s"(s_zneg), [eins] "s"(eIns), [itail] "s"(i_tail), [tailrows] "s"(s_tailrows), [u0] "s"(u0), [qa] "s"(qa), \
        [sel] "s"(s_sel), [profaddr] "s"(s_profaddr), [zc1] "s"(s_zc1), [zlim] "s"(s_zlim) \
      : "vcc", "scc", "memory");  /* (M0 is written too: the compiler never keeps a value in it across statements on gfx9) */
// sel: 0 the general loop over [i, row_end), 1 / 2 the fast loop (LIVE / DEAD phase of h1) over [i, fast_end) under the preconditions
// listed at ROWS1F_TEXT, which the caller (sw_extend_adaptive) establishes
__device__ __forceinline__ int rows1_asm(RowState& st, const int lane_arg, const int qLen, const int row_end, int& vTS, const int w,
                                         const int eDel, const int oeDel, const int oeIns, const int eIns, const int zdrop, const int zmode,
                                         const int i_tail, const int u0, const int qa, const bool tail_rows, const int sel, const int tLen,
                                         const int i_h1z, const unsigned ts_addr, const unsigned prof_addr) {
  const int lane = rows_opaque(lane_arg);
  int vH = st.H0, vE = st.E0;
  int vP = st.plo0;
  const int vNegC = -(lane * eIns);     // g(k) = a(k) + k*eIns = a - vNegC;  F(j) = Pex(j) + vNegC + (eIns - oeIns)
  int vPp = NEG;                        // the exclusive prefix: lane 0 keeps "nothing to the left"
  int vNEG = rows_opaque(NEG_A);
  int s_i = st.i, s_beg = st.beg, s_end = st.end, s_h1raw = st.h1raw, s_mx = st.mx, s_maxi = st.max_i, s_maxj = st.max_j;
  int s_maxie = st.max_ie, s_gs = st.gscore, s_moff = st.max_off;
  // the fast loop's forms of max and (gscore, max_ie): mx << 7 | 127 against the scan key; gscore << 16 | max_ie under a signed max
  // ((-1, -1), "never reached the query end", is -1; a row index fits 16 bits: target flanks are a few hundred rows)
  int s_mxhi = (st.mx << 7) | 127, s_gskey = (int)(((unsigned)st.gscore << 16) | ((unsigned)st.max_ie & 0xffffu));
  int s_base = st.base, s_b65 = st.base + 65;  // (the fast loop moves the window itself)
  const int s_w1 = w + 1, s_nkc = eIns - oeIns;
  // z-drop of a row that did not improve: k = (i - max_i) - (mj - max_j), X = max - m.  k > 0: X + k * zpos > zdrop with zpos = eIns
  // (Scala parse: its B || C is C) or -eDel (BWA parse: B); k <= 0: the BWA parse alone tests X + k * eIns (zdrop_stop)
  const int s_zpos = uni(zmode == BPSW_ZDROP_SCALA ? eIns : -eDel), s_zneg = uni(zmode == BPSW_ZDROP_SCALA ? 0 : 1);
  const int s_zc1 = uni(zmode == BPSW_ZDROP_SCALA ? eIns : 0), s_zlim = uni(zdrop > 0 ? zdrop : 0x3fffffff);  // the fast loops' first z-drop test
  int reason, s_fastend, s_hardend, s_form;
  int s_rowend = row_end;  // (the fast loop fetches the next chunk of target rows itself)
  int vS, vA, vG, vK, vT0;
  int t, h1, span, mkey, m, mj, mja, t1, t2, t3, t4;
  unsigned long long act, z, u64;
  // (the general loop exists twice in the statement: the rows below the query end need no tail-row test at their top)
  const int s_tailrows = uni((int)tail_rows), s_sel = uni(sel), s_profaddr = uni((int)prof_addr), s_tlen = uni(tLen),
            s_ih1z = uni(i_h1z), s_tsaddr = uni((int)ts_addr);
  ROWS1_ASM
  st.H0 = vH; st.E0 = vE; st.plo0 = vP; st.base = s_base;
  st.i = s_i; st.beg = s_beg; st.end = s_end; st.h1raw = s_h1raw; st.max_i = s_maxi; st.max_j = s_maxj; st.max_off = s_moff;
  if (s_form) {  // the statement ended in the fast loop: its forms of max and (gscore, max_ie)
    st.mx = ROWSF_MX_OUT(s_mx, s_mxhi); st.gscore = s_gskey >> 16; st.max_ie = (int)(short)(s_gskey & 0xffff);
  } else {
    st.mx = s_mx; st.max_ie = s_maxie; st.gscore = s_gs;
  }
  return reason;
}

// ---- the two-column loop in assembly ---------------------------------------------------------------------------------------
// The same for the layout COLS == 2 (lane l holds columns base + 2l in H0 / E0 and base + 2l + 1 in H1 / E1).  It also leaves with
// ROWS_SLOW when the band has become narrow enough for one column per lane (rows_cpp<2> then reports ROWS_OTHER_MODE).
#define ROWS2_TEXT(TOP, SFX) \
      "L_row" SFX "_%=:\n\t" TOP "L_rowb" SFX "_%=:\n\t" \
      "v_readlane_b32 %[t], %[vTS], %[i]\n\t" \
      "s_sub_i32 %[t1], %[i], %[w]\n\t" \
      "s_max_i32 %[beg], %[beg], %[t1]\n\t"  /* SWUtil.scala:140-142 */ \
      "s_add_i32 %[t1], %[i], %[w1]\n\t" \
      "s_min_i32 %[end], %[end], %[t1]\n\t" \
      "s_min_i32 %[end], %[end], %[qlen]\n\t" \
      "s_sub_i32 %[t2], %[end], %[base]\n\t" \
      "s_cmp_gt_i32 %[t2], 127\n\t" \
      "s_cbranch_scc1 L_slow" SFX "_%=\n\t"  /* column `end` beyond the window */ \
      "s_sub_i32 %[span], %[end], %[beg]\n\t" \
      "s_cmp_lt_i32 %[span], %[narrow1]\n\t" \
      "s_cbranch_scc1 L_slow" SFX "_%=\n\t"  /* an empty band, or one that fits one column per lane again */ \
      "s_sub_i32 m0, %[beg], %[base]\n\t"  /* rbeg */ \
      "v_bfe_i32 %[vS0], %[vP0], %[t], 8\n\t" \
      "v_bfe_i32 %[vS1], %[vP1], %[t], 8\n\t" \
      "v_subrev_u32 %[vT0], m0, %[vL2]\n\t"  /* rel0 = 2 lane - rbeg */ \
      "v_add_u32 %[vT1], 1, %[vT0]\n\t"  /* rel1 */ \
      "v_cmp_gt_u32 %[act0], %[span], %[vT0]\n\t" \
      "v_cmp_gt_u32 %[act1], %[span], %[vT1]\n\t" \
      "v_add_u32 %[vA0], %[vH0], %[vS0]\n\t" \
      "v_add_u32 %[vA1], %[vH1], %[vS1]\n\t" \
      "v_max_i32 %[vA0], %[vA0], %[vE0]\n\t" \
      "v_max_i32 %[vA1], %[vA1], %[vE1]\n\t" \
      "v_cndmask_b32 %[vA0], %[vNEG], %[vA0], %[act0]\n\t" \
      "v_cndmask_b32 %[vA1], %[vNEG], %[vA1], %[act1]\n\t" \
      "v_sub_u32 %[vG0], %[vA0], %[vNegC]\n\t"  /* g of the even column */ \
      "v_sub_u32 %[vG], %[vA1], %[vNegC]\n\t" \
      "v_add_u32 %[vG], %[eins], %[vG]\n\t"  /* g of the odd column */ \
      "v_lshl_or_b32 %[vK], %[vA0], 7, %[vL2]\n\t" \
      "v_lshl_or_b32 %[vS0], %[vA1], 7, %[vL2]\n\t" \
      "v_or_b32 %[vS0], 1, %[vS0]\n\t" \
      "v_max_i32 %[vG], %[vG], %[vG0]\n\t"  /* the lane's two columns folded */ \
      "v_max_i32 %[vK], %[vK], %[vS0]\n\t"  /* a << 7 | column: row maximum and its LAST column */ \
      "s_sub_i32 %[h1raw], %[h1raw], %[edel]\n\t" \
      "s_max_i32 %[h1], %[h1raw], 0\n\t"  /* SWUtil.scala:137-138 */ \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
      "s_cmp_eq_u32 %[h1], 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
      "s_addc_u32 %[t3], %[beg], 0\n\t"  /* nb0 = beg + (h1 == 0) */ \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
      "v_mov_b32 %[vh1], %[h1]\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
      "s_nop 0\n\t" \
      "v_max_i32_dpp %[vG], %[vG], %[vG] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
      "v_max_i32_dpp %[vK], %[vK], %[vK] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
      "s_nop 1\n\t" \
      "v_mov_b32_dpp %[vPp], %[vG] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"  /* prefix over the columns of the lanes below */ \
      "v_readlane_b32 %[mkey], %[vK], 63\n\t" \
      "v_add3_u32 %[vS0], %[vPp], %[vNegC], %[nkc]\n\t"  /* F of the even column */ \
      "v_max_i32 %[vS1], %[vPp], %[vG0]\n\t" \
      "v_add3_u32 %[vS1], %[vS1], %[vNegC], %[nkc1]\n\t"  /* F of the odd column */ \
      "v_max_i32 %[vA0], %[vA0], %[vS0]\n\t"  /* H even (>= 0 wherever the cell is in the band) */ \
      "v_max_i32 %[vA1], %[vA1], %[vS1]\n\t"  /* H odd */ \
      "v_cmp_gt_i32 vcc, 1, %[vA0]\n\t" \
      "s_and_b64 %[z0], vcc, %[act0]\n\t"  /* zero cells of the band, even columns */ \
      "v_cmp_gt_i32 vcc, 1, %[vA1]\n\t" \
      "s_and_b64 %[z1], vcc, %[act1]\n\t"  /* ... odd columns */ \
      "v_subrev_u32 %[vE0], %[edel], %[vE0]\n\t" \
      "v_subrev_u32 %[vS0], %[oedel], %[vA0]\n\t" \
      "v_max3_i32 %[vE0], %[vE0], %[vS0], 0\n\t" \
      "v_cndmask_b32 %[vE0], 0, %[vE0], %[act0]\n\t"  /* E(i+1,j); eh[end].e = 0 */ \
      "v_subrev_u32 %[vE1], %[edel], %[vE1]\n\t" \
      "v_subrev_u32 %[vS1], %[oedel], %[vA1]\n\t" \
      "v_max3_i32 %[vE1], %[vE1], %[vS1], 0\n\t" \
      "v_cndmask_b32 %[vE1], 0, %[vE1], %[act1]\n\t" \
      "v_mov_b32_dpp %[vH0], %[vA1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"  /* eh[j].h = H(i,j-1): even column <- odd of the lane below */ \
      "v_cmp_eq_u32 vcc, 0, %[vT0]\n\t" \
      "v_cndmask_b32 %[vH0], %[vH0], %[vh1], vcc\n\t"  /* eh[beg].h = h1   SWUtil.scala:153 */ \
      "v_cmp_eq_u32 vcc, 0, %[vT1]\n\t" \
      "v_cndmask_b32 %[vH1], %[vA0], %[vh1], vcc\n\t"  /* odd column <- even of the same lane */ \
      /* SWUtil.scala:177-182 */ \
      "s_cmp_lg_u32 %[end], %[qlen]\n\t" \
      "s_cbranch_scc1 L_nogs" SFX "_%=\n\t" \
      "s_lshr_b32 %[t1], %[t2], 1\n\t"  /* t2 = end - base: lane, and which of the lane's columns */ \
      "s_and_b32 %[t2], %[t2], 1\n\t" \
      "v_readlane_b32 %[t4], %[vH0], %[t1]\n\t" \
      "v_readlane_b32 %[t1], %[vH1], %[t1]\n\t" \
      "s_cmp_eq_u32 %[t2], 0\n\t" \
      "s_cselect_b32 %[t1], %[t4], %[t1]\n\t" \
      "s_cmp_le_i32 %[gs], %[t1]\n\t" \
      "s_cselect_b32 %[maxie], %[i], %[maxie]\n\t" \
      "s_max_i32 %[gs], %[gs], %[t1]\n\t" \
      "L_nogs" SFX "_%=:\n\t" \
      "s_cmp_lt_i32 %[mkey], 128\n\t" \
      "s_cbranch_scc1 L_done" SFX "_%=\n\t"  /* m == 0 */ \
      "s_lshr_b32 %[m], %[mkey], 7\n\t" \
      "s_and_b32 %[mj], %[mkey], 127\n\t" \
      "s_add_i32 %[mja], %[mj], %[base]\n\t" \
      "s_cmp_gt_i32 %[m], %[mx]\n\t" \
      "s_cbranch_scc0 L_noimp" SFX "_%=\n\t" \
      "s_sub_i32 %[t1], %[mja], %[i]\n\t" \
      "s_abs_i32 %[t1], %[t1]\n\t" \
      "s_max_i32 %[moff], %[moff], %[t1]\n\t" \
      "s_mov_b32 %[mx], %[m]\n\t" \
      "s_mov_b32 %[maxi], %[i]\n\t" \
      "s_mov_b32 %[maxj], %[mja]\n\t" \
      "L_trim" SFX "_%=:\n\t"  /* band trimming, SWUtil.scala:202-214, on the even / odd zero masks */ \
      "s_or_b64 %[u64], %[z0], %[z1]\n\t" \
      "s_cmp_eq_u64 %[u64], 0\n\t" \
      "s_cbranch_scc1 L_nozero" SFX "_%=\n\t" \
      /* last zero left of mj: even columns 2l < mj <=> l < (mj+1)>>1; odd columns 2l+1 < mj <=> l < mj>>1 */ \
      "s_add_i32 %[t5], %[mj], 1\n\t" \
      "s_lshr_b32 %[t6], %[t5], 1\n\t"  /* (mj+1)>>1 */ \
      "s_bfm_b64 %[u64], %[t6], 0\n\t" \
      "s_and_b64 %[u64], %[u64], %[z0]\n\t" \
      "s_flbit_i32_b64 %[t1], %[u64]\n\t"  /* -1: none, else 63 - lane */ \
      "s_lshr_b32 %[t4], %[mj], 1\n\t"  /* mj>>1 */ \
      "s_bfm_b64 %[u64], %[t4], 0\n\t" \
      "s_and_b64 %[u64], %[u64], %[z1]\n\t" \
      "s_flbit_i32_b64 %[t2], %[u64]\n\t" \
      /* column of each candidate + 2, or a value below every real one: even 2(63 - t1) + 2 = 128 - 2 t1; odd 2(63 - t2) + 3 = 129 - 2 t2 */ \
      "s_lshl_b32 %[m], %[t1], 1\n\t" \
      "s_sub_i32 %[m], 128, %[m]\n\t" \
      "s_cmp_lt_i32 %[t1], 0\n\t" \
      "s_cselect_b32 %[m], -1, %[m]\n\t" \
      "s_lshl_b32 %[t1], %[t2], 1\n\t" \
      "s_sub_i32 %[t1], 129, %[t1]\n\t" \
      "s_cmp_lt_i32 %[t2], 0\n\t" \
      "s_cselect_b32 %[t1], -1, %[t1]\n\t" \
      "s_max_i32 %[m], %[m], %[t1]\n\t"  /* cl + 2, or -1 */ \
      "s_add_i32 %[t1], %[m], %[base]\n\t" \
      "s_cmp_lt_i32 %[m], 0\n\t" \
      "s_cselect_b32 %[beg], %[t3], %[t1]\n\t"  /* beg = base + cl + 2, or nb0 */ \
      /* first zero right of mj: even columns 2l > mj <=> l >= (mj+2)>>1; odd columns 2l+1 > mj <=> l >= (mj+1)>>1 = t6 */ \
      "s_and_b32 %[t5], %[t5], 1\n\t"  /* (mj+1) & 1 */ \
      "s_add_i32 %[t4], %[t6], %[t5]\n\t"  /* se = (mj+2)>>1 (at most 64: shifted in two steps) */ \
      "s_lshr_b64 %[u64], %[z0], %[t6]\n\t" \
      "s_lshr_b64 %[u64], %[u64], %[t5]\n\t" \
      "s_ff1_i32_b64 %[t1], %[u64]\n\t" \
      "s_lshr_b64 %[u64], %[z1], %[t6]\n\t" \
      "s_ff1_i32_b64 %[t2], %[u64]\n\t" \
      "s_add_i32 %[m], %[t4], %[t1]\n\t" \
      "s_lshl_b32 %[m], %[m], 1\n\t"  /* 2 (se + fe) */ \
      "s_cmp_lt_i32 %[t1], 0\n\t" \
      "s_cselect_b32 %[m], 0x100000, %[m]\n\t" \
      "s_add_i32 %[t1], %[t6], %[t2]\n\t" \
      "s_lshl_b32 %[t1], %[t1], 1\n\t" \
      "s_add_i32 %[t1], %[t1], 1\n\t"  /* 2 (so + fo) + 1 */ \
      "s_cmp_lt_i32 %[t2], 0\n\t" \
      "s_cselect_b32 %[t1], 0x100000, %[t1]\n\t" \
      "s_min_i32 %[m], %[m], %[t1]\n\t"  /* cr, or 0x100000 */ \
      "s_add_i32 %[t1], %[m], %[base]\n\t" \
      "s_add_i32 %[t1], %[t1], 1\n\t" \
      "s_add_i32 %[t2], %[end], 1\n\t" \
      "s_cmp_lt_i32 %[m], 0x100000\n\t" \
      "s_cselect_b32 %[end], %[t1], %[t2]\n\t"  /* end = base + cr + 1, or end + 1 */ \
      "s_branch L_next" SFX "_%=\n\t" \
      "L_nozero" SFX "_%=:\n\t" \
      "s_mov_b32 %[beg], %[t3]\n\t" \
      "s_add_i32 %[end], %[end], 1\n\t" \
      "L_next" SFX "_%=:\n\t" \
      "s_add_i32 %[i], %[i], 1\n\t" \
      "s_cmp_lt_i32 %[i], %[rowend]\n\t" \
      "s_cbranch_scc1 L_row" SFX "_%=\n\t" \
      "s_mov_b32 %[reason], 1\n\t" \
      "s_branch L_out" SFX "_%=\n\t" \
      "L_noimp" SFX "_%=:\n\t"  /* SWUtil.scala:194-199 / native/ksw.c:455-461 */ \
      "s_cmp_lt_i32 %[zdrop], 1\n\t" \
      "s_cbranch_scc1 L_trim" SFX "_%=\n\t" \
      "s_sub_i32 %[t1], %[i], %[maxi]\n\t" \
      "s_sub_i32 %[t2], %[mja], %[maxj]\n\t" \
      "s_sub_i32 %[t1], %[t1], %[t2]\n\t" \
      "s_sub_i32 %[t2], %[mx], %[m]\n\t" \
      "s_cmp_gt_i32 %[t1], 0\n\t" \
      "s_cbranch_scc0 L_zneg" SFX "_%=\n\t" \
      "s_mul_i32 %[t4], %[t1], %[zpos]\n\t" \
      "s_add_i32 %[t4], %[t4], %[t2]\n\t" \
      "s_cmp_gt_i32 %[t4], %[zdrop]\n\t" \
      "s_cbranch_scc1 L_done" SFX "_%=\n\t" \
      "s_branch L_trim" SFX "_%=\n\t" \
      "L_zneg" SFX "_%=:\n\t" \
      "s_cmp_eq_u32 %[zneg], 0\n\t" \
      "s_cbranch_scc1 L_trim" SFX "_%=\n\t" \
      "s_mul_i32 %[t4], %[t1], %[eins]\n\t" \
      "s_add_i32 %[t4], %[t4], %[t2]\n\t" \
      "s_cmp_gt_i32 %[t4], %[zdrop]\n\t" \
      "s_cbranch_scc1 L_done" SFX "_%=\n\t" \
      "s_branch L_trim" SFX "_%=\n\t" \
      "L_tail" SFX "_%=:\n\t" \
      "s_mul_i32 %[t1], %[i], %[edel]\n\t" \
      "s_sub_i32 %[t1], %[u0], %[t1]\n\t" \
      "s_max_i32 %[t1], %[t1], %[qa]\n\t" \
      "s_cmp_le_i32 %[t1], %[mx]\n\t" \
      "s_cbranch_scc0 L_rowb" SFX "_%=\n\t" \
      "s_cmp_lt_i32 %[t1], %[gs]\n\t" \
      "s_cbranch_scc0 L_rowb" SFX "_%=\n\t" \
      "L_done" SFX "_%=:\n\t" \
      "s_mov_b32 %[reason], 0\n\t" \
      "s_branch L_out" SFX "_%=\n\t" \
      "L_slow" SFX "_%=:\n\t" \
      "s_mov_b32 %[reason], 3\n\t" \
      "L_out" SFX "_%=:\n\t"
// ---- the two-column loop, FAST form (round 4): as ROWS1F_TEXT -- same preconditions, the window of 128 columns -----------------
// In the DEAD phase of h1 the clamp of H at 0 replaces the whole injection of eh[beg].h (a move, two compares and two selects):
// the column left of the band hands on a 0 by itself -- to an even column through the lane shift (the odd column of the lane
// below), to an odd column as the even column of its own lane.  gscore reads column qLen - 1 out of the shifted row, whose lane and
// parity in the window (gsl, gsp) only change when the window moves.
// the two-column row and its gscore step as texts of their own: a fast loop has two bodies (ROWS1F_QBODY tells why)
#define ROWS2F_ROW(H1STEP, HMAX, HSHIFT) \
      "v_readlane_b32 %[t], %[vTS], %[i]\n\t" \
      "s_nop 1\n\t" \
      "v_bfe_i32 %[vS0], %[vP0], %[t], 8\n\t" \
      "v_bfe_i32 %[vS1], %[vP1], %[t], 8\n\t" \
      "v_add_u32 %[vA0], %[vH0], %[vS0]\n\t" \
      "v_add_u32 %[vA1], %[vH1], %[vS1]\n\t" \
      "v_max_i32 %[vA0], %[vA0], %[vE0]\n\t" \
      "v_max_i32 %[vA1], %[vA1], %[vE1]\n\t" \
      "v_cndmask_b32 %[vA0], %[vNEG], %[vA0], %[act0]\n\t" \
      "v_cndmask_b32 %[vA1], %[vNEG], %[vA1], %[act1]\n\t" \
      "v_sub_u32 %[vG0], %[vA0], %[vNegC]\n\t"  /* g of the even column */ \
      "v_sub_u32 %[vG], %[vA1], %[vNegC1]\n\t"  /* g of the odd column (vNegC1 = vNegC - eIns) */ \
      ROWS2F_KEY_AND_SCANS(H1STEP) \
      "v_add3_u32 %[vS0], %[vPp], %[vNegC], %[nkc]\n\t"  /* F of the even column */ \
      "v_max_i32 %[vS1], %[vPp], %[vG0]\n\t" \
      "v_add3_u32 %[vS1], %[vS1], %[vNegC], %[nkc1]\n\t"  /* F of the odd column */ \
      HMAX  /* H even, H odd */ \
      "v_cmp_gt_i32 vcc, 1, %[vA0]\n\t" \
      "s_and_b64 %[z0], vcc, %[act0]\n\t"  /* zero cells of the band, even columns */ \
      "v_cmp_gt_i32 vcc, 1, %[vA1]\n\t" \
      "s_and_b64 %[z1], vcc, %[act1]\n\t"  /* ... odd columns */ \
      "v_subrev_u32 %[vE0], %[edel], %[vE0]\n\t" \
      "v_subrev_u32 %[vS0], %[oedel], %[vA0]\n\t" \
      "v_max3_i32 %[vE0], %[vE0], %[vS0], 0\n\t" \
      "v_cndmask_b32 %[vE0], 0, %[vE0], %[act0]\n\t"  /* E(i+1,j); eh[end].e = 0 */ \
      "v_subrev_u32 %[vE1], %[edel], %[vE1]\n\t" \
      "v_subrev_u32 %[vS1], %[oedel], %[vA1]\n\t" \
      "v_max3_i32 %[vE1], %[vE1], %[vS1], 0\n\t" \
      "v_cndmask_b32 %[vE1], 0, %[vE1], %[act1]\n\t" \
      HSHIFT  /* eh[j].h = H(i,j-1): even column <- odd of the lane below, odd column <- even of the same lane; eh[beg].h = h1 */
#define ROWS2F_GSCORE(SFX) \
      "s_cmp_eq_u32 %[gsp], 0\n\t" \
      "s_cbranch_scc0 L_g2gsodd" SFX "_%=\n\t" \
      "v_lshl_or_b32 %[vS0], %[vH0], 16, %[i]\n\t"  /* H(i, qLen-1) << 16 | i */ \
      "s_nop 0\n\t" \
      "v_readlane_b32 %[t1], %[vS0], %[gsl]\n\t" \
      "s_max_i32 %[gskey], %[gskey], %[t1]\n\t" \
      "s_branch L_g2nogs" SFX "_%=\n\t" \
      "L_g2gsodd" SFX "_%=:\n\t" \
      "v_lshl_or_b32 %[vS0], %[vH1], 16, %[i]\n\t" \
      "s_nop 0\n\t" \
      "v_readlane_b32 %[t1], %[vS0], %[gsl]\n\t" \
      "s_max_i32 %[gskey], %[gskey], %[t1]\n\t" \
      "L_g2nogs" SFX "_%=:\n\t"
// the second body (LIVE phase only: a DEAD row's band moves and goes through the set-up): QS = the body's own label suffix
#define ROWS2F_QBODY(SFX, QS, H1STEP, HMAX, HSHIFT) \
      "L_g2bodyq" SFX "_%=:\n\t" \
      ROWS2F_ROW(H1STEP, HMAX, HSHIFT) \
      ROWS2F_GSCORE(QS) \
      ROWS2F_DECIDE(SFX) \
      "s_or_b64 %[u64], %[z0], %[z1]\n\t"  /* SCC = the band has a zero cell */ \
      "s_cbranch_scc1 L_g2zero" SFX "_%=\n\t" \
      "s_add_i32 %[i], %[i], 1\n\t" \
      "s_cmp_lt_i32 %[i], %[fastend]\n\t" \
      "s_cbranch_scc1 L_g2bodyq" SFX "_%=\n\t" \
      "s_branch L_g2bound" SFX "_%=\n\t"
#define ROWS2F_TEXT(SFX, H1STEP, HMAX, HSHIFT, NB0_NOZERO, NB0_ZERO, PHASE_MIN, PHASE_SWITCH, BAND_EXTRA, SAME_BAND, TAIL_MIN, TAIL_SWITCH, TAILTOP, QBODY) \
      /* the band's set-up: only when the band has changed (in the LIVE phase a row without a zero cell leaves beg where it is, and */ \
      /* end as well once it has reached the query end -- most rows of a flank whose score is high) */ \
      "L_g2row" SFX "_%=:\n\t" \
      "s_sub_i32 %[span], %[end], %[beg]\n\t" \
      "s_cmp_lt_i32 %[span], %[narrow1]\n\t" \
      "s_cbranch_scc1 L_fslow_%=\n\t"  /* an empty band, or one that fits one column per lane again */ \
      "s_sub_i32 m0, %[beg], %[base]\n\t"  /* rbeg */ \
      "v_subrev_u32 %[vT0], m0, %[vL2]\n\t"  /* rel0 = 2 lane - rbeg */ \
      "v_add_u32 %[vT1], 1, %[vT0]\n\t"  /* rel1 */ \
      "v_cmp_gt_u32 %[act0], %[span], %[vT0]\n\t" \
      "v_cmp_gt_u32 %[act1], %[span], %[vT1]\n\t" \
      BAND_EXTRA \
      "L_g2body" SFX "_%=:\n\t" \
      TAILTOP \
      ROWS2F_ROW(H1STEP, HMAX, HSHIFT) \
      /* SWUtil.scala:177-182 */ \
      "s_cmp_lg_u32 %[end], %[qlen]\n\t" \
      "s_cbranch_scc1 L_g2nogs" SFX "_%=\n\t" \
      ROWS2F_GSCORE(SFX) \
      ROWS2F_DECIDE(SFX) \
      "L_g2trim" SFX "_%=:\n\t"  /* band trimming, SWUtil.scala:202-214 */ \
      "s_or_b64 %[u64], %[z0], %[z1]\n\t"  /* SCC = the band has a zero cell */ \
      "s_cbranch_scc1 L_g2zero" SFX "_%=\n\t" \
      "L_g2nz" SFX "_%=:\n\t" \
      NB0_NOZERO \
      SAME_BAND \
      "s_cmp_lt_i32 %[end], %[qlen]\n\t" \
      "s_addc_u32 %[end], %[end], 0\n\t"  /* end = min(end + 1, qLen) */ \
      "L_g2next" SFX "_%=:\n\t" \
      "s_add_i32 %[i], %[i], 1\n\t" \
      "s_cmp_lt_i32 %[i], %[fastend]\n\t" \
      "s_cbranch_scc1 L_g2row" SFX "_%=\n\t" \
      /* row i is the first one this run may not sweep as it is: out of rows (the caller looks), or out of window */ \
      "L_g2bound" SFX "_%=:\n\t" \
      "s_add_i32 %[t1], %[beg], %[w1]\n\t"  /* rows up to beg + w: the left clamp cannot bind */ \
      "s_min_i32 %[hardend], %[rowend], %[t1]\n\t" \
      TAIL_MIN \
      PHASE_MIN \
      "s_cmp_lt_i32 %[i], %[hardend]\n\t" \
      "s_cbranch_scc1 L_g2win" SFX "_%=\n\t" \
      /* no row to run: the end of the target chunk, the query end, the end of the phase, or the clamp */ \
      "s_cmp_ge_i32 %[i], %[rowend]\n\t" \
      "s_cbranch_scc1 L_g2chunk" SFX "_%=\n\t" \
      TAIL_SWITCH \
      PHASE_SWITCH \
      "s_branch L_ftogen_%=\n\t" \
      "L_g2chunk" SFX "_%=:\n\t"  /* the next 64 target rows, when row i starts a chunk that holds no N */ \
      "s_cmp_ge_i32 %[i], %[tlen]\n\t" \
      "s_cbranch_scc1 L_fmore_%=\n\t" \
      "s_and_b32 %[t1], %[i], 63\n\t" \
      "s_cmp_lg_u32 %[t1], 0\n\t" \
      "s_cbranch_scc1 L_fmore_%=\n\t"  /* an N row ahead: the caller's */ \
      "s_add_i32 %[t1], %[tsaddr], %[i]\n\t" \
      "v_add_u32 %[vT0], %[t1], %[vLane]\n\t" \
      "ds_read_u8 %[vTS], %[vT0]\n\t" \
      "s_waitcnt lgkmcnt(0)\n\t" \
      "v_cmp_eq_u32 vcc, 32, %[vTS]\n\t" \
      "s_nop 4\n\t" \
      "s_cmp_lg_u64 vcc, 0\n\t" \
      "s_cbranch_scc1 L_fmore_%=\n\t"  /* an N row in the chunk (or stale bytes past the target's end that look like one): the caller's */ \
      "s_add_i32 %[rowend], %[i], 64\n\t" \
      "s_min_i32 %[rowend], %[rowend], %[tlen]\n\t" \
      "s_branch L_g2bound" SFX "_%=\n\t" \
      "L_g2win" SFX "_%=:\n\t" \
      "s_sub_i32 %[t1], %[end], %[base]\n\t" \
      "s_cmp_lt_i32 %[t1], 128\n\t" \
      "s_cbranch_scc1 L_g2room" SFX "_%=\n\t" \
      /* column `end` would fall outside the window: move the window up to the band's left end, an even column (as rows_cpp does) */ \
      "s_and_b32 %[t2], %[beg], -2\n\t"  /* the new base */ \
      "s_sub_i32 %[t1], %[end], %[t2]\n\t" \
      "s_cmp_gt_i32 %[t1], 127\n\t" \
      "s_cbranch_scc1 L_fslow_%=\n\t"  /* a band wider than the window: the caller reports the overflow */ \
      "s_sub_i32 %[t4], %[t2], %[base]\n\t" \
      "s_lshr_b32 %[t4], %[t4], 1\n\t"  /* lanes to move down */ \
      "v_add_lshl_u32 %[vT0], %[vLane], %[t4], 2\n\t"  /* byte address of the source lane; lanes past 63 wrap (don't-cares) */ \
      "v_add_u32 %[vS0], %[t2], %[vL2]\n\t" \
      "v_min_i32 %[vS0], %[vS0], %[qlen]\n\t" \
      "v_lshl_add_u32 %[vS0], %[vS0], 2, %[profaddr]\n\t"  /* the profile words of columns nb + 2 lane, + 1 (entries qLen, qLen + 1: past the query) */ \
      "ds_bpermute_b32 %[vH0], %[vT0], %[vH0]\n\t" \
      "ds_bpermute_b32 %[vE0], %[vT0], %[vE0]\n\t" \
      "ds_bpermute_b32 %[vH1], %[vT0], %[vH1]\n\t" \
      "ds_bpermute_b32 %[vE1], %[vT0], %[vE1]\n\t" \
      "ds_read_b32 %[vP0], %[vS0]\n\t" \
      "ds_read_b32 %[vP1], %[vS0] offset:4\n\t" \
      "s_mov_b32 %[base], %[t2]\n\t" \
      "s_sub_i32 %[t2], %[qlen], %[base]\n\t" \
      "s_lshr_b32 %[gsl], %[t2], 1\n\t" \
      "s_and_b32 %[gsp], %[t2], 1\n\t" \
      "s_waitcnt lgkmcnt(0)\n\t" \
      "L_g2room" SFX "_%=:\n\t"  /* t1 = end - base <= 127: 128 - t1 rows can run before `end` can leave the window ... */ \
      "s_sub_i32 %[t1], 128, %[t1]\n\t" \
      "s_add_i32 %[fastend], %[i], %[t1]\n\t" \
      "s_min_i32 %[fastend], %[fastend], %[hardend]\n\t" \
      ROWSF_ROOM_ATQ \
      "s_branch L_g2row" SFX "_%=\n\t" \
      "L_g2zero" SFX "_%=:\n\t"  /* on the even / odd zero masks (as ROWS2_TEXT) */ \
      /* last zero left of mj: even columns 2l < mj <=> l < (mj+1)>>1; odd columns 2l+1 < mj <=> l < mj>>1 */ \
      "s_add_i32 %[t5], %[mj], 1\n\t" \
      "s_lshr_b32 %[t6], %[t5], 1\n\t"  /* (mj+1)>>1 */ \
      "s_bfm_b64 %[u64], %[t6], 0\n\t" \
      "s_and_b64 %[u64], %[u64], %[z0]\n\t" \
      "s_flbit_i32_b64 %[t1], %[u64]\n\t"  /* -1: none, else 63 - lane */ \
      "s_lshr_b32 %[t4], %[mj], 1\n\t"  /* mj>>1 */ \
      "s_bfm_b64 %[u64], %[t4], 0\n\t" \
      "s_and_b64 %[u64], %[u64], %[z1]\n\t" \
      "s_flbit_i32_b64 %[t2], %[u64]\n\t" \
      "s_lshl_b32 %[m], %[t1], 1\n\t" \
      "s_sub_i32 %[m], 128, %[m]\n\t" \
      "s_cmp_lt_i32 %[t1], 0\n\t" \
      "s_cselect_b32 %[m], -1, %[m]\n\t" \
      "s_lshl_b32 %[t1], %[t2], 1\n\t" \
      "s_sub_i32 %[t1], 129, %[t1]\n\t" \
      "s_cmp_lt_i32 %[t2], 0\n\t" \
      "s_cselect_b32 %[t1], -1, %[t1]\n\t" \
      "s_max_i32 %[m], %[m], %[t1]\n\t"  /* cl + 2, or -1 */ \
      "s_add_i32 %[t1], %[m], %[base]\n\t" \
      NB0_ZERO \
      "s_cmp_lt_i32 %[m], 0\n\t" \
      "s_cselect_b32 %[beg], %[t3], %[t1]\n\t"  /* beg = base + cl + 2, or nb0 */ \
      /* first zero right of mj: even columns 2l > mj <=> l >= (mj+2)>>1; odd columns 2l+1 > mj <=> l >= (mj+1)>>1 = t6 */ \
      "s_and_b32 %[t5], %[t5], 1\n\t"  /* (mj+1) & 1 */ \
      "s_add_i32 %[t4], %[t6], %[t5]\n\t"  /* se = (mj+2)>>1 (at most 64: shifted in two steps) */ \
      "s_lshr_b64 %[u64], %[z0], %[t6]\n\t" \
      "s_lshr_b64 %[u64], %[u64], %[t5]\n\t" \
      "s_ff1_i32_b64 %[t1], %[u64]\n\t" \
      "s_lshr_b64 %[u64], %[z1], %[t6]\n\t" \
      "s_ff1_i32_b64 %[t2], %[u64]\n\t" \
      "s_add_i32 %[m], %[t4], %[t1]\n\t" \
      "s_lshl_b32 %[m], %[m], 1\n\t"  /* 2 (se + fe) */ \
      "s_cmp_lt_i32 %[t1], 0\n\t" \
      "s_cselect_b32 %[m], 0x100000, %[m]\n\t" \
      "s_add_i32 %[t1], %[t6], %[t2]\n\t" \
      "s_lshl_b32 %[t1], %[t1], 1\n\t" \
      "s_add_i32 %[t1], %[t1], 1\n\t"  /* 2 (so + fo) + 1 */ \
      "s_cmp_lt_i32 %[t2], 0\n\t" \
      "s_cselect_b32 %[t1], 0x100000, %[t1]\n\t" \
      "s_min_i32 %[m], %[m], %[t1]\n\t"  /* cr, or 0x100000 */ \
      "s_add_i32 %[t1], %[m], %[base]\n\t" \
      "s_add_i32 %[t1], %[t1], 1\n\t" \
      "s_add_i32 %[t2], %[end], 1\n\t" \
      "s_min_i32 %[t2], %[t2], %[qlen]\n\t" \
      "s_cmp_lt_i32 %[m], 0x100000\n\t" \
      "s_cselect_b32 %[end], %[t1], %[t2]\n\t"  /* end = base + cr + 1, or min(end + 1, qLen) */ \
      "s_branch L_g2next" SFX "_%=\n\t" \
      ROWS2F_OUTOFLINE(SFX) \
      "L_g2noimp" SFX "_%=:\n\t" \
      "s_cmp_lt_i32 %[mkey], 128\n\t" \
      "s_cbranch_scc1 L_fdone_%=\n\t"  /* m == 0                                SWUtil.scala:184-185 */ \
      "s_and_b32 %[mj], %[mkey], 127\n\t" \
      /* SWUtil.scala:194-199 (Scala parse) / native/ksw.c:455-461 (BWA parse): k = (i - max_i) - (mj - max_j), X = max - m.  First a */ \
      /* test every stop passes -- X + k * eIns > zdrop (Scala: that IS its test once k > 0) / X > zdrop (BWA: both its tests take */ \
      /* something off X) -- which nearly every row fails; zlim = zdrop, or a value no score reaches when the z-drop is off */ \
      "s_add_i32 %[mja], %[mj], %[base]\n\t" \
      "s_sub_i32 %[t1], %[i], %[maxi]\n\t" \
      "s_sub_i32 %[t2], %[mja], %[maxj]\n\t" \
      "s_sub_i32 %[t1], %[t1], %[t2]\n\t"  /* k */ \
      "s_sub_i32 %[t2], %[mxhi], %[mkey]\n\t" \
      "s_lshr_b32 %[t2], %[t2], 7\n\t"  /* X (the low 7 bits of mxhi are all ones: no borrow from the column) */ \
      "s_mul_i32 %[t4], %[t1], %[zc1]\n\t" \
      "s_add_i32 %[t4], %[t4], %[t2]\n\t" \
      "s_cmp_gt_i32 %[t4], %[zlim]\n\t" \
      "s_cbranch_scc0 L_g2trim" SFX "_%=\n\t" \
      "s_cmp_gt_i32 %[t1], 0\n\t" \
      "s_cbranch_scc0 L_g2zneg" SFX "_%=\n\t" \
      "s_mul_i32 %[t4], %[t1], %[zpos]\n\t" \
      "s_add_i32 %[t4], %[t4], %[t2]\n\t" \
      "s_cmp_gt_i32 %[t4], %[zlim]\n\t" \
      "s_cbranch_scc1 L_fdone_%=\n\t" \
      "s_branch L_g2trim" SFX "_%=\n\t" \
      "L_g2zneg" SFX "_%=:\n\t" \
      "s_cmp_eq_u32 %[zneg], 0\n\t" \
      "s_cbranch_scc1 L_g2trim" SFX "_%=\n\t" \
      "s_mul_i32 %[t4], %[t1], %[eins]\n\t" \
      "s_add_i32 %[t4], %[t4], %[t2]\n\t" \
      "s_cmp_gt_i32 %[t4], %[zlim]\n\t" \
      "s_cbranch_scc1 L_fdone_%=\n\t" \
      "s_branch L_g2trim" SFX "_%=\n\t" \
      QBODY
#define ROWS2F_LIVE_H1STEP "s_sub_i32 %[h1raw], %[h1raw], %[edel]\n\t"
#define ROWS2F_LIVE_HMAX "v_max_i32 %[vA0], %[vA0], %[vS0]\n\tv_max_i32 %[vA1], %[vA1], %[vS1]\n\t"
#define ROWS2F_LIVE_HSHIFT \
              "v_mov_b32 %[vh1], %[h1raw]\n\t" \
              "v_mov_b32_dpp %[vH0], %[vA1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
              "v_cndmask_b32 %[vH1], %[vA0], %[vh1], %[inj1]\n\t" \
              "v_cndmask_b32 %[vH0], %[vH0], %[vh1], %[inj0]\n\t"
#define ROWS2F_LIVE_(SFX, DEADSFX, TAIL_MIN, TAIL_SWITCH, TAILTOP, QT, QBODY) \
  ROWS2F_TEXT(SFX, ROWS2F_LIVE_H1STEP, ROWS2F_LIVE_HMAX, ROWS2F_LIVE_HSHIFT, \
              "", "s_mov_b32 %[t3], %[beg]\n\t", "s_min_i32 %[hardend], %[hardend], %[ih1z]\n\t", \
              "s_cmp_ge_i32 %[i], %[ih1z]\n\ts_cbranch_scc1 L_g2bound" DEADSFX "_%=\n\t", \
              /* the lanes that take eh[beg].h = h1: column beg is an even or an odd one */ \
              "v_cmp_eq_u32 %[inj0], 0, %[vT0]\n\tv_cmp_eq_u32 %[inj1], 0, %[vT1]\n\t", \
              /* end at the query end, beg untouched: the next row's band is this row's */ \
              "s_cmp_lt_i32 %[end], %[qlen]\n\t" \
              "s_cbranch_scc1 L_g2grow" SFX "_%=\n\t" \
              "s_add_i32 %[i], %[i], 1\n\t" \
              "s_cmp_lt_i32 %[i], %[fastend]\n\t" \
              "s_cbranch_scc1 " QT SFX "_%=\n\t" \
              "s_branch L_g2bound" SFX "_%=\n\t" \
              "L_g2grow" SFX "_%=:\n\t", TAIL_MIN, TAIL_SWITCH, TAILTOP, QBODY)
#define ROWS2F_LIVE_QBODY(SFX, QS) ROWS2F_QBODY(SFX, QS, ROWS2F_LIVE_H1STEP, ROWS2F_LIVE_HMAX, ROWS2F_LIVE_HSHIFT)
#define ROWS2F_DEAD_(SFX, TAIL_MIN, TAIL_SWITCH, TAILTOP) /* (no second body: a DEAD row's band moves every row) */ \
  ROWS2F_TEXT(SFX, "", \
              "v_max3_i32 %[vA0], %[vA0], %[vS0], 0\n\tv_max3_i32 %[vA1], %[vA1], %[vS1], 0\n\t", \
              "v_mov_b32_dpp %[vH0], %[vA1] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
              "v_mov_b32 %[vH1], %[vA0]\n\t", \
              "s_add_i32 %[beg], %[beg], 1\n\t", "s_add_i32 %[t3], %[beg], 1\n\t", "", "", "", "", TAIL_MIN, TAIL_SWITCH, TAILTOP, "")
#if BPSW_ROWS_ATQ
#define ROWS2F_ALL \
  ROWS2F_LIVE_("_l", "_d", ROWSF_TAILMIN, "s_cmp_ge_i32 %[i], %[itail]\n\ts_cbranch_scc1 L_g2bound_lt_%=\n\t", "", "L_g2bodyq", ROWS2F_LIVE_QBODY("_l", "_lq")) \
  ROWS2F_DEAD_("_d", ROWSF_TAILMIN, "s_cmp_ge_i32 %[i], %[itail]\n\ts_cbranch_scc1 L_g2bound_dt_%=\n\t", "") \
  ROWS2F_LIVE_("_lt", "_dt", "", "", ROWSF_TAILTEST("_lt2"), "L_g2body", "") \
  ROWS2F_DEAD_("_dt", "", "", ROWSF_TAILTEST("_dt2"))
#else
#define ROWS2F_ALL \
  ROWS2F_LIVE_("_l", "_d", ROWSF_TAILMIN, "s_cmp_ge_i32 %[i], %[itail]\n\ts_cbranch_scc1 L_g2bound_lt_%=\n\t", "", "L_g2body", "") \
  ROWS2F_DEAD_("_d", ROWSF_TAILMIN, "s_cmp_ge_i32 %[i], %[itail]\n\ts_cbranch_scc1 L_g2bound_dt_%=\n\t", "") \
  ROWS2F_LIVE_("_lt", "_dt", "", "", ROWSF_TAILTEST("_lt2"), "L_g2body", "") \
  ROWS2F_DEAD_("_dt", "", "", ROWSF_TAILTEST("_dt2"))
#endif
// all the instantiations in one statement, as ROWS1_ASM: sel 0 = the general loop, 1 = the fast loop in its LIVE phase, 2 = DEAD
#define ROWS2_ASM \
  asm volatile( \
      "s_mov_b32 %[form], %[sel]\n\t" \
      "s_cmp_eq_u32 %[sel], 1\n\t" \
      "s_cbranch_scc1 L_g2bound_l_%=\n\t" \
      "s_cmp_eq_u32 %[sel], 2\n\t" \
      "s_cbranch_scc1 L_g2bound_d_%=\n\t" \
      "s_cmp_eq_u32 %[sel], 3\n\t" \
      "s_cbranch_scc1 L_g2bound_lt_%=\n\t" \
      "s_cmp_eq_u32 %[sel], 4\n\t" \
      "s_cbranch_scc1 L_g2bound_dt_%=\n\t" \
      "s_cmp_eq_u32 %[tailrows], 0\n\t" \
      "s_cbranch_scc1 L_rowb_n_%=\n\t" \
      ROWS2_TEXT(ROWS_TAIL_TOP, "_t") \
      "s_branch L_end_%=\n\t" \
      ROWS2_TEXT("", "_n") \
      "s_branch L_end_%=\n\t" \
      ROWS2F_ALL \
      "L_ftotail_%=:\n\t"  /* the rows at and past the query end: the general loop with the tail-row test, in its own form of the state */ \
      ROWSF_MX_FROM_KEY \
      "s_ashr_i32 %[gs], %[gskey], 16\n\t" \
      "s_sext_i32_i16 %[maxie], %[gskey]\n\t" \
      "s_mov_b32 %[form], 0\n\t" \
      "s_branch L_row_t_%=\n\t" \
      "L_ftogen_%=:\n\t"  /* the left clamp may bind from here on: the general loop */ \
      ROWSF_MX_FROM_KEY \
      "s_ashr_i32 %[gs], %[gskey], 16\n\t" \
      "s_sext_i32_i16 %[maxie], %[gskey]\n\t" \
      "s_mov_b32 %[form], 0\n\t" \
      "s_branch L_row_t_%=\n\t"  /* (the instantiation with the tail-row test: it may run past the query end) */ \
      "L_fmore_%=:\n\t" \
      "s_mov_b32 %[reason], 1\n\t"  /* ROWS_MORE */ \
      "s_branch L_end_%=\n\t" \
      "L_fdone_%=:\n\t" \
      "s_mov_b32 %[reason], 0\n\t"  /* ROWS_DONE */ \
      "s_branch L_end_%=\n\t" \
      "L_fslow_%=:\n\t" \
      "s_mov_b32 %[reason], 3\n\t"  /* ROWS_SLOW */ \
      "L_end_%=:\n\t" \
      : [vH0] "+&v"(vH0), [vE0] "+&v"(vE0), [vH1] "+&v"(vH1), [vE1] "+&v"(vE1), [vPp] "+&v"(vPp), [i] "+&s"(s_i), [beg] "+&s"(s_beg), \
        [end] "+&s"(s_end), [h1raw] "+&s"(s_h1raw), [mx] "+&s"(s_mx), [maxi] "+&s"(s_maxi), [maxj] "+&s"(s_maxj), [maxie] "+&s"(s_maxie), \
        [gs] "+&s"(s_gs), [moff] "+&s"(s_moff), [mxhi] "+&s"(s_mxhi), [gskey] "+&s"(s_gskey), [base] "+&s"(s_base), [gsl] "+&s"(s_gsl), \
        [gsp] "+&s"(s_gsp), [vP0] "+&v"(vP0), [vP1] "+&v"(vP1), \
        [rowend] "+&s"(s_rowend), [vTS] "+&v"(vTS), [reason] "=&s"(reason), [fastend] "=&s"(s_fastend), [hardend] "=&s"(s_hardend), [form] "=&s"(s_form), [vS0] "=&v"(vS0), [vS1] "=&v"(vS1), [vA0] "=&v"(vA0), \
        [vA1] "=&v"(vA1), [vG0] "=&v"(vG0), [vG] "=&v"(vG), [vK] "=&v"(vK), [vT0] "=&v"(vT0), [vT1] "=&v"(vT1), [vh1] "=&v"(vh1), \
        [t] "=&s"(t), [h1] "=&s"(h1), [span] "=&s"(span), [mkey] "=&s"(mkey), [m] "=&s"(m), [mj] "=&s"(mj), [mja] "=&s"(mja), \
        [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3), [t4] "=&s"(t4), [t5] "=&s"(t5), [t6] "=&s"(t6), [act0] "=&s"(act0), \
        [act1] "=&s"(act1), [z0] "=&s"(z0), [z1] "=&s"(z1), [u64] "=&s"(u64), [inj0] "=&s"(inj0), [inj1] "=&s"(inj1) \
      : [vL2] "v"(vL2), [vL2p1] "v"(vL2p1), [vLane] "v"(lane), [vNegC] "v"(vNegC), [vNegC1] "v"(vNegC1), [vNEG] "v"(vNEG), \
        [qlen] "s"(qLen), [tlen] "s"(s_tlen), [tsaddr] "s"(s_tsaddr), [ih1z] "s"(s_ih1z), [w] "s"(w), [w1] "s"(s_w1), [edel] "s"(eDel), [oedel] "s"(oeDel), [nkc] "s"(s_nkc), \
        [nkc1] "s"(s_nkc1), [zdrop] "s"(zdrop), [zpos] "s"(s_zpos), [zneg] "s"(s_zneg), [eins] "s"(eIns), [itail] "s"(i_tail), [tailrows] "s"(s_tailrows), [u0] "s"(u0), \
        [qa] "s"(qa), [narrow1] "s"(ROWS_NARROW + 1), [sel] "s"(s_sel), [profaddr] "s"(s_profaddr), [zc1] "s"(s_zc1), [zlim] "s"(s_zlim) \
      : "vcc", "scc", "memory");  /* (M0 is written too) */
__device__ __forceinline__ int rows2_asm(RowState& st, const int lane_arg, const int qLen, const int row_end, int& vTS, const int w,
                                         const int eDel, const int oeDel, const int oeIns, const int eIns, const int zdrop, const int zmode,
                                         const int i_tail, const int u0, const int qa, const bool tail_rows, const int sel, const int tLen,
                                         const int i_h1z, const unsigned ts_addr, const unsigned prof_addr) {
  const int lane = rows_opaque(lane_arg);
  int vH0 = st.H0, vE0 = st.E0, vH1 = st.H1, vE1 = st.E1;
  int vP0 = st.plo0, vP1 = st.plo1;
  const int vL2 = 2 * lane, vL2p1 = 2 * lane + 1;  // the lane's columns, in window coordinates
  const int vNegC = -(2 * lane * eIns);   // g(k) = a(k) + k*eIns;  F(j) = Pex(j) - (j-1)*eIns - oeIns
  const int vNegC1 = vNegC - eIns;
  int vPp = NEG;
  int vNEG = rows_opaque(NEG_A);
  int s_i = st.i, s_beg = st.beg, s_end = st.end, s_h1raw = st.h1raw, s_mx = st.mx, s_maxi = st.max_i, s_maxj = st.max_j;
  int s_maxie = st.max_ie, s_gs = st.gscore, s_moff = st.max_off;
  int s_mxhi = (st.mx << 7) | 127, s_gskey = (int)(((unsigned)st.gscore << 16) | ((unsigned)st.max_ie & 0xffffu));  // (see rows1_asm)
  int s_base = st.base, s_gsl = (qLen - st.base) >> 1, s_gsp = (qLen - st.base) & 1;  // column qLen of the shifted row: lane, parity
  const int s_w1 = w + 1, s_nkc = eIns - oeIns, s_nkc1 = -oeIns;
  const int s_zpos = uni(zmode == BPSW_ZDROP_SCALA ? eIns : -eDel), s_zneg = uni(zmode == BPSW_ZDROP_SCALA ? 0 : 1);
  const int s_zc1 = uni(zmode == BPSW_ZDROP_SCALA ? eIns : 0), s_zlim = uni(zdrop > 0 ? zdrop : 0x3fffffff);  // the fast loops' first z-drop test
  int reason, s_fastend, s_hardend, s_form;
  int s_rowend = row_end;
  int vS0, vS1, vA0, vA1, vG0, vG, vK, vT0, vT1, vh1;
  int t, h1, span, mkey, m, mj, mja, t1, t2, t3, t4, t5, t6;
  unsigned long long act0, act1, z0, z1, u64, inj0, inj1;
  // (the general loop exists twice in the statement: the rows below the query end need no tail-row test at their top)
  const int s_tailrows = uni((int)tail_rows), s_sel = uni(sel), s_profaddr = uni((int)prof_addr), s_tlen = uni(tLen),
            s_ih1z = uni(i_h1z), s_tsaddr = uni((int)ts_addr);
  ROWS2_ASM
  st.H0 = vH0; st.E0 = vE0; st.H1 = vH1; st.E1 = vE1; st.plo0 = vP0; st.plo1 = vP1; st.base = s_base;
  st.i = s_i; st.beg = s_beg; st.end = s_end; st.h1raw = s_h1raw; st.max_i = s_maxi; st.max_j = s_maxj; st.max_off = s_moff;
  if (s_form) {
    st.mx = ROWSF_MX_OUT(s_mx, s_mxhi); st.gscore = s_gskey >> 16; st.max_ie = (int)(short)(s_gskey & 0xffff);
  } else {
    st.mx = s_mx; st.max_ie = s_maxie; st.gscore = s_gs;
  }
  return reason;
}

// Where the assembly loop may run to from row i: the end of the 64-row target chunk it has in a register (reloaded here when i has
// left it), the last target row, the first N row (the loops have no N-row path: a mask of the chunk's N rows replaces a test per row),
// (and, for the general loop, the caller adds the first row at the query end when i is still before it: the instantiation without
// the tail-row test serves those rows).
__device__ __forceinline__ int rows_asm_end(const int i, const int tLen, const int i_tail, const uint8_t* __restrict__ ts, const int lane,
                                            int* vTS, int* ts_chunk, unsigned long long* n_rows) {
  if ((i >> 6) != *ts_chunk) {  // 8 * target base of the 64 rows around row i, one per lane
    *ts_chunk = i >> 6;
    const int at = (*ts_chunk << 6) + lane;
    *vTS = at < tLen ? (int)ts[at] : 0;
    *n_rows = __builtin_amdgcn_ballot_w64(*vTS == 32);
  }
  int end = min(tLen, (*ts_chunk + 1) << 6);
  const unsigned long long ahead = *n_rows >> (i & 63);
  if (ahead) end = min(end, i + (int)__builtin_ctzll(ahead));
  // (the caller caps this at the query end for the general loop without the tail-row test; the fast loops change over themselves)
  return uni(end);
}

// The wide phase of a call (rows_cpp4): from a band that outgrew the 128-column window -- or a first row that already is wider -- until
// the call ends or the band fits two columns per lane again.  `q`: the (H,E) row four columns per lane, window origin st.base.
// (forceinline, like rows_cpp4: a real call would put RowState in memory, and the assembly loops take its fields as scalar operands)
__device__ __forceinline__ int rows_wide_phase(RowState& st, Row4& q, const int lane, const int qLen, const int tLen, const ProfLds& pl,
                                               const uint8_t* __restrict__ ts, const int oDel, const int eDel, const int oIns, const int eIns,
                                               const int w, const int zdrop, const int zmode, const int h0, const int amax) {
  const int r4 = rows_cpp4(st, q, lane, qLen, tLen, pl, ts, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax);
  if (r4 != ROWS_OTHER_MODE) return r4;
  // back to two columns per lane, window at the band's left end: new column nb + 2 lane + s sits in old lane (D + 2 lane) >> 2, slot
  // ((D + 2 lane) & 2) + s, with D = nb - base (even)
  const int nb = st.beg & ~1;
  const int x = nb - st.base + 2 * lane;
  const int src = (x >> 2) << 2;
  const bool hi = (x & 2) != 0;
#define ROWS4_M(c) const int hv##c = __builtin_amdgcn_ds_bpermute(src, q.H##c), ev##c = __builtin_amdgcn_ds_bpermute(src, q.E##c);
  ROWS4_EACH(ROWS4_M)
#undef ROWS4_M
  st.H0 = hi ? hv2 : hv0; st.H1 = hi ? hv3 : hv1;
  st.E0 = hi ? ev2 : ev0; st.E1 = hi ? ev3 : ev1;
  st.base = nb;
  rows_load_profile<2>(st, pl, qLen, lane);
  return ROWS_OTHER_MODE;
}

// SWExtend on the adaptive window, for flanks of up to 255 bases: one, two or -- bands wider than 127 columns, round 5 -- four columns
// per lane.  *overflow is set in ONE place only -- a band that is wider than 127 columns at the very moment it leaves the one-column
// layout (the two-column window could not take it): a band grows by a column a row, so this is not reachable from a 64-column window, and
// no input has been found that gets there (advisor, round 5).  Should it ever happen the launched short kernel defers the task to the full
// kernel; the resident one has no list to defer to, leaves the record unwritten, and the rings' integrity check (bpsw_ring.cpp) has the
// batch computed again through a launch.
template <class QC>
__device__ ExtRes sw_extend_adaptive(const int lane, const int qLen, const int tLen, const QC& qcode, const uint8_t* __restrict__ ts,
                                     const ProfLds& pl, const MatRows& mat, const int oDel, const int eDel, const int oIns, const int eIns, const int w,
                                     const int zdrop, const int zmode, const int h0, const int amax, int* __restrict__ overflow) {
  const int oeIns = oIns + eIns, oeDel = oDel + eDel;
#if defined(BPSW_DIAG_SKIP) && BPSW_DIAG_SKIP == 1  // instruction-count experiments only: the call returns before any setup
  return ExtRes{h0, 0, 0, 0, 0, 0};
#endif
  RowState st;
  st.i = 0; st.beg = 0; st.end = qLen; st.h1raw = h0 - oDel;
  st.mx = h0; st.max_i = -1; st.max_j = -1; st.max_ie = -1; st.gscore = -1; st.max_off = 0;  // SWUtil.scala:118-125
  st.base = 0;
  // row 0 spans min(qLen, w + 1) columns (+ the column `end` it writes)
  int cols = min(qLen, w + 1) <= 63 ? 1 : (min(qLen, w + 1) <= 127 ? 2 : 4);
  st.H1 = 0; st.E1 = 0; st.plo1 = 0;
  rows_build_profile(pl, qcode, mat, qLen, lane);
  if (cols == 4) {  // (w >= 127 and a flank of 128 bases or more: the first row is wider than the two-column window)
    Row4 q;
    st.base = 0;
    // row -1, SWUtil.scala:97-104
#define ROWS4_M(c) q.H##c = 4 * lane + c == 0 ? h0 : max(0, h0 - oeIns - (4 * lane + c - 1) * eIns); q.E##c = 0;
    ROWS4_EACH(ROWS4_M)
#undef ROWS4_M
    rows_load_profile4(q, 0, pl, qLen, lane);
    if (rows_wide_phase(st, q, lane, qLen, tLen, pl, ts, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax) != ROWS_OTHER_MODE) {
      ExtRes res0;
      res0.max = st.mx; res0.qle = st.max_j + 1; res0.tle = st.max_i + 1; res0.gtle = st.max_ie + 1; res0.gscore = st.gscore; res0.max_off = st.max_off;
      return res0;
    }
    cols = 2;
  } else if (cols == 1) {
    rows_load_profile<1>(st, pl, qLen, lane);
    st.H0 = lane == 0 ? h0 : max(0, h0 - oeIns - (lane - 1) * eIns);  // row -1, SWUtil.scala:97-104
    st.E0 = 0;
  } else {
    rows_load_profile<2>(st, pl, qLen, lane);
    const int j0 = 2 * lane, j1 = 2 * lane + 1;
    st.H0 = j0 == 0 ? h0 : max(0, h0 - oeIns - (j0 - 1) * eIns);
    st.H1 = max(0, h0 - oeIns - (j1 - 1) * eIns);
    st.E0 = 0; st.E1 = 0;
  }
  const int i_tail = amax > 0 ? qLen : 0x7fffffff;
  const int u0 = h0 + qLen * amax - oDel + (qLen - 1) * eDel, qa = qLen * amax;  // tail_row_bound(i) = max(u0 - i*eDel, qa)
  // the first row whose h1 = max(0, h0 - oDel - eDel (i + 1)) is zero (SWUtil.scala:137-138): ceil((h0 - oDel) / eDel) - 1, at least 0
  const int i_h1z = uni(h0 - oDel > 0 ? (h0 - oDel + eDel - 1) / eDel - 1 : 0);
  int vTS = 0, ts_chunk = -1;
  unsigned long long n_rows = 0ull;  // the N rows of the target chunk in vTS
  const unsigned ts_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) const uint8_t*)ts);  // for the loops' own chunk reload
#if defined(BPSW_DIAG_SKIP) && BPSW_DIAG_SKIP == 2  // ... after the setup, before the first row
  return ExtRes{h0 + st.H0 * 0, 0, 0, 0, 0, 0};
#endif
  for (;;) {
    int r;
    if (cols == 1) {
#if BPSW_EXT_ROWS_ASM
      if (st.i >= tLen) break;
      r = ROWS_SLOW;
      if (const int row_end = rows_asm_end(st.i, tLen, i_tail, ts, lane, &vTS, &ts_chunk, &n_rows); row_end > st.i) {
        // the band clamp of row i (SWUtil.scala:140-142; idempotent), then how far the fast loop may run (ROWS1F_TEXT): while the
        // left clamp cannot bind (it looks after the window, the target chunks, the change of h1's phase and the hand-over to the
        // general loop at the query end itself)
        st.beg = smax2(st.beg, st.i - w);
        st.end = smin2(smin2(st.end, st.i + w + 1), qLen);
        const bool live = st.i < i_h1z;
        // (the tail rows' bound U <= u0, qa must fit the loops' 16-bit forms of gscore)
        const bool tail = st.i >= i_tail;
        const int sel = uni((BPSW_EXT_ROWS_FAST && st.beg + w + 1 > st.i && (!tail || (u0 < 30000 && qa < 30000))) ? (live ? 1 : 2) + (tail ? 2 : 0) : 0);
        r = rows1_asm(st, lane, qLen, (sel == 0 && !tail) ? smin2(row_end, i_tail) : row_end, vTS, w, eDel, oeDel, oeIns, eIns, zdrop, zmode, i_tail, u0, qa, st.i >= i_tail, sel, tLen, i_h1z, ts_addr, pl.addr);
        if (r == ROWS_MORE) continue;
      }
      if (r == ROWS_SLOW)
#endif
        r = rows_cpp<1>(st, lane, qLen, tLen, pl, ts, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax, BPSW_EXT_ROWS_ASM ? 1 : 0x7fffffff);
      if (r == ROWS_OTHER_MODE) {  // the band outgrew 64 columns: two columns per lane, window at the band's left end
        const int nb = st.beg & ~1;
        if (st.end - nb > 127) { *overflow = 1; return ExtRes{0, 0, 0, 0, 0, 0}; }
        const int src0 = (nb - st.base + 2 * lane) << 2, src1 = src0 + 4;  // lanes past 63 wrap: columns the band has not reached
        const int h0v = __builtin_amdgcn_ds_bpermute(src0, st.H0), h1v = __builtin_amdgcn_ds_bpermute(src1, st.H0);
        const int e0v = __builtin_amdgcn_ds_bpermute(src0, st.E0), e1v = __builtin_amdgcn_ds_bpermute(src1, st.E0);
        st.H0 = h0v; st.H1 = h1v; st.E0 = e0v; st.E1 = e1v;
        st.base = nb;
        rows_load_profile<2>(st, pl, qLen, lane);
        cols = 2;
        continue;
      }
    } else {
#if BPSW_EXT_ROWS_ASM
      if (st.i >= tLen) break;
      r = ROWS_SLOW;
      if (const int row_end = rows_asm_end(st.i, tLen, i_tail, ts, lane, &vTS, &ts_chunk, &n_rows); row_end > st.i) {
        st.beg = smax2(st.beg, st.i - w);  // (as for the one-column loop above)
        st.end = smin2(smin2(st.end, st.i + w + 1), qLen);
        const bool live = st.i < i_h1z;
        // (the tail rows' bound U <= u0, qa must fit the loops' 16-bit forms of gscore)
        const bool tail = st.i >= i_tail;
        const int sel = uni((BPSW_EXT_ROWS_FAST && st.beg + w + 1 > st.i && (!tail || (u0 < 30000 && qa < 30000))) ? (live ? 1 : 2) + (tail ? 2 : 0) : 0);
        r = rows2_asm(st, lane, qLen, (sel == 0 && !tail) ? smin2(row_end, i_tail) : row_end, vTS, w, eDel, oeDel, oeIns, eIns, zdrop, zmode, i_tail, u0, qa, st.i >= i_tail, sel, tLen, i_h1z, ts_addr, pl.addr);
        if (r == ROWS_MORE) continue;
      }
      if (r == ROWS_SLOW)
#endif
        r = rows_cpp<2>(st, lane, qLen, tLen, pl, ts, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax, BPSW_EXT_ROWS_ASM ? 1 : 0x7fffffff);
      if (r == ROWS_OVERFLOW) {
        // the band outgrew 128 columns: four columns per lane (round 4 gave the task up here: *overflow).  New column nb + 4 lane + c
        // sits in old lane d + 2 lane + (c >> 1), slot c & 1, with d = (nb - base) / 2
        Row4 q;
        const int nb = st.beg & ~3;
        const int srcA = (((nb - st.base) >> 1) + 2 * lane) << 2, srcB = srcA + 4;
        q.H0 = __builtin_amdgcn_ds_bpermute(srcA, st.H0); q.H1 = __builtin_amdgcn_ds_bpermute(srcA, st.H1);
        q.H2 = __builtin_amdgcn_ds_bpermute(srcB, st.H0); q.H3 = __builtin_amdgcn_ds_bpermute(srcB, st.H1);
        q.E0 = __builtin_amdgcn_ds_bpermute(srcA, st.E0); q.E1 = __builtin_amdgcn_ds_bpermute(srcA, st.E1);
        q.E2 = __builtin_amdgcn_ds_bpermute(srcB, st.E0); q.E3 = __builtin_amdgcn_ds_bpermute(srcB, st.E1);
        st.base = nb;
        rows_load_profile4(q, nb, pl, qLen, lane);
        if (rows_wide_phase(st, q, lane, qLen, tLen, pl, ts, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax) != ROWS_OTHER_MODE) break;
        ts_chunk = -1;
        continue;
      }
      if (r == ROWS_OTHER_MODE) {  // the band fits one column per lane again: window at its left end
        const int nb = st.beg;
        const int col = nb + lane - st.base;  // this lane's new column, in the old window
        const int src = (col >> 1) << 2;
        const int ha = __builtin_amdgcn_ds_bpermute(src, st.H0), hb = __builtin_amdgcn_ds_bpermute(src, st.H1);
        const int ea = __builtin_amdgcn_ds_bpermute(src, st.E0), eb = __builtin_amdgcn_ds_bpermute(src, st.E1);
        st.H0 = (col & 1) ? hb : ha;
        st.E0 = (col & 1) ? eb : ea;
        st.base = nb;
        rows_load_profile<1>(st, pl, qLen, lane);
        cols = 1;
        ts_chunk = -1;
        continue;
      }
    }
    if (r == ROWS_DONE) break;
  }
  ExtRes res;
  res.max = st.mx; res.qle = st.max_j + 1; res.tle = st.max_i + 1; res.gtle = st.max_ie + 1; res.gscore = st.gscore; res.max_off = st.max_off;
  return res;
}

}  // namespace
}  // namespace bpsw
