// bpsw_extend_sift_core.h -- the arithmetic of the sift kernel (bpsw_extend_sift.hip: the exact shortcuts of the extension, one task per
// lane): what a lane does with the nibble streams of its task, free of anything the device alone has, so that the same code also
// compiles for the host -- tests/sift_host runs it on a CPU against the oracle's full DP (tests/test_sift_host.py), the GPU tests run
// it where it ships.  The forms and their proofs are in bpsw_extend_core.h; the head of bpsw_extend_sift.hip says how the
// certificate is evaluated without scans.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BPSW_HD __host__ __device__ __forceinline__
#else
#define BPSW_HD inline
#endif

namespace bpsw {
namespace sift {

BPSW_HD int sift_max(int a, int b) { return a > b ? a : b; }
BPSW_HD int sift_min(int a, int b) { return a < b ? a : b; }

// one side of a task: the nibble streams (LDS; first base in the top nibble of word 0) that hold its query and its target flank, and
// where each starts.  A wire batch of format 1 keeps both in the task's one stream; a coordinate batch has the target flank in a
// block of its own, expanded from the 2-bit reference.
struct SiftSeq {
  const uint32_t* qraw;
  int qs;
  const uint32_t* traw;
  int ts;
  // the eight bases from base k of a stream on
  static BPSW_HD uint32_t at8(const uint32_t* raw, int k) {
    const int wi = k >> 3;
    const unsigned long long v = ((unsigned long long)raw[wi] << 32) | raw[wi + 1];
    return (uint32_t)((v << ((k & 7) << 2)) >> 32);
  }
  static BPSW_HD int at1(const uint32_t* raw, int k) { return (int)((raw[k >> 3] >> (28 - 4 * (k & 7))) & 0xFu); }
  BPSW_HD uint32_t q8(int k) const { return at8(qraw, qs + k); }
  BPSW_HD uint32_t t8(int k) const { return at8(traw, ts + k); }
  BPSW_HD int qn(int k) const { return at1(qraw, qs + k); }
  BPSW_HD int tn(int k) const { return at1(traw, ts + k); }
};

// the top `cnt` nibbles of a word (cnt >= 1; 8 and more: all of it)
BPSW_HD uint32_t top_nibbles(int cnt) { return cnt >= 8 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (32 - 4 * cnt); }
// one flag (bit 0 of its nibble) per nibble in which two words of codes 0..3 differ
BPSW_HD uint32_t differ(uint32_t x, uint32_t y) {
  const uint32_t v = x ^ y;
  return (v | (v >> 1)) & 0x11111111u;
}

enum { SIFT_UNSEEN = 0, SIFT_FAIL = 1, SIFT_FORM = 2 };
struct SideRec {
  int kind, hmin;                                   // SIFT_FORM: the side is resolved when its start score is >= hmin
  int max_rel, g_rel, qle, tle, gtle, max_off;      // max - hInit, gscore - hInit, and the rest of ExtRes
};

struct SiftParams {
  int a, dm;                 // match score, a - (mismatch score)
  int oDel, eDel, oIns, eIns, zdrop, certify, wBand;
};

// how many of the columns lo..hi have q[y + dq] != t[y + dt]; stops counting at `cap` (lo <= hi)
BPSW_HD int sift_count_differ(const SiftSeq& s, const int lo, const int hi, const int dq, const int dt, const int cap) {
  int cnt = 0;
  for (int k = lo; k <= hi && cnt < cap; k += 8) cnt += __builtin_popcount(differ(s.q8(k + dq), s.t8(k + dt)) & top_nibbles(hi + 1 - k));
  return cnt;
}
BPSW_HD bool sift_equal_run(const SiftSeq& s, const int lo, const int hi, const int dq, const int dt) {
  return sift_count_differ(s, lo, hi, dq, dt, 1) == 0;
}

// One shift of single_gap_certificate (bpsw_extend_core.h) for a flank without N whose main diagonal mismatches in columns
// p0 < p1 < p2 (the first k of them, k <= 3; D = k dm): see the head of this file.  `ins`: one insertion of d query bases, then the
// diagonal shifted right by d; else one deletion of d target bases, then the diagonal shifted down by d.  W is only ever above 0 in
// a short run of columns behind a gain, so the columns are walked one by one there and nowhere else; where W = 0 the deletion
// condition is -tail(x) < T, which depends on the deficit columns alone.  The shifts of a flank are independent: the kernel spreads
// (flank, shift) pairs over the lanes of the wavefront.
BPSW_HD bool sift_certificate_shift(const SiftSeq& s, const int n, const int tLen, const SiftParams& P, const int k, const int p0,
                                       const int p1, const int p2, const bool ins, const int d) {
  const int a = P.a, dm = P.dm;
  const auto deficits_in = [&](const int lo, const int hi) {  // deficit columns in lo..hi
    return (int)(k > 0 && p0 >= lo && p0 <= hi) + (int)(k > 1 && p1 >= lo && p1 <= hi) + (int)(k > 2 && p2 >= lo && p2 <= hi);
  };
  const auto col = [&](const int i) { return i == 0 ? p0 : (i == 1 ? p1 : p2); };
  if (ins) {
    if (d >= n) return true;
    const int xl = n - 1 - d, T = P.oIns + d * P.eIns;
    const int tail_main = a * d - dm * deficits_in(xl + 1, n - 1);  // A(n-1) - A(xl)
    int W = 0, x = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int p = col(i);
      if (i < k && p <= xl) {
        while (W > 0 && x < p) { if (s.qn(x + d) != s.tn(x)) --W; ++x; }  // losses
        if (s.qn(p + d) == s.tn(p)) {                                      // a gain
          if (dm * (W + 1) >= T) return false;
          ++W;
        }
        x = p + 1;
      }
    }
    while (W > 0 && x <= xl) { if (s.qn(x + d) != s.tn(x)) --W; ++x; }
    return dm * W - tail_main <= T;  // (W = W(xl): the walk ended at xl, or at 0 before it)
  }
  const int T = P.oDel + d * P.eDel;
  const int xmax = sift_min(n - 1, tLen - d - 1);  // the columns whose shifted cell exists
  if (xmax < 0) return true;
  const auto tail = [&](const int x) {  // A(min(x+d, n-1)) - A(x)
    const int z = sift_min(x + d, n - 1);
    return a * (z - x) - dm * deficits_in(x + 1, z);
  };
  // where W = 0: -tail(x) < T for every column; -tail rises only where a deficit column enters the tail (x = p - d), and in the
  // last d columns (the tail gets shorter) up to the last column or the one before a deficit column leaves it
  if (dm * sift_min(k, d) >= T) {  // (else no window of d columns holds enough deficit)
    if (-tail(0) >= T || -tail(xmax) >= T) return false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < k) {
        const int p = col(i);
        if (p - d >= 0 && p - d <= xmax && -tail(p - d) >= T) return false;
        if (p - 1 >= 0 && p - 1 <= xmax && -tail(p - 1) >= T) return false;
      }
    }
  }
  // where W > 0: behind a gain, column by column
  int W = 0, x = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = col(i);
    if (i < k && p <= xmax) {
      while (W > 0 && x < p) {
        if (s.tn(x + d) != s.qn(x)) --W;
        if (W > 0 && dm * W - tail(x) >= T) return false;
        ++x;
      }
      if (s.tn(p + d) == s.qn(p)) ++W;
      if (W > 0 && dm * W - tail(p) >= T) return false;
      x = p + 1;
    }
  }
  while (W > 0 && x <= xmax) {
    if (s.tn(x + d) != s.qn(x)) --W;
    if (W > 0 && dm * W - tail(x) >= T) return false;
    ++x;
  }
  return true;
}

enum { CF_HOLDS = 0, CF_IF_CERTIFIED = 1, CF_FAILS = 2, CF_UNSEEN = 3 };
constexpr int SIFT_MAX_SHIFTS = 16;  // shifts per direction the certificate lanes take (default scoring: 9); more: left to ext_kernel

// flank_closed_form of bpsw_extend_core.h for one side up to its certificate, with everything that depends on the start score
// factored out (hmin, *_rel).  n = qLen (1..127), tLen = the target flank's length; the sequences hold no N.  CF_IF_CERTIFIED: `rec`
// is the result provided every shift 1..dI (insertion) and 1..dD (deletion) passes sift_certificate_shift for the deficit columns
// p[0..k).
BPSW_HD int sift_closed_form(const SiftSeq& s, const int n, const int tLen, const SiftParams& P, SideRec* rec, int* k_out, int* p,
                                int* dI, int* dD) {
  if (tLen < n) return CF_FAILS;
  const int a = P.a, dm = P.dm;
  const int oe_min = sift_min(P.oIns + P.eIns, P.oDel + P.eDel);
  const bool family = a == 1 && P.eIns == 1 && P.eDel == 1 && P.oIns + P.eIns == P.oDel + P.eDel && oe_min >= 2;
  const bool two_opens = P.certify >= 2 && family;  // (n <= 128 always here)
  const int limit = P.certify ? (two_opens ? 2 * oe_min + 2 : 2 * oe_min) : oe_min;
  int D = 0, best_rel = 0, best_i = -1, k = 0;
  int p_last = -1, p_prev = -1, p_prev2 = -1;
  for (int j = 0; j < n; j += 8) {
    uint32_t m = differ(s.q8(j), s.t8(j)) & top_nibbles(n - j);
    while (m) {  // the (very few) diagonal cells that are not a match
      const int i = __builtin_clz(m) >> 2;
      m &= ~(0x10000000u >> (4 * i));
      const int pos = j + i;
      const int v = pos * a - D;  // m(pos-1) - h0: the last row before this deficit
      if (pos >= 1 && v > best_rel) { best_rel = v; best_i = pos - 1; }
      D += dm;
      p_prev2 = p_prev; p_prev = p_last; p_last = pos;
      ++k;
      if (D >= limit) return CF_FAILS;
    }
  }
  if (P.zdrop > 0 && D > P.zdrop) return CF_FAILS;
  *dI = D >= oe_min ? sift_max(0, (D - P.oIns) / P.eIns) : 0;  // (a deficit below the dearer of the two gap opens: no shift of that kind)
  *dD = D >= oe_min ? sift_max(0, (D - P.oDel) / P.eDel) : 0;
  // (a mismatch score so mild that four deficit columns stay below two gap opens, or gap costs that ask for very long shifts)
  if (D >= oe_min && (k > 3 || *dI > SIFT_MAX_SHIFTS || *dD > SIFT_MAX_SHIFTS)) return CF_UNSEEN;
  if (D >= 2 * oe_min) {  // two gap opens: tests 1 and 2 of flank_closed_form
    const auto is_match = [&](const int ti, const int qi) {
      return ti >= 0 && qi >= 0 && ti < tLen && qi < n && s.tn(ti) == s.qn(qi);
    };
    for (int L = 2; L <= 2 + (D - 2 * oe_min); ++L) {
      if (is_match(p_last, p_last + L)) {
        bool m1 = p_prev < 0, m2 = p_prev2 < 0;
        for (int sft = 1; sft <= L; ++sft) { m1 = m1 || is_match(p_prev, p_prev + sft); m2 = m2 || is_match(p_prev2, p_prev2 + sft); }
        if (m1 && m2) return CF_FAILS;
      }
      if (is_match(n - 1 + L, n - 1)) {
        bool m0 = false, m1 = p_prev < 0, m2 = p_prev2 < 0;
        for (int sft = 1; sft <= L; ++sft) {
          m0 = m0 || is_match(p_last + sft, p_last); m1 = m1 || is_match(p_prev + sft, p_prev); m2 = m2 || is_match(p_prev2 + sft, p_prev2);
        }
        if (m0 && m1 && m2) return CF_FAILS;
      }
    }
  }
  const int g_rel = n * a - D;
  if (g_rel > best_rel) { best_rel = g_rel; best_i = n - 1; }
  rec->kind = SIFT_FORM; rec->hmin = D + 1;
  rec->max_rel = best_rel; rec->g_rel = g_rel; rec->qle = best_i + 1; rec->tle = best_i + 1; rec->gtle = n; rec->max_off = 0;
  if (D < oe_min) return CF_HOLDS;
  *k_out = k;  // ascending deficit columns
  p[0] = k == 1 ? p_last : (k == 2 ? p_prev : p_prev2); p[1] = k == 2 ? p_last : p_prev; p[2] = p_last;
  return CF_IF_CERTIFIED;
}

// flank_start_gap_form of bpsw_extend_core.h (tried when the closed form does not hold); false: no form holds
BPSW_HD bool sift_start_gap_form(const SiftSeq& s, const int n, const int tLen, const SiftParams& P, SideRec* rec) {
  const int a = P.a, dm = P.dm;
  const int oe = P.oIns + P.eIns;
  const bool family = a == 1 && P.eIns == 1 && P.eDel == 1 && oe == P.oDel + P.eDel && oe >= 2;
  if (P.certify < 3 || !family) return false;
  if (P.wBand < 4 || (P.zdrop > 0 && P.zdrop < oe) || n < oe + 3) return false;
  bool ins = tLen >= n - 1, del = tLen >= n + 1, del2 = tLen >= n + 2;
  if (!ins && !del) return false;
  // the shifted diagonals: t[j] == q[j+1] (j <= n-2), t[j+1] == q[j], t[j+2] == q[j] (j <= n-1)
  if (ins) ins = sift_equal_run(s, 0, n - 2, 1, 0);
  if (del) del = sift_equal_run(s, 0, n - 1, 0, 1);
  if (!ins && !del) return false;
  if (del2) del2 = sift_equal_run(s, 0, n - 1, 0, 2);
  if (ins && (del || del2)) return false;
  // the main diagonal must never get back above h0: S(j) = a(j+1) - dm * (mismatches up to j) <= 0 for every j < sift_min(n, tLen);
  // S peaks on the last base of a run of matches
  {
    const int nt = sift_min(n, tLen);
    int mm = 0;
    for (int j = 0; j < nt; j += 8) {
      uint32_t m = differ(s.q8(j), s.t8(j)) & top_nibbles(nt - j);
      while (m) {
        const int i = __builtin_clz(m) >> 2;
        m &= ~(0x10000000u >> (4 * i));
        const int pos = j + i;
        if (pos * a - dm * mm > 0) return false;  // S(pos - 1)
        ++mm;
      }
    }
    if (nt * a - dm * mm > 0) return false;
  }
  if (del) {
    const int s0 = s.tn(0) == s.qn(0) ? a : a - dm;
    if (s0 + oe + 1 <= 0) return false;
    if (s.tn(0) == s.qn(1) || s.tn(0) == s.qn(2)) return false;  // n >= 5 here
    const int g_rel = -oe + a * n;
    rec->kind = SIFT_FORM; rec->hmin = sift_max(2 * oe + 1, a - s0 + 1);  // h0 >= 2 oe + 1 and h0 + s0 > a
    rec->max_rel = g_rel; rec->g_rel = g_rel; rec->qle = n; rec->tle = n + 1; rec->gtle = n + 1; rec->max_off = 1;
    return true;
  }
  const int g_rel = -oe + a * (n - 1);
  rec->kind = SIFT_FORM; rec->hmin = 2 * oe + 1;
  rec->max_rel = g_rel; rec->g_rel = g_rel; rec->qle = n; rec->tle = n - 1; rec->gtle = n - 1; rec->max_off = 1;
  return true;
}

// extension(), MemChainToAlignBatched.scala:789-883, over two sides whose verdicts are in (as ext_kernel chains them, bpsw_extend.hip):
// the record of a task both of whose sides are resolved for the start scores they get.  false: some side is not.
struct SiftTask {
  int lq, rq, regScore0, qBeg, h0, idx, penClip5, penClip3, wBand;
};
BPSW_HD bool sift_chain(const SiftTask& T, const SideRec& sr0, const SideRec& sr1, uint32_t o[5]) {
  int regScore = T.regScore0;
  int outQBeg = 0, outRBeg = 0, outQEnd = T.rq, outREnd = 0, trueScore = T.regScore0, score = -1;
  for (int side = 0; side < 2; ++side) {
    const int qLen = side ? T.rq : T.lq;
    if (qLen <= 0) continue;
    const SideRec& r = side ? sr1 : sr0;
    const int hInit = side ? regScore : T.h0;
    if (r.kind != SIFT_FORM || hInit < r.hmin) return false;
    const int penClip = side ? T.penClip3 : T.penClip5;
    const int sc0 = regScore;
    const int rmax = hInit + r.max_rel, gscore = hInit + r.g_rel;
    regScore = rmax;
    score = regScore;
    const bool local = gscore <= 0 || gscore <= regScore - penClip;
    if (side == 0) {
      outQBeg = local ? T.qBeg - r.qle : 0;
      outRBeg = local ? -r.tle : -r.gtle;
      trueScore = local ? regScore : gscore;
    } else {
      outQEnd = local ? r.qle : T.rq;
      outREnd = local ? r.tle : r.gtle;
      trueScore += (local ? regScore : gscore) - sc0;
    }
  }
  o[0] = (uint32_t)T.idx;
  o[1] = ((uint32_t)outQBeg & 0xffffu) | ((uint32_t)outQEnd << 16);
  o[2] = ((uint32_t)outRBeg & 0xffffu) | ((uint32_t)outREnd << 16);
  o[3] = ((uint32_t)score & 0xffffu) | ((uint32_t)trueScore << 16);
  o[4] = (uint32_t)T.wBand & 0xffffu;
  return true;
}

}  // namespace sift
}  // namespace bpsw
