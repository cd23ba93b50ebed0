// bpsw_extend_core.h -- the SWExtend device routines (SWUtil.scala:61-230) shared by the extension kernels:
// ext_kernel (bpsw_extend.hip, tasks from the boundary-2 wire batch) and chain2aln_kernel (bpsw_chain2aln.hip, tasks
// generated on the device from seed chains).  Everything sits in an unnamed namespace: each translation unit gets its own
// copy and the compiler inlines it into the kernel.
#pragma once
#include "bpsw_internal.h"
#include "bpsw_wave.h"

namespace bpsw {
namespace {

// base k (0-based) of a task's nibble stream: 8 nibbles per word, first base in the top nibble
__device__ __forceinline__ int nibble_at(const uint32_t* __restrict__ words, int k) {
  const uint32_t w = words[k >> 3];
  const int c = (int)((w >> (28 - 4 * (k & 7))) & 0xFu);
  return c > 4 ? 4 : c;  // codes are 0..4 (LocusEncode); never index the matrix out of bounds
}

struct ExtRes {
  int max, qle, tle, gtle, gscore, max_off;
};

// Rows past the end of the query that cannot change the result.  For i >= qLen every cell (i,j) has j <= qLen-1 < i, so a
// path into it deletes at least i-j target bases in one or more runs (>= oDel + (i-j)*eDel) and takes at most j+1 diagonal
// steps of at most amax = max(mat) each, starting from h0 or -- restarted at a zero cell -- from 0; trimming only lowers
// values (a column that re-enters the band reads E = 0).  Hence, with amax > 0,
//     H(i,j) <= U(i) = max(h0 + qLen*amax - oDel - (i-qLen+1)*eDel, qLen*amax)          for all i >= qLen,
// and U is non-increasing in i while max and gscore never decrease.  Once U(i) <= max and U(i) < gscore, no later row can
// satisfy `m > max` (SWUtil.scala:187) or `gscore <= h1` (SWUtil.scala:178): max, max_i, max_j, max_off, gscore and max_ie
// are final, and the remaining rows could only end the loop.  The target flank is about twice as long as the query
// (calMaxGap), so for a good alignment this removes almost half of the rows.  Checked with the oracle by truncating the
// target at the first such row on 8 000 flanks (repeats, restarts after unrelated sequence, long deletions, low h0, five
// gap-cost sets, both parses: identical outputs), and by every parity test.
__device__ __forceinline__ int tail_row_bound(const int qLen, const int i, const int h0, const int amax, const int oDel,
                                              const int eDel) {
  return max(h0 + qLen * amax - oDel - (i - qLen + 1) * eDel, qLen * amax);
}

// One SWExtend call (SWUtil.scala:61-230) executed by a whole wave.  All scalar state is wave-uniform.
__device__ ExtRes sw_extend_wave(const int lane, const int qLen, const int tLen, int2* __restrict__ eh,
                                 const int8_t* __restrict__ qp, const uint8_t* __restrict__ ts, const int oDel,
                                 const int eDel, const int oIns, const int eIns, const int w, const int zdrop,
                                 const int zmode, const int h0, const int amax) {
  const int oeDel = oDel + eDel, oeIns = oIns + eIns;
  // row -1 (SWUtil.scala:75-78, 97-104): eh[0].h = h0, eh[j].h = max(0, h0 - oeIns - (j-1)*eIns), e = 0
  for (int j = lane; j <= qLen; j += 64) {
    const int h = j == 0 ? h0 : max(0, h0 - oeIns - (j - 1) * eIns);
    eh[j] = make_int2(h, 0);
  }
  __builtin_amdgcn_wave_barrier();

  int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;  // SWUtil.scala:118-125
  int beg = 0, end = qLen;
  int t_next = tLen > 0 ? uni((int)ts[0]) : 0;

  for (int i = 0; i < tLen; ++i) {
    if (i >= qLen && amax > 0) {  // nothing past this row can change the result (tail_row_bound)
      const int U = tail_row_bound(qLen, i, h0, amax, oDel, eDel);
      if (U <= mx && U < gscore) break;
    }
    const int t = t_next;
    if (i + 1 < tLen) t_next = uni((int)ts[i + 1]);  // prefetch the next row's target base
    const int h1 = max(0, h0 - (oDel + eDel * (i + 1)));  // SWUtil.scala:137-138
    beg = max(beg, i - w);                                // SWUtil.scala:140-142
    end = min(min(end, i + w + 1), qLen);
    const int8_t* __restrict__ q = qp + t * qLen;

    int carry = NEG;   // running max of g(k) = a(k) - oeIns + k*eIns over the columns already swept
    int hleft = h1;    // H(i, j0-1); for the first chunk the "first column" value of SWUtil.scala:137
    int m = 0, mj = -1;
    int lz_all = -1;   // last column with H == 0 among the columns already swept
    int lz_best = -1;  // last column < mj with H == 0
    int fz_after = -1; // first column > mj with H == 0

    for (int j0 = beg; j0 < end; j0 += 64) {
      const int j = j0 + lane;
      const bool act = j < end;
      int2 he = make_int2(0, 0);
      int s = 0;
      if (act) {
        he = eh[j];
        s = q[j];
      }
      const int a = act ? max(he.x + s, he.y) : NEG;
      const int jE = j * eIns - oeIns;
      const int P = max(wave_scan_max(a + jE), carry);
      const int Pex = wave_shr1(carry, P);  // exclusive prefix; lane 0 takes the carry
      const int H = max3i(a, Pex - (jE + oeIns - eIns), 0);     // F(i,j) = max(0, Pex - (j-1)*eIns)
      carry = __builtin_amdgcn_readlane(P, 63);
      const int E = max3i(he.y - eDel, H - oeDel, 0);           // E(i+1,j)
      const int Hprev = wave_shr1(hleft, H);  // H(i,j-1), stored at eh[j].h
      if (act) eh[j] = make_int2(Hprev, E);

      const int nact = min(64, end - j0);
      hleft = __builtin_amdgcn_readlane(H, nact - 1);
      const unsigned long long actmask = nact == 64 ? ~0ull : ((1ull << nact) - 1ull);
      const unsigned long long zmask = __builtin_amdgcn_ballot_w64(H == 0) & actmask;
      const int Hm = act ? H : -1;
      const int cm = __builtin_amdgcn_readlane(wave_scan_max(Hm), 63);
      if (cm >= m) {  // "m <= h": a later chunk with an equal maximum takes over (last arg-max)
        const unsigned long long eq = __builtin_amdgcn_ballot_w64(Hm == cm);
        const int b = 63 - __builtin_clzll(eq);
        m = cm;
        mj = j0 + b;
        const unsigned long long below = zmask & ((1ull << b) - 1ull);
        lz_best = below ? j0 + 63 - __builtin_clzll(below) : lz_all;
        const unsigned long long above = b == 63 ? 0ull : (zmask >> (b + 1));
        fz_after = above ? j0 + b + 1 + __builtin_ctzll(above) : -1;
      } else if (fz_after < 0 && zmask) {
        fz_after = j0 + __builtin_ctzll(zmask);
      }
      if (zmask) lz_all = j0 + 63 - __builtin_clzll(zmask);
    }
    if (lane == 0) eh[end] = make_int2(hleft, 0);  // SWUtil.scala:174-175
    __builtin_amdgcn_wave_barrier();

    const int jfin = beg < end ? end : beg;  // value of j after the column loop
    if (jfin == qLen && gscore <= hleft) {   // SWUtil.scala:177-182
      max_ie = i;
      gscore = hleft;
    }
    if (m == 0) break;  // SWUtil.scala:184-185
    if (m > mx) {       // SWUtil.scala:187-193
      mx = m;
      max_i = i;
      max_j = mj;
      max_off = max(max_off, abs(mj - i));
    } else if (zdrop > 0) {  // SWUtil.scala:194-199 (Scala parse) / native/ksw.c:455-461 (BWA parse)
      const bool A = (i - max_i) > (mj - max_j);
      const bool B = mx - m - ((i - max_i) - (mj - max_j)) * eDel > zdrop;
      const bool C = mx - m - ((mj - max_j) - (i - max_i)) * eIns > zdrop;
      const bool stop = zmode == BPSW_ZDROP_SCALA ? (A && (B || C)) : (A ? B : C);
      if (stop) break;
    }
    // SWUtil.scala:202-214 on V(p) = eh[p].h: V(beg) = h1, V(j+1) = H(i,j)
    beg = lz_best >= 0 ? lz_best + 2 : (h1 == 0 ? beg + 1 : beg);
    end = fz_after >= 0 ? fz_after + 1 : end + 1;
  }
  ExtRes r;
  r.max = mx; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
  return r;
}

// ---------------------------------------------------------------------------------------------------
// Register-resident form of the same SWExtend for qLen <= 255 (every 2x150 / 2x250 bp task).
// Column j lives in lane j&63 of slot j>>6, so the (H,E) row never leaves VGPRs and the diagonal
// H(i-1,j-1) is one DPP wave_shr:1.  Two interleaved fused-DPP max-scans per slot give
//   prefix max of g(k) = a(k) - oeIns + k*eIns   -> F(i,j)
//   wave max of a(k)                             -> the row maximum m
// With oeIns > 0, H(i,j) == m > 0 iff a(j) == m (F stays strictly below the maximum), so the row
// maximum and its LAST arg-max (SWUtil.scala:158-161) are read off `a` with one ballot.
// The kernel is bound by the CU's single scalar unit, not by VALU, so the per-row control below is
// written to need as few SALU instructions as possible (s_bfm/s_flbit/s_ff1 on the 64-bit zero mask).
// ---------------------------------------------------------------------------------------------------

// Two independent inclusive max-scans over the 64 lanes, interleaved so each DPP read sees its operand
// two wait states after the write (the hazard hipcc does not handle inside asm statements).
__device__ __forceinline__ void dual_scan_max(int& g, int& a) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(g), "+v"(a));
}
__device__ __forceinline__ unsigned long long s_below_mask(int width) {  // (1 << width) - 1, width 0..63
  unsigned long long r;
  asm("s_bfm_b64 %0, %1, 0" : "=s"(r) : "s"(width));
  return r;
}
__device__ __forceinline__ int s_lead_zeros(unsigned long long v) {  // -1 when v == 0
  int r;
  asm("s_flbit_i32_b64 %0, %1" : "=s"(r) : "s"(v));
  return r;
}
__device__ __forceinline__ int s_first_one(unsigned long long v) {  // -1 when v == 0
  int r;
  asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(v));
  return r;
}

// Where the wave-uniform row control runs.  Measured on MI355X (tools/microbench_issue.hip, profiles/): a SIMD issues
// one integer VALU / DPP / v_cmp / v_readlane wave-instruction per ~3.7 cycles and one SALU instruction per ~3.7
// cycles; mixed streams from several waves reach about one instruction per 2.4 cycles.  Two builds:
//   BPSW_EXT_VECTOR_CONTROL 1: uniform values kept in VGPRs through an opaque asm (fewer SALU, more VALU instructions)
//   BPSW_EXT_VECTOR_CONTROL 0: control on the scalar pipe
// Early in the round the kernel issued 355 M VALU + 200 M SALU (build 1) or 239 M + 323 M (build 0) per 30 k-task batch and
// both took the same time: the sum was what counted.  After the closed forms and the row-loop work the kernel is at
// 95 M VALU + 39 M SALU (build 1), the step is bound by its VALU instructions, and build 0 is faster (8 batches in flight
// 172 -> 176.5 M reads/s, bench step 116.7 -> 118.6): it is the default again.
#ifndef BPSW_EXT_VECTOR_CONTROL
#define BPSW_EXT_VECTOR_CONTROL 0
#endif
__device__ __forceinline__ int vu(int s) {
#if BPSW_EXT_VECTOR_CONTROL
  int v;
  asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
#else
  return s;
#endif
}
__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
constexpr int NEG_A = -(1 << 20);  // "no cell" for the register sweeps: below every score, and NEG_A << 7 does not overflow

// The z-drop test of a row that did not improve the maximum, with k = (i - max_i) - (mj - max_j) and X = max - m >= 0:
//   A = k > 0,  B = X - k*eDel > zdrop,  C = X + k*eIns > zdrop.
// Scala parse (SWUtil.scala:194-199): A && (B || C); for k > 0 and eDel, eIns >= 1 (validated) C >= B, so it is A && C.
// BWA parse (native/ksw.c:455-461): A ? B : C.
__device__ __forceinline__ bool zdrop_stop(const int k, const int X, const int eDel, const int eIns, const int zdrop, const int zmode) {
  const int with_gap = zmode == BPSW_ZDROP_SCALA ? X + k * eIns : X - k * eDel;  // k > 0
  const int other = X + k * eIns;                                                // k <= 0: the BWA parse only
  return k > 0 ? with_gap > zdrop : (zmode != BPSW_ZDROP_SCALA && other > zdrop);
}

// The wave-uniform state of a call in flight, handed from the slot sweep (sw_extend_reg) to the sliding sweep
// (sw_extend_leanS) together with the (H,E) row in LDS: see sw_extend_reg_any.
struct ExtCarry {
  int handed;  // 1: the slot sweep stopped before row `row` and left the state here and in eh[]
  int mx, max_i, max_j, max_ie, gscore, max_off, beg, end, h1raw, row;
};

// QC: query source, qcode(j) = base code (0..4) of column j < qLen
// eh / carry (optional): LDS row of qLen + 2 (H,E) pairs and the hand-over record.  When given, the sweep stops at the first
// row whose band [beg, end] fits the 128-column window of the sliding sweep and leaves the call's state there.
template <int S, class QC>
__device__ ExtRes sw_extend_reg(const int lane, const int qLen, const int tLen, const QC& qcode,
                                const uint8_t* __restrict__ ts, const MatRows& mat, const int oDel,
                                const int eDel, const int oIns, const int eIns, const int w, const int zdrop,
                                const int zmode, const int h0, const int amax, int2* __restrict__ eh = nullptr,
                                ExtCarry* __restrict__ carry = nullptr) {
  const int oeDel = oDel + eDel, oeIns = oIns + eIns;
  int Hs[S], Es[S], As[S], plo[S], phi[S], jE[S], c2[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int j = 64 * s + lane;
    const int code = j < qLen ? qcode(j) : 4;
    const int sh = 8 * code;
    plo[s] = (int)(((mat.row[0] >> sh) & 0xff) | (((mat.row[1] >> sh) & 0xff) << 8) | (((mat.row[2] >> sh) & 0xff) << 16) |
                   (((mat.row[3] >> sh) & 0xff) << 24));
    phi[s] = (int)(int8_t)((mat.row[4] >> sh) & 0xff);
    Hs[s] = j == 0 ? h0 : max(0, h0 - oeIns - (j - 1) * eIns);  // row -1, SWUtil.scala:97-104
    Es[s] = 0;
    As[s] = NEG;
    jE[s] = j * eIns - oeIns;
    c2[s] = (j - 1) * eIns;
  }
  // SWUtil.scala:118-125 -- wave-uniform state, held in VGPRs (see vu)
  int mx = vu(h0), max_i = vu(-1), max_j = vu(-1), max_ie = vu(-1), gscore = vu(-1), max_off = vu(0);
  int beg = vu(0), end = vu(qLen);
  int h1raw = vu(h0 - oDel);  // h0 - (oDel + eDel*(i+1)) after the decrement below
  int iv = vu(0);             // vector copy of the row index

  const int i_tail = amax > 0 ? qLen : 0x7fffffff;  // first row the tail bound applies to (one scalar compare per row)
  for (int i = 0; i < tLen; ++i, iv += 1) {
    if (i >= i_tail) {  // nothing past this row can change the result (tail_row_bound)
      const int U = tail_row_bound(qLen, i, h0, amax, oDel, eDel);
      if (any_lane(U <= mx && U < gscore)) break;
    }
    if (S > 2 && carry) {  // this row's band fits the sliding sweep's window: hand the call over (sw_extend_reg_any)
      const int nb = max(beg, iv - w), ne = min(min(end, iv + (w + 1)), qLen);
      if (any_lane(ne - (nb & ~1) <= 127)) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const int j = 64 * s + lane;
          if (j <= qLen) eh[j] = make_int2(Hs[s], Es[s]);
        }
        __builtin_amdgcn_wave_barrier();
        carry->handed = 1;
        carry->mx = __builtin_amdgcn_readfirstlane(mx); carry->max_i = __builtin_amdgcn_readfirstlane(max_i);
        carry->max_j = __builtin_amdgcn_readfirstlane(max_j); carry->max_ie = __builtin_amdgcn_readfirstlane(max_ie);
        carry->gscore = __builtin_amdgcn_readfirstlane(gscore); carry->max_off = __builtin_amdgcn_readfirstlane(max_off);
        carry->beg = __builtin_amdgcn_readfirstlane(beg); carry->end = __builtin_amdgcn_readfirstlane(end);
        carry->h1raw = __builtin_amdgcn_readfirstlane(h1raw); carry->row = i;
        return ExtRes{0, 0, 0, 0, 0, 0};
      }
    }
    const int tsv = __builtin_amdgcn_readfirstlane((int)ts[i]);  // 8 * target base, same in every lane
    const bool isN = tsv == 32;  // wave-uniform
    h1raw -= eDel;
    const int h1 = max(0, h1raw);      // SWUtil.scala:137-138
    beg = max(beg, iv - w);            // SWUtil.scala:140-142
    end = min(min(end, iv + (w + 1)), qLen);
    const int span = end - beg;
    const unsigned spanA = (unsigned)max(span, 0);      // columns beg <= j <  end
    const unsigned spanU = (unsigned)max(span + 1, 0);  // columns beg <= j <= end (eh[end] is written too)

    int carry_g = NEG, carry_a = NEG;  // running maxima over the slots already swept (scalars, S > 1)
    int hl_prev = h1;                  // H(i, 64*s - 1) for the next slot's lane 0
    int scan_a = NEG;
    unsigned long long zm[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      zm[s] = 0;
      const unsigned rel = (unsigned)(64 * s + lane - beg);
      const bool upd = rel < spanU;
      if (S > 1 && !any_lane(upd)) {  // slot entirely outside [beg, end]
        As[s] = NEG;
        continue;
      }
      const bool act = rel < spanA;
      int sc;
      if (isN) sc = phi[s];  // a scalar branch (N rows are rare), not a select on every row
      else sc = __builtin_amdgcn_sbfe(plo[s], (unsigned)tsv, 8u);
      const int araw = max(Hs[s] + sc, Es[s]);  // >= 0: E never goes below 0
      const int a = act ? araw : NEG_A;
      As[s] = a;
      int Pg = a + jE[s];
      // one slot: the row maximum and its LAST column come out of the same scan (NEG_A << 6 stays far below 0)
      scan_a = S == 1 ? ((a << 6) | lane) : a;
      dual_scan_max(Pg, scan_a);
      if (S > 1) {
        Pg = max(Pg, carry_g);
        scan_a = max(scan_a, carry_a);
      }
      const int Pex = wave_shr1(carry_g, Pg);  // exclusive prefix; lane 0 takes the carry of the earlier slots
      if (S > 1) {
        carry_g = __builtin_amdgcn_readlane(Pg, 63);
        carry_a = __builtin_amdgcn_readlane(scan_a, 63);
      }
      const int H = max3i(a, Pex - c2[s], 0);  // F(i,j) = max(0, Pex - (j-1)*eIns)
      zm[s] = __builtin_amdgcn_ballot_w64(H < 1) & __builtin_amdgcn_ballot_w64(act);  // H >= 0: the zero cells of the band
      const int En = act ? max3i(Es[s] - eDel, H - oeDel, 0) : 0;  // E(i+1,j); eh[end].e = 0
      int hsh = wave_shr1(hl_prev, H);                             // H(i,j-1)
      if (S > 1) hl_prev = __builtin_amdgcn_readlane(H, 63);
      hsh = rel == 0u ? h1 : hsh;                                  // eh[beg].h = h1, SWUtil.scala:153
      if (S == 1) {  // one slot: every lane is written (columns outside the band are never read before the band rewrites them, see sw_extend_lean2)
        Hs[s] = hsh;
        Es[s] = En;
      } else {
        Hs[s] = upd ? hsh : Hs[s];
        Es[s] = upd ? En : Es[s];
      }
    }
    const int mkey = max(0, S > 1 ? carry_a : __builtin_amdgcn_readlane(scan_a, 63));  // scalar
    const int m = S == 1 ? mkey >> 6 : mkey;

    // SWUtil.scala:177-182: j after the column loop is end (or beg for an empty band); h1 there is eh[end].h
    if (any_lane((span > 0 ? end : beg) == qLen)) {
      int hlast = h1;
      if (any_lane(span > 0)) {
        const int e = __builtin_amdgcn_readfirstlane(end);
#pragma unroll
        for (int s = 0; s < S; ++s)
          if (S == 1 || (e >> 6) == s) hlast = __builtin_amdgcn_readlane(Hs[s], e & 63);
      }
      const bool better = gscore <= hlast;
      max_ie = better ? iv : max_ie;
      gscore = better ? hlast : gscore;
    }
    if (m == 0) break;  // SWUtil.scala:184-185

    int sm = 0, bm;  // slot and lane of the LAST column whose a == m  (SWUtil.scala:158-161)
    if (S == 1) {
      bm = mkey & 63;
    } else {
      bm = -1;
#pragma unroll
      for (int s = S - 1; s >= 0; --s) {
        const int lzc = s_lead_zeros(__builtin_amdgcn_ballot_w64(As[s] == m));
        if (bm < 0 && lzc >= 0) { bm = 63 - lzc; sm = s; }
      }
    }
    const int mj = 64 * sm + bm;  // scalar
    const bool improved = m > mx;
    if (!any_lane(improved) && zdrop > 0) {  // SWUtil.scala:194-199 (Scala parse) / native/ksw.c:455-461 (BWA parse)
      if (any_lane(zdrop_stop((iv - max_i) - (mj - max_j), mx - m, eDel, eIns, zdrop, zmode))) break;
    }
    {  // SWUtil.scala:187-193
      const int d = mj - iv;
      const int off = max3i(max_off, d, -d);
      mx = improved ? m : mx;
      max_i = improved ? iv : max_i;
      max_j = improved ? mj : max_j;
      max_off = improved ? off : max_off;
    }
    // band trimming, SWUtil.scala:202-214: last zero of H left of mj, first zero right of mj
    int lzc, fo, lbase = 65, fbase = bm + 2;
    if (S == 1) {
      if (zm[0] == 0ull) {  // no zero in the band at all -- most rows while the score is high
        lzc = -1;
        fo = -1;
      } else {
        lzc = s_lead_zeros(zm[0] & s_below_mask(bm));
        fo = s_first_one((zm[0] >> bm) >> 1);
      }
    } else {
      lzc = -1;
      fo = -1;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const unsigned long long below = s == sm ? (zm[s] & s_below_mask(bm)) : zm[s];
        const unsigned long long above = s == sm ? (zm[s] >> bm) >> 1 : zm[s];
        const int l = s_lead_zeros(below);
        if (s <= sm && l >= 0) { lzc = l; lbase = 64 * s + 65; }
        const int f = s_first_one(above);
        if (s >= sm && fo < 0 && f >= 0) { fo = f; fbase = s == sm ? 64 * s + bm + 2 : 64 * s + 1; }
      }
    }
    const int nb0 = beg + (h1 == 0 ? 1 : 0);
    beg = lzc >= 0 ? vu(lbase - lzc) : nb0;
    end = fo >= 0 ? vu(fbase + fo) : end + 1;
  }
  ExtRes r;
  r.max = __builtin_amdgcn_readfirstlane(mx);
  r.qle = __builtin_amdgcn_readfirstlane(max_j) + 1;
  r.tle = __builtin_amdgcn_readfirstlane(max_i) + 1;
  r.gtle = __builtin_amdgcn_readfirstlane(max_ie) + 1;
  r.gscore = __builtin_amdgcn_readfirstlane(gscore);
  r.max_off = __builtin_amdgcn_readfirstlane(max_off);
  return r;
}


// ---------------------------------------------------------------------------------------------------
// Closed form for near-exact flanks.  Let a = the match score (every other matrix entry < a), s_j = S(t_j, q_j) the score
// on the main diagonal, D_i = sum_{j<=i} (a - s_j) the diagonal deficit, D = D_{qLen-1}.  If
//        D < min(oeIns, oeDel),   h0 > D,   tLen >= qLen        (default scoring: at most ONE substitution, or <= 3 N)
// then SWExtend's result is known without running the DP:
//   * the gapless path gives H(i,i) >= h0 + a(i+1) - D_i; any path that leaves the main diagonal pays a gap open (the
//     first column and row -1 are themselves gap-penalised, SWUtil.scala:97-104,137), a path restarted from a zero cell
//     starts at 0 < h0 - D, and no path collects more than min(i,j)+1 diagonal steps of at most +a each; hence every
//     off-diagonal cell of row i is <= h0 + a(min(i,j)+1) - min(oeIns,oeDel) < H(i,i), and H(i,i) equals the gapless value:
//     the row maximum is m(i) = h0 + a(i+1) - D_i at mj = i for every i < qLen (never 0);
//   * the trimming keeps column mj+1 (SWUtil.scala:202-214: beg <= mj+1, end >= mj+2) and the band keeps the diagonal
//     (w >= 1), so the next diagonal cell is always computed; rows with m <= max reach the z-drop test with
//     i - max_i == mj - max_j: the Scala parse does nothing, the BWA parse compares max - m <= D with zdrop (so D <= zdrop
//     is required when zdrop > 0);
//   * max / max_i follow from folding m(i) with the strict `m > max` (m rises by a between deficit columns, so the only
//     candidates are the rows just before a deficit column and the last row); max_off stays 0;
//   * at row qLen-1 the loop ends at j == qLen: gscore = H(qLen-1,qLen-1) = h0 + a*qLen - D, max_ie = qLen-1; every later
//     row has H(i,qLen-1) <= h0 + a*qLen - oeDel < gscore, so nothing after row qLen-1 (z-drop, m == 0, tLen) matters.
// The caller's band retry stops after its first try (max_off = 0 < 3/4 aw needs aw >= 2).  Seeds are maximal exact
// matches, so a flank always STARTS with an error; at 1 % substitutions about half of all flanks have no second one.
// Verified against the oracle on 45 894 random flanks (homopolymers, tandem repeats, N, five gap-cost sets, w 2..200,
// z-drop 0/3/5/100, both parses: 0 differences) and by every parity test, whose batches take this path for ~half the sides.
// ExtScoring::exact_a / ChainParams::exact_a carry a (0: matrix not of that form, or BPSW_EXT_EXACT=0).
__device__ __forceinline__ int mat_score(const MatRows& mat, const int t, const int q) {
  return (int)(int8_t)((mat.row[t] >> (8 * q)) & 0xff);
}

// Single-gap certificate for oe_min <= D < 2*oe_min (default scoring: two substitutions, or one plus N).  With D below TWO
// gap opens every path with two or more gaps stays below the gapless diagonal of its row (and below the final gscore / max
// for rows past the query), and so does every path restarted from a zero cell (h0 > D); what is left are the paths that
// follow the main diagonal to (r-1,r-1), open ONE gap of length d and then run along the shifted diagonal.  With
//     A(x) = sum_{y<=x} S(t_y,q_y),   G_d(x) = sum_{y<=x} [S(t_y,q_{y+d}) - S(t_y,q_y)],   B_d(x) = sum_{y<=x} S(t_{y+d},q_y)
// the gapless diagonal is the unique row maximum of every row, and gscore / max are final after row qLen-1, iff for every d
//   insertion:  G_d(x) - min_{z<x} G_d(z) <  oIns + d*eIns            for all x <= qLen-1-d          (z from -1, G_d(-1) = 0)
//               G_d(xl) - min_{z<=xl} G_d(z) - [A(qLen-1) - A(xl)] <= oIns + d*eIns,  xl = qLen-1-d   (the shifted path ends in
//               the LAST column at row xl: it must not beat the final gscore; a tie is fine, `gscore <= h1` lets the later row win)
//   deletion:   [B_d(x) - A(min(x+d,qLen-1))] - min_{z<=x} [B_d(z) - A(z)] < oDel + d*eDel   for all x <= qLen-1, x+d < tLen
//               (for x+d >= qLen the cell lies below the query end and is compared with the final gscore <= max).
// A gain can never exceed the main-diagonal deficit it avoids, so only d <= (D - o)/e need checking.  Each condition is a
// prefix sum and a running minimum: two DPP scans per shift and 64 columns.  Checked against the oracle's full DP on 116 000
// adversarial flanks (repeats, clustered defects, eight gap-cost sets, both parses) and on the bench batch (no difference; the
// certificate passes for 30 % of the DP cost that the deficit rule alone leaves).
template <class QC, class TC>
__device__ bool single_gap_certificate(const int lane, const int qLen, const int tLen, const QC& qcode, const TC& tcode,
                                       const MatRows& mat, const int D, const int oDel, const int eDel, const int oIns,
                                       const int eIns) {
  const int nC = (qLen + 63) >> 6;  // 1 or 2 (the caller checks qLen <= 128)
  int sm[2] = {0, 0}, A[2] = {0, 0}, qv[2] = {4, 4}, tv[2] = {4, 4};
  int carry = 0;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c < nC) {
      const int y = 64 * c + lane;
      const bool valid = y < qLen;
      const int q = valid ? qcode(y) : 4, t = valid ? tcode(y) : 4;
      qv[c] = q; tv[c] = t;
      sm[c] = valid ? mat_score(mat, t, q) : 0;
      A[c] = wave_scan_add(sm[c]) + carry;
      carry = __builtin_amdgcn_readlane(A[c], 63);
    }
  }
  const int Atot = carry;  // A(qLen-1)
  auto A_at = [&](int z) {  // A at the (wave-uniform or per-lane) index z < qLen
    const int a0 = __builtin_amdgcn_ds_bpermute((z & 63) << 2, A[0]);
    const int a1 = __builtin_amdgcn_ds_bpermute((z & 63) << 2, A[1]);
    return (z >> 6) ? a1 : a0;
  };
  unsigned long long viol = 0ull;
  // ---- one insertion of d query bases, then the diagonal shifted right by d ---------------------------------------
  const int dI = (D - oIns) / eIns;
  for (int d = 1; d <= dI && d < qLen; ++d) {
    const int xl = qLen - 1 - d, T = oIns + d * eIns;
    const int tail_main = Atot - uni(A_at(xl));
    int gcar = 0, mcar = 0;  // G_d and its running minimum at the end of the previous chunk (G_d(-1) = 0)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c < nC && 64 * c <= xl) {
        const int y = 64 * c + lane;
        const bool valid = y <= xl;
        int cv = 0;
        if (valid) cv = mat_score(mat, tv[c], qcode(y + d)) - sm[c];
        const int G = wave_scan_add(cv) + gcar;
        const int Gex = wave_shr1(gcar, G);                 // G_d(y-1)
        const int mn = min(wave_scan_min(Gex), mcar);       // min over G_d(-1 .. y-1)
        bool bad = valid && G - mn >= T;
        bad = bad || (y == xl && G - min(mn, G) - tail_main > T);
        viol |= __builtin_amdgcn_ballot_w64(bad);
        gcar = __builtin_amdgcn_readlane(G, 63);
        mcar = min(__builtin_amdgcn_readlane(mn, 63), gcar);
      }
    }
    if (viol) return false;
  }
  // ---- one deletion of d target bases, then the diagonal shifted down by d ------------------------------------------
  const int dD = (D - oDel) / eDel;
  for (int d = 1; d <= dD; ++d) {
    const int T = oDel + d * eDel;
    int bcar = 0, mcar = 0;  // B_d and the running minimum of B_d - A at the end of the previous chunk (both 0 at -1)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c < nC) {
        const int x = 64 * c + lane;
        const bool valid = x < qLen && x + d < tLen;
        int sv = 0;
        if (valid) sv = mat_score(mat, tcode(x + d), qv[c]);
        const int B = wave_scan_add(sv) + bcar;
        const int z = min(x + d, qLen - 1);
        const int val = B - A_at(z);
        const int base = valid ? B - A[c] : POS;
        const int mnb = min(wave_scan_min(base), mcar);     // min over z <= x of B_d(z) - A(z), and 0 for z = -1
        viol |= __builtin_amdgcn_ballot_w64(valid && val - mnb >= T);
        bcar = __builtin_amdgcn_readlane(B, 63);
        mcar = min(mcar, __builtin_amdgcn_readlane(mnb, 63));
      }
    }
    if (viol) return false;
  }
  return true;
}

// Two gap opens.  With the default scoring family (match a = 1, every other matrix entry <= -1 so that one imperfect diagonal
// step costs at least 2, both gap extensions 1, oIns + eIns = oDel + eDel = oe) the form also holds for D = 2*oe + s, s in {0, 1}
// (three substitutions, D = 15, are a third of the DP rows the single-gap certificate leaves on 2x150 bp reads), under two more
// tests.  A path with k diagonal steps of deficit P in total, deletions of total length Ld and insertions of total length Li
// ends in cell (i, i + Li - Ld) with score  Hd(i) + D_i - a*Ld - P - G  (Hd the gapless diagonal, G its gap cost).
//  * Three or more gaps: G >= 3*oe > D (oe >= 2).  Two deletions (rows < qLen): D - 2a - 2*oe = s - 2 < 0.  One of each, lengths
//    (1,1): s - 1 - P <= 0, and the cell is (i,i) itself, where a tie changes nothing; longer ones lose e (+a) per base.
//  * Two insertions: s - P - (Li - 2): not below 0 only for P = 0 (P >= 2 otherwise) and Li <= 2 + s, in a cell RIGHT of the
//    diagonal, where even a tie would move the last arg-max.  D_i >= 2*oe needs every deficit position behind it (the rest
//    would have to sum to <= 1), so such a path takes its step of row p_last (the last deficit) on the diagonal shifted by
//    Li', 2 <= Li' <= 2 + s -- or still has one gap only there, and then that prefix beats the diagonal by
//    D - oIns - L1 >= oe + s - 1 > 0 and the single-gap certificate refuses the flank.  Test 1: S(t[p_last], q[p_last+L]) != a,
//    or one of the two deficit rows before it (if there are that many) has no match on any diagonal shifted by 1..L: P = 0
//    needs every deficit position off the main diagonal.
//  * Two deletions below the query end: the last column at row qLen-1+Ld holds g + s - P - (Ld - 2); a value >= g would move
//    gscore / max_ie (`gscore <= h1`) or the maximum.  Its last diagonal step is (t[qLen-1+Ld], q[qLen-1]), a match.
//    Test 2: S(t[qLen-1+L], q[qLen-1]) != a for 2 <= L <= 2 + s, or one of the last three deficit positions p has no match
//    S(t[p+sft], q[p]), 1 <= sft <= L.  (Paths ending in a gap there sit oe below a single-gap
//    prefix, which the certificate bounds by g; one deletion plus one insertion reach at most g + s - 1 - oe.)
// The single-gap certificate itself is unchanged: its conditions never used D < 2*oe, only its caller did.
template <class QC, class TC>
__device__ __forceinline__ bool flank_closed_form(const int lane, const int qLen, const int tLen, const QC& qcode, const TC& tcode,
                                                  const MatRows& mat, const int h0, const int a, const int oDel, const int eDel,
                                                  const int oIns, const int eIns, const int zdrop, const int certify,
                                                  ExtRes* out) {
  const int oe_min = min(oIns + eIns, oDel + eDel);
  const bool two_opens = certify >= 2 && qLen <= 128 && a == 1 && eIns == 1 && eDel == 1 && oIns + eIns == oDel + eDel && oe_min >= 2;
  const int limit = certify && qLen <= 128 ? (two_opens ? 2 * oe_min + 2 : 2 * oe_min) : oe_min;  // deficit below which the form can still hold
  int D = 0, best = h0, best_i = -1;  // SWUtil.scala:118-121: max = h0, max_i = max_j = -1
  int p_last = -1, p_prev = -1, p_prev2 = -1;  // the last three deficit positions
  for (int j0 = 0; j0 < qLen; j0 += 64) {
    const int j = j0 + lane;
    int d = 0;
    if (j < qLen) d = a - mat_score(mat, tcode(j), qcode(j));
    unsigned long long m = __builtin_amdgcn_ballot_w64(d > 0);
    while (m) {  // the (very few) diagonal cells that are not a match
      const int b = (int)__builtin_ctzll(m);
      m &= m - 1;
      const int pos = j0 + b;
      const int v = h0 + pos * a - D;  // m(pos-1): the last row before this deficit
      if (pos >= 1 && v > best) { best = v; best_i = pos - 1; }
      D += __builtin_amdgcn_readlane(d, b);
      p_prev2 = p_prev; p_prev = p_last; p_last = pos;
      if (D >= limit) return false;
    }
  }
  if (h0 <= D || (zdrop > 0 && D > zdrop)) return false;
  if (D >= 2 * oe_min) {  // two gap opens: tests 1 and 2 above, sharpened by the two deficit positions before the last
    const auto is_match = [&](const int ti, const int qi) {
      return ti >= 0 && qi >= 0 && ti < tLen && qi < qLen && mat_score(mat, tcode(ti), qcode(qi)) == a;
    };
    for (int L = 2; L <= 2 + (D - 2 * oe_min); ++L) {
      // P = 0 also needs the steps of rows p_prev, p_prev2 on a shifted diagonal (1..L) and matching there
      if (is_match(p_last, p_last + L)) {
        bool m1 = p_prev < 0, m2 = p_prev2 < 0;  // fewer than three deficit positions: no further condition
        for (int sft = 1; sft <= L; ++sft) { m1 = m1 || is_match(p_prev, p_prev + sft); m2 = m2 || is_match(p_prev2, p_prev2 + sft); }
        if (m1 && m2) return false;
      }
      // ... and the steps that consume q[p] for the three deficit positions p on a diagonal shifted down by 1..L
      if (is_match(qLen - 1 + L, qLen - 1)) {
        bool m0 = false, m1 = p_prev < 0, m2 = p_prev2 < 0;
        for (int sft = 1; sft <= L; ++sft) {
          m0 = m0 || is_match(p_last + sft, p_last); m1 = m1 || is_match(p_prev + sft, p_prev); m2 = m2 || is_match(p_prev2 + sft, p_prev2);
        }
        if (m0 && m1 && m2) return false;
      }
    }
  }
  if (D >= oe_min && !single_gap_certificate(lane, qLen, tLen, qcode, tcode, mat, D, oDel, eDel, oIns, eIns)) return false;
  const int g = h0 + qLen * a - D;
  if (g > best) { best = g; best_i = qLen - 1; }
  out->max = best; out->qle = best_i + 1; out->tle = best_i + 1; out->gtle = qLen; out->gscore = g; out->max_off = 0;
  return true;
}

// One gap of ONE base right at the start of the flank, everything after it matching -- a third of the flanks that carry an
// indel (the seed is a maximal exact match, so when the first difference is the indel it sits at position 0).  Default scoring
// family as above (a = 1, imperfect steps cost >= 2, gap extensions 1, oIns + eIns = oDel + eDel = oe).
//  * Insertion (the read has one base more): t[i] == q[i+1] for i <= n-2.  The path "insert q[0], then the diagonal" gives
//    R(i) = h0 - oe + a(i+1) in cell (i, i+1).  Any path into cell (i, i+d) takes at most min(i, i+d) + 1 diagonal steps and
//    pays at least o + |d|e, so it scores at most R(i) - (d-1) for d >= 2 and R(i) + a - 2a|d| for d <= -1: every cell
//    off the two diagonals d = 0, 1 stays strictly below R(i); a restart from a zero cell gives at most a(i+1) < R(i) (h0 > oe);
//    cell (i,i) is the gapless M(i) = h0 + sum of the main diagonal, or something at least a below R(i).  With M(i) <= h0
//    (checked; the main diagonal starts with a mismatch and is unrelated sequence after it) no row improves on h0 until R
//    does, from then on every row improves in column i+1 up to row n-2, where the path reaches the last column:
//    max = gscore = h0 - oe + a(n-1), max_i = max_ie = n-2, max_j = n-1, max_off = 1.  Below that row only the diagonal
//    shifted DOWN by one or two could come back, with R + a in cell (n, n-1) resp. exactly R in cell (n+1, n-1) (a tie moves
//    gscore to the later row); they cannot when their gap-at-the-start paths are not perfect (any other path into those cells
//    takes the mismatch (t0,q0)): periodic sequence fails this and is left to the DP.
//  * Deletion (the reference has one base more): t[i+1] == q[i] for i <= n-1, R'(r) = h0 - oe + a r in cell (r, r-1), last cell
//    (n, n-1): max = gscore = h0 - oe + a n, max_i = max_ie = n, max_j = n-1, max_off = 1.  Cells right of the main diagonal
//    are bounded by R'(r) + a for the shift +1 and by R'(r) for +2 only: both drop by >= 2 when their gap-at-the-start path
//    begins with a mismatch ((t0,q1) resp. (t0,q2)); every other path into them takes (t0,q0).  Ties with cells LEFT of
//    the row maximum do not move the last arg-max.
//  * Trimming keeps the path: the cells next to it on the side away from the main diagonal are >= R - oe > 0 (h0 >= 2 oe),
//    so the nearest zero is at least two columns off; the band needs w >= 4 (the caller's retry stops at max_off = 1 < 3).
//    Z-drop: non-improving rows have i - max_i - (mj - max_j) in {0, -1} and max - m <= oe - a, never above zdrop >= oe.
template <class QC, class TC>
__device__ bool flank_start_gap_form(const int lane, const int n, const int tLen, const QC& qcode, const TC& tcode, const MatRows& mat,
                                     const int h0, const int a, const int oDel, const int eDel, const int oIns, const int eIns,
                                     const int zdrop, const int wBand, ExtRes* out) {
  const int oe = oIns + eIns;
  if (!(a == 1 && eIns == 1 && eDel == 1 && oe == oDel + eDel && oe >= 2)) return false;
  if (wBand < 4 || h0 < 2 * oe + 1 || (zdrop > 0 && zdrop < oe) || n < oe + 3 || n > 255) return false;
  bool ins = tLen >= n - 1, del = tLen >= n + 1;
  if (!ins && !del) return false;
  bool del2 = tLen >= n + 2;  // the diagonal shifted down by TWO perfect as well: its gap-at-the-start path ties the insertion form's gscore
  int carry = 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const bool valid = j < n, has_t = valid && j < tLen;
    const int q = valid ? qcode(j) : 4;
    const int tm = has_t ? tcode(j) : 4;
    const int sm = has_t ? mat_score(mat, tm, q) : 0;
    const int S = wave_scan_add(sm) + carry;  // M(j) - h0
    carry = __builtin_amdgcn_readlane(S, 63);
    if (any_lane(has_t && S > 0)) return false;  // the main diagonal got back above h0
    if (ins && any_lane(valid && j <= n - 2 && mat_score(mat, tm, qcode(min(j + 1, n - 1))) != a)) ins = false;
    if (del && any_lane(valid && mat_score(mat, tcode(min(j + 1, tLen - 1)), q) != a)) del = false;
    if (del2 && any_lane(valid && mat_score(mat, tcode(min(j + 2, tLen - 1)), q) != a)) del2 = false;
    if (!ins && !del) return false;
  }
  // The insertion form also needs the two diagonals below the main one imperfect: a deletion of one base at the start followed by n
  // matches gives R + a in cell (n, n-1), one of two bases gives exactly R in cell (n+1, n-1) -- and `gscore <= h1` lets the later
  // row win a tie.  (A deletion opened later takes the mismatch (t0,q0) first and is at least 2 lower.)
  if (ins && (del || del2)) return false;
  if (del) {
    // row 0 has no cell of the path yet: its maximum must be the main-diagonal cell h0 + s0, above every restart (<= a) and
    // above the cells fed from row -1 (<= h0 - oe - 1 once (t0,q1), (t0,q2) are mismatches) -- else the last arg-max of row 0
    // sits somewhere to the right and the trimming cuts the path off (found by tools/soak_cert2.py with oe = 2, h0 = 5)
    const int s0 = mat_score(mat, tcode(0), qcode(0));
    if (h0 + s0 <= a || s0 + oe + 1 <= 0) return false;
    if (mat_score(mat, tcode(0), qcode(1)) == a) return false;                // n >= 5 here
    if (mat_score(mat, tcode(0), qcode(2)) == a) return false;
    const int g = h0 - oe + a * n;
    out->max = g; out->qle = n; out->tle = n + 1; out->gtle = n + 1; out->gscore = g; out->max_off = 1;
    return true;
  }
  const int g = h0 - oe + a * (n - 1);
  out->max = g; out->qle = n; out->tle = n - 1; out->gtle = n - 1; out->gscore = g; out->max_off = 1;
  return true;
}

// query source of a wire-batch task: nibble stream `words`, first column at base qStart
struct NibbleQ {
  const uint32_t* __restrict__ words;
  int qStart;
  __device__ __forceinline__ int operator()(int j) const { return nibble_at(words, qStart + j); }
};

// target sources of one side: the nibble stream of a wire batch, or the device-resident 2-bit reference (coordinate batches)
struct NibbleT {
  const uint32_t* __restrict__ words;
  int rStart;
  __device__ __forceinline__ int operator()(int i) const { return nibble_at(words, rStart + i); }
};
struct PacT {  // bnsGetSeq, util/BNTSeqUtil.scala:56-73: positions >= l_pac are the reverse strand, complemented
  const uint8_t* __restrict__ pac;
  long long l_pac, pos;
  int step;  // -1: the left flank walks backwards from the seed (MemChainToAlignBatched.scala:511-517)
  __device__ __forceinline__ int operator()(int i) const {
    const long long p = pos + (long long)step * i;
    const bool rev = p >= l_pac;
    const long long k = rev ? (l_pac << 1) - 1 - p : p;
    const int b = (pac[k >> 2] >> ((~k & 3) << 1)) & 3;
    return rev ? 3 - b : b;
  }
};
struct LdsShiftT {  // a target already staged as 8*code bytes
  const uint8_t* __restrict__ ts;
  __device__ __forceinline__ int operator()(int i) const { return (int)(ts[i] >> 3); }
};

// ---- the two-columns-per-lane sweeps ----------------------------------------------------------------------------------------
// 64 <= qLen <= 127 with the columns INTERLEAVED: lane l holds columns 2l and 2l+1.  The lane folds its two columns first, so a row
// needs ONE dual scan (the exclusive prefix of the odd column is max(prefix of the lane, the even column's term)), H(i,j-1) of the
// odd column is the even column of the same lane, and only the even column's comes from lane l-1.  The control (band, last arg-max,
// trimming, z-drop) works on bit masks of the even and the odd columns.  Every lane's (H,E) is WRITTEN every row, inside the band or
// not: a column outside the band is never read before the band has rewritten it (SWUtil.scala:174-175 writes eh[end] before the band
// can grow over it) -- which is also why a column that enters a sliding window needs no initial value.
// The sliding form (sw_extend_leanS): the same sweep over a WINDOW of 128 columns that follows the band, for flanks of any length.
// The band of a row is 44 columns wide on average and at most 127 for 99.7 % of the rows of 2x250 bp reads at 8 % / 2 % error, however
// long the flank.  Lane l holds columns base + 2l and base + 2l + 1; the per-lane constants of the prefix scan depend only on the
// column's position in the window (F(i,j) = max_k (a(k) + k e - oe) - (j-1) e is invariant under a shift of the origin); when the
// band's right end leaves the window, the window moves up to the band's left end: the (H,E) state shifts down by (new base - base) / 2
// lanes (ds_bpermute), the profile of the new columns is reloaded.  A row whose band does not fit 128 columns ends the sweep with
// *overflow = 1: the caller runs the call on the slot sweep.  `in`: continue a call the slot sweep started, state in eh[] and *in.
// ---- written for their instruction count ---------------------------------------------
// A SIMD issues about one instruction per 2.6 cycles whatever pipe it goes to: a row costs what its vector AND
// scalar AND branch instructions add up to.  sw_extend_il2 spends 69 scalar instructions and 15 branches per row on a control
// flow the compiler derives from nested breaks and from conditions it cannot prove uniform (boolean phis materialised as
// s_cselect_b64 / s_and_b64 exec pairs, v_cmp + s_cmp for a compare of two scalars).  Here every loop-carried scalar is pinned to a
// scalar register (see smax2 / smin2 below for what was dragging them onto the vector pipe), every
// break sits at the top level of the loop body, and the rare parts (N rows, the rows past the query end, z-drop) are out of line.
// (Rounds 1-2 had an earlier form of this sweep, sw_extend_il2, built in by -DBPSW_EXT_LEAN=0 as the reference these were compared with;
// it was removed in round 5: rows_cpp in bpsw_extend_rows.h is the C++ reference of the assembly rows, the oracle is everybody's.)
// min / max of two SCALARS whose result stays scalar: the DAG combiner folds max(max(a, b), c) into a three-operand node that exists
// only as a vector instruction (v_max3_i32), and the vector result then drags every user -- the band ends, the whole row control --
// onto the vector pipe.  The readfirstlane hides the inner result from that combine and folds away when its operand is scalar.
__device__ __forceinline__ int smax2(int a, int b) { return __builtin_amdgcn_readfirstlane(max(a, b)); }
__device__ __forceinline__ int smin2(int a, int b) { return __builtin_amdgcn_readfirstlane(min(a, b)); }
template <class QC>
__device__ ExtRes sw_extend_lean2(const int lane, const int qLen, const int tLen, const QC& qcode,
                                  const uint8_t* __restrict__ ts, const MatRows& mat, const int oDel,
                                  const int eDel, const int oIns, const int eIns, const int w, const int zdrop,
                                  const int zmode, const int h0, const int amax) {
  const int oeDel = oDel + eDel, oeIns = oIns + eIns;
  int Hs[2], Es[2], plo[2], phi2 = 0;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = 2 * lane + s;
    const int code = j < qLen ? qcode(j) : 4;
    const int sh = 8 * code;
    plo[s] = (int)(((mat.row[0] >> sh) & 0xff) | (((mat.row[1] >> sh) & 0xff) << 8) | (((mat.row[2] >> sh) & 0xff) << 16) |
                   (((mat.row[3] >> sh) & 0xff) << 24));
    phi2 |= (int)((mat.row[4] >> sh) & 0xff) << (8 * s);
    Hs[s] = j == 0 ? h0 : max(0, h0 - oeIns - (j - 1) * eIns);  // row -1, SWUtil.scala:97-104
    Es[s] = 0;
  }
  const int jE0 = 2 * lane * eIns - oeIns;  // j*eIns - oeIns of the even column; the odd one adds eIns
  const int kC = oeIns - eIns;              // (j-1)*eIns = (j*eIns - oeIns) + kC
  const int col0 = 2 * lane, col1 = 2 * lane + 1;
  int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
  int beg = 0, end = qLen, h1raw = h0 - oDel;
  const int i_tail = amax > 0 ? qLen : 0x7fffffff;
  const int w1 = w + 1;
  for (int i = 0; i < tLen; ++i) {
    if (i >= i_tail) {  // nothing past this row can change the result (tail_row_bound)
      const int U = tail_row_bound(qLen, i, h0, amax, oDel, eDel);
      const int stop = (U <= mx ? 1 : 0) & (U < gscore ? 1 : 0);
      if (stop) break;
    }
    const int tsv = __builtin_amdgcn_readfirstlane((int)ts[i]);  // 8 * target base
    h1raw -= eDel;
    const int h1 = smax2(0, h1raw);    // SWUtil.scala:137-138
    beg = smax2(beg, i - w);           // SWUtil.scala:140-142
    end = smin2(smin2(end, i + w1), qLen);
    const int span = end - beg;
    const unsigned spanA = (unsigned)smax2(span, 0);
    int scv[2];
    if (__builtin_expect(tsv == 32, 0)) {  // an N row
      scv[0] = __builtin_amdgcn_sbfe(phi2, 0u, 8u);
      scv[1] = __builtin_amdgcn_sbfe(phi2, 8u, 8u);
      asm volatile("" : "+v"(scv[0]), "+v"(scv[1]));  // keeps the branch: two selects per row otherwise
    } else {
      scv[0] = __builtin_amdgcn_sbfe(plo[0], (unsigned)tsv, 8u);
      scv[1] = __builtin_amdgcn_sbfe(plo[1], (unsigned)tsv, 8u);
    }
    const unsigned rel0 = (unsigned)(col0 - beg), rel1 = (unsigned)(col1 - beg);
    const bool act0 = rel0 < spanA, act1 = rel1 < spanA;
    const int a0 = act0 ? max(Hs[0] + scv[0], Es[0]) : NEG_A;
    const int a1 = act1 ? max(Hs[1] + scv[1], Es[1]) : NEG_A;
    const int Pg0 = a0 + jE0, Pg1 = a1 + jE0 + eIns;
    int Pl = max(Pg0, Pg1);
    int scan_a = max((a0 << 7) | col0, (a1 << 7) | col1);  // the row maximum and its LAST column in one scan
    dual_scan_max(Pl, scan_a);
    const int Pprev = wave_shr1(NEG, Pl);
    const int H0 = max3i(a0, Pprev - kC - jE0, 0);
    const int H1 = max3i(a1, max(Pprev, Pg0) - kC - jE0 - eIns, 0);
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(act0), m1 = __builtin_amdgcn_ballot_w64(act1);
    const unsigned long long z0 = __builtin_amdgcn_ballot_w64(H0 < 1) & m0, z1 = __builtin_amdgcn_ballot_w64(H1 < 1) & m1;
    const int En0 = act0 ? max3i(Es[0] - eDel, H0 - oeDel, 0) : 0;
    const int En1 = act1 ? max3i(Es[1] - eDel, H1 - oeDel, 0) : 0;
    const int hs0 = wave_shr1(h1, H1), hs1 = H0;  // H(i,j-1)
    Hs[0] = rel0 == 0u ? h1 : hs0;               // eh[beg].h = h1, SWUtil.scala:153 (written in every lane: see sw_extend_lean2)
    Hs[1] = rel1 == 0u ? h1 : hs1;
    Es[0] = En0;
    Es[1] = En1;
    const int mkey = smax2(0, __builtin_amdgcn_readlane(scan_a, 63));
    const int m = mkey >> 7, mj = mkey & 127;

    // SWUtil.scala:177-182: j after the column loop is end (or beg for an empty band); h1 there is eh[end].h
    const int jlast = span > 0 ? end : beg;
    if (jlast == qLen) {
      int hlast = h1;
      if (span > 0) {
        const int he = __builtin_amdgcn_readlane(Hs[0], end >> 1), ho = __builtin_amdgcn_readlane(Hs[1], end >> 1);
        hlast = (end & 1) ? ho : he;
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
    // band trimming, SWUtil.scala:202-214: last zero of H left of mj, first zero right of mj
    const int nb0 = beg + (h1 == 0 ? 1 : 0);
    if ((z0 | z1) == 0ull) {  // no zero in the band at all
      beg = nb0;
      end = end + 1;
    } else {
      const int ze_l = s_lead_zeros(z0 & s_below_mask((mj + 1) >> 1));  // even columns 2l < mj
      const int zo_l = s_lead_zeros(z1 & s_below_mask(mj >> 1));        // odd columns 2l+1 < mj
      const int cl = smax2(ze_l >= 0 ? 2 * (63 - ze_l) : -1, zo_l >= 0 ? 2 * (63 - zo_l) + 1 : -1);
      const int se = (mj + 2) >> 1, so = (mj + 1) >> 1;                 // first even / odd lane with a column > mj
      const int fe = s_first_one((z0 >> ((mj + 1) >> 1)) >> ((mj + 1) & 1));
      const int fo = s_first_one(z1 >> so);
      const int cr = smin2(fe >= 0 ? 2 * (se + fe) : 1 << 20, fo >= 0 ? 2 * (so + fo) + 1 : 1 << 20);
      beg = cl >= 0 ? cl + 2 : nb0;
      end = cr < (1 << 20) ? cr + 1 : end + 1;
    }
  }
  ExtRes r;
  r.max = mx; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
  return r;
}

// The sliding 128-column window (sw_extend_leanS: flanks of any length, continuation of a call the slot sweep started) in the
// same style.  `in` / `eh` / `overflow` as there.
template <class QC>
__device__ ExtRes sw_extend_leanS(const int lane, const int qLen, const int tLen, const QC& qcode,
                                  const uint8_t* __restrict__ ts, const MatRows& mat, const int oDel,
                                  const int eDel, const int oIns, const int eIns, const int w, const int zdrop,
                                  const int zmode, const int h0, const int amax, const int2* __restrict__ eh,
                                  const ExtCarry* __restrict__ in, int* __restrict__ overflow) {
  const int oeDel = oDel + eDel, oeIns = oIns + eIns;
  const auto cin = [&](int ExtCarry::*f, const int fresh) { return in ? __builtin_amdgcn_readfirstlane(in->*f) : fresh; };
  int Hs[2], Es[2], plo[2], phi2 = 0;
  int base = in ? (smax2(cin(&ExtCarry::beg, 0), cin(&ExtCarry::row, 0) - w) & ~1) : 0;  // first column of the window (even)
  const auto load_profile = [&]() {
    phi2 = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int j = base + 2 * lane + s;
      const int code = j < qLen ? qcode(j) : 4;
      const int sh = 8 * code;
      plo[s] = (int)(((mat.row[0] >> sh) & 0xff) | (((mat.row[1] >> sh) & 0xff) << 8) | (((mat.row[2] >> sh) & 0xff) << 16) |
                     (((mat.row[3] >> sh) & 0xff) << 24));
      phi2 |= (int)((mat.row[4] >> sh) & 0xff) << (8 * s);
    }
  };
  load_profile();
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int j = base + 2 * lane + s;
    if (in) {
      const int2 v = j <= qLen ? eh[j] : make_int2(0, 0);
      Hs[s] = v.x;
      Es[s] = v.y;
    } else {
      Hs[s] = j == 0 ? h0 : max(0, h0 - oeIns - (j - 1) * eIns);  // row -1, SWUtil.scala:97-104
      Es[s] = 0;
    }
  }
  const int jE0 = 2 * lane * eIns - oeIns;  // j*eIns - oeIns of the even column, j counted from the window's origin
  const int kC = oeIns - eIns;
  const int col0 = 2 * lane, col1 = 2 * lane + 1;  // window coordinates
  int mx = cin(&ExtCarry::mx, h0), max_i = cin(&ExtCarry::max_i, -1), max_j = cin(&ExtCarry::max_j, -1);
  int max_ie = cin(&ExtCarry::max_ie, -1), gscore = cin(&ExtCarry::gscore, -1), max_off = cin(&ExtCarry::max_off, 0);
  int beg = cin(&ExtCarry::beg, 0), end = cin(&ExtCarry::end, qLen), h1raw = cin(&ExtCarry::h1raw, h0 - oDel);
  const int i0 = cin(&ExtCarry::row, 0);
  const int i_tail = amax > 0 ? qLen : 0x7fffffff;
  const int w1 = w + 1;
  for (int i = i0; i < tLen; ++i) {
    if (i >= i_tail) {  // nothing past this row can change the result (tail_row_bound)
      const int U = tail_row_bound(qLen, i, h0, amax, oDel, eDel);
      const int stop = (U <= mx ? 1 : 0) & (U < gscore ? 1 : 0);
      if (stop) break;
    }
    const int tsv = __builtin_amdgcn_readfirstlane((int)ts[i]);  // 8 * target base
    h1raw -= eDel;
    const int h1 = smax2(0, h1raw);    // SWUtil.scala:137-138
    beg = smax2(beg, i - w);           // SWUtil.scala:140-142
    end = smin2(smin2(end, i + w1), qLen);
    if (end - base > 127) {  // column `end` (written this row) lies beyond the window: move the window up
      const int nb = beg & ~1;
      if (end - nb > 127) {  // a band wider than the window: not for this sweep
        *overflow = 1;
        return ExtRes{0, 0, 0, 0, 0, 0};
      }
      const int from = (lane + ((nb - base) >> 1)) << 2;  // byte address of the source lane; lanes past 63 wrap and fetch
#pragma unroll                                            // columns the band has not reached yet (never read before written)
      for (int s = 0; s < 2; ++s) {
        Hs[s] = __builtin_amdgcn_ds_bpermute(from, Hs[s]);
        Es[s] = __builtin_amdgcn_ds_bpermute(from, Es[s]);
      }
      base = nb;
      load_profile();
    }
    const int rbeg = beg - base;       // the band in window coordinates
    const int span = end - beg;
    const unsigned spanA = (unsigned)smax2(span, 0);
    int scv[2];
    if (__builtin_expect(tsv == 32, 0)) {  // an N row
      scv[0] = __builtin_amdgcn_sbfe(phi2, 0u, 8u);
      scv[1] = __builtin_amdgcn_sbfe(phi2, 8u, 8u);
      asm volatile("" : "+v"(scv[0]), "+v"(scv[1]));  // keeps the branch: two selects per row otherwise
    } else {
      scv[0] = __builtin_amdgcn_sbfe(plo[0], (unsigned)tsv, 8u);
      scv[1] = __builtin_amdgcn_sbfe(plo[1], (unsigned)tsv, 8u);
    }
    const unsigned rel0 = (unsigned)(col0 - rbeg), rel1 = (unsigned)(col1 - rbeg);
    const bool act0 = rel0 < spanA, act1 = rel1 < spanA;
    const int a0 = act0 ? max(Hs[0] + scv[0], Es[0]) : NEG_A;
    const int a1 = act1 ? max(Hs[1] + scv[1], Es[1]) : NEG_A;
    const int Pg0 = a0 + jE0, Pg1 = a1 + jE0 + eIns;
    int Pl = max(Pg0, Pg1);
    int scan_a = max((a0 << 7) | col0, (a1 << 7) | col1);  // the row maximum and its LAST column (window coordinates) in one scan
    dual_scan_max(Pl, scan_a);
    const int Pprev = wave_shr1(NEG, Pl);
    const int H0 = max3i(a0, Pprev - kC - jE0, 0);
    const int H1 = max3i(a1, max(Pprev, Pg0) - kC - jE0 - eIns, 0);
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(act0), m1 = __builtin_amdgcn_ballot_w64(act1);
    const unsigned long long z0 = __builtin_amdgcn_ballot_w64(H0 < 1) & m0, z1 = __builtin_amdgcn_ballot_w64(H1 < 1) & m1;
    const int En0 = act0 ? max3i(Es[0] - eDel, H0 - oeDel, 0) : 0;
    const int En1 = act1 ? max3i(Es[1] - eDel, H1 - oeDel, 0) : 0;
    const int hs0 = wave_shr1(h1, H1), hs1 = H0;  // H(i,j-1)
    Hs[0] = rel0 == 0u ? h1 : hs0;               // eh[beg].h = h1, SWUtil.scala:153 (written in every lane: see sw_extend_lean2)
    Hs[1] = rel1 == 0u ? h1 : hs1;
    Es[0] = En0;
    Es[1] = En1;
    const int mkey = smax2(0, __builtin_amdgcn_readlane(scan_a, 63));
    const int m = mkey >> 7, mjr = mkey & 127, mj = base + mjr;

    const int jlast = span > 0 ? end : beg;  // SWUtil.scala:177-182
    if (jlast == qLen) {
      int hlast = h1;
      if (span > 0) {
        const int e = end - base;
        const int he = __builtin_amdgcn_readlane(Hs[0], e >> 1), ho = __builtin_amdgcn_readlane(Hs[1], e >> 1);
        hlast = (e & 1) ? ho : he;
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
    } else {
      const int ze_l = s_lead_zeros(z0 & s_below_mask((mjr + 1) >> 1));
      const int zo_l = s_lead_zeros(z1 & s_below_mask(mjr >> 1));
      const int cl = smax2(ze_l >= 0 ? 2 * (63 - ze_l) : -1, zo_l >= 0 ? 2 * (63 - zo_l) + 1 : -1);
      const int se = (mjr + 2) >> 1, so = (mjr + 1) >> 1;
      const int fe = s_first_one((z0 >> ((mjr + 1) >> 1)) >> ((mjr + 1) & 1));
      const int fo = s_first_one(z1 >> so);
      const int cr = smin2(fe >= 0 ? 2 * (se + fe) : 1 << 20, fo >= 0 ? 2 * (so + fo) + 1 : 1 << 20);
      beg = cl >= 0 ? base + cl + 2 : nb0;
      end = cr < (1 << 20) ? base + cr + 1 : end + 1;
    }
  }
  ExtRes r;
  r.max = mx; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
  return r;
}

// The one-column-per-lane sweep (qLen <= 63) in the same style: see sw_extend_lean2.
template <class QC>
__device__ ExtRes sw_extend_lean1(const int lane, const int qLen, const int tLen, const QC& qcode,
                                  const uint8_t* __restrict__ ts, const MatRows& mat, const int oDel,
                                  const int eDel, const int oIns, const int eIns, const int w, const int zdrop,
                                  const int zmode, const int h0, const int amax) {
  const int oeDel = oDel + eDel, oeIns = oIns + eIns;
  const int code = lane < qLen ? qcode(lane) : 4;
  const int sh = 8 * code;
  const int plo = (int)(((mat.row[0] >> sh) & 0xff) | (((mat.row[1] >> sh) & 0xff) << 8) | (((mat.row[2] >> sh) & 0xff) << 16) |
                        (((mat.row[3] >> sh) & 0xff) << 24));
  const int phi = (int)(int8_t)((mat.row[4] >> sh) & 0xff);
  int Hs = lane == 0 ? h0 : max(0, h0 - oeIns - (lane - 1) * eIns);  // row -1, SWUtil.scala:97-104
  int Es = 0;
  const int jE = lane * eIns - oeIns;  // j*eIns - oeIns
  const int kC = oeIns - eIns;         // (j-1)*eIns = (j*eIns - oeIns) + kC
  int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
  int beg = 0, end = qLen, h1raw = h0 - oDel;
  const int i_tail = amax > 0 ? qLen : 0x7fffffff;
  const int w1 = w + 1;
  for (int i = 0; i < tLen; ++i) {
    if (i >= i_tail) {  // nothing past this row can change the result (tail_row_bound)
      const int U = tail_row_bound(qLen, i, h0, amax, oDel, eDel);
      const int stop = (U <= mx ? 1 : 0) & (U < gscore ? 1 : 0);
      if (stop) break;
    }
    const int tsv = __builtin_amdgcn_readfirstlane((int)ts[i]);  // 8 * target base
    h1raw -= eDel;
    const int h1 = smax2(0, h1raw);    // SWUtil.scala:137-138
    beg = smax2(beg, i - w);           // SWUtil.scala:140-142
    end = smin2(smin2(end, i + w1), qLen);
    const int span = end - beg;
    const unsigned spanA = (unsigned)smax2(span, 0);
    int scv;
    if (__builtin_expect(tsv == 32, 0)) {  // an N row
      scv = phi;
      asm volatile("" : "+v"(scv));  // keeps the branch: a select per row otherwise
    } else {
      scv = __builtin_amdgcn_sbfe(plo, (unsigned)tsv, 8u);
    }
    const unsigned rel = (unsigned)(lane - beg);
    const bool act = rel < spanA;
    const int a = act ? max(Hs + scv, Es) : NEG_A;
    const int Pg = a + jE;
    int Pl = Pg;
    int scan_a = (a << 7) | lane;  // the row maximum and its LAST column in one scan
    dual_scan_max(Pl, scan_a);
    const int Pprev = wave_shr1(NEG, Pl);
    const int H = max3i(a, Pprev - kC - jE, 0);
    const unsigned long long z = __builtin_amdgcn_ballot_w64(H < 1) & __builtin_amdgcn_ballot_w64(act);
    const int En = act ? max3i(Es - eDel, H - oeDel, 0) : 0;
    const int hs = wave_shr1(h1, H);  // H(i,j-1)
    Hs = rel == 0u ? h1 : hs;         // eh[beg].h = h1, SWUtil.scala:153 (written in every lane: see sw_extend_lean2)
    Es = En;
    const int mkey = smax2(0, __builtin_amdgcn_readlane(scan_a, 63));
    const int m = mkey >> 7, mj = mkey & 127;

    const int jlast = span > 0 ? end : beg;  // SWUtil.scala:177-182
    if (jlast == qLen) {
      const int hlast = span > 0 ? __builtin_amdgcn_readlane(Hs, end) : h1;  // end == qLen <= 63 here
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
    // band trimming, SWUtil.scala:202-214: last zero of H left of mj, first zero right of mj
    const int nb0 = beg + (h1 == 0 ? 1 : 0);
    if (z == 0ull) {
      beg = nb0;
      end = end + 1;
    } else {
      const int lzc = s_lead_zeros(z & s_below_mask(mj));
      const int fo = s_first_one((z >> mj) >> 1);
      beg = lzc >= 0 ? 65 - lzc : nb0;
      end = fo >= 0 ? mj + 2 + fo : end + 1;
    }
  }
  ExtRes r;
  r.max = mx; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
  return r;
}

// SWExtend on the register path for any qLen <= 255.  Up to 63 columns: one column per lane; up to 127: two per lane; longer
// flanks: the sliding 128-column window (sw_extend_leanS), started by the slot sweep when the first rows are wider than the
// window (eh: LDS row for the hand-over, qLen + 2 pairs; without it such calls stay on the slot sweep), and run again on the
// slot sweep in the rare case that a later row outgrows the window.  BPSW_EXT_SLIDE=0 at compile time: slot sweep only.
#ifndef BPSW_EXT_SLIDE
#define BPSW_EXT_SLIDE 1
#endif
template <class QC>
__device__ __forceinline__ ExtRes sw_extend_reg_any(const int lane, const int qLen, const int tLen, const QC& qcode,
                                                    const uint8_t* __restrict__ ts, const MatRows& mat, const int oDel,
                                                    const int eDel, const int oIns, const int eIns, const int w,
                                                    const int zdrop, const int zmode, const int h0, const int amax,
                                                    int2* __restrict__ eh = nullptr) {
  const int slots = (qLen + 64) >> 6;
  if (slots == 1) return sw_extend_lean1(lane, qLen, tLen, qcode, ts, mat, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax);
  if (slots == 2) return sw_extend_lean2(lane, qLen, tLen, qcode, ts, mat, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax);
#if BPSW_EXT_SLIDE
  {
    int overflow = 0;
    ExtRes r;
    if (min(qLen, w + 1) <= 127) {  // row 0's band [0, min(qLen, w+1)] fits the window
      r = sw_extend_leanS(lane, qLen, tLen, qcode, ts, mat, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax, nullptr, nullptr, &overflow);
      if (!overflow) return r;
    } else if (eh) {
      ExtCarry c;
      c.handed = 0;
      r = slots == 3 ? sw_extend_reg<3>(lane, qLen, tLen, qcode, ts, mat, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax, eh, &c)
                     : sw_extend_reg<4>(lane, qLen, tLen, qcode, ts, mat, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax, eh, &c);
      if (!c.handed) return r;
      r = sw_extend_leanS(lane, qLen, tLen, qcode, ts, mat, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax, eh, &c, &overflow);
      if (!overflow) return r;
    }
  }
#endif
  // the slot sweep from the first row to the last
  if (slots == 3) return sw_extend_reg<3>(lane, qLen, tLen, qcode, ts, mat, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax);
  return sw_extend_reg<4>(lane, qLen, tLen, qcode, ts, mat, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax);
}

// The sweeps that need few registers: one or two columns per lane and the sliding window, none of which needs an LDS row.
// ext_kernel<.., SHORT> is built from these alone and fits 48 VGPRs -- eight waves per SIMD instead of five (bpsw_extend.hip).
constexpr int EXT_SHORT_QMAX = 255;  // the register sweeps' limit; the host may set a launch's limit lower (127: no window, no deferral)
template <bool WINDOW, class QC>
__device__ __forceinline__ ExtRes sw_extend_reg_short(const int lane, const int qLen, const int tLen, const QC& qcode,
                                                      const uint8_t* __restrict__ ts, const MatRows& mat, const int oDel,
                                                      const int eDel, const int oIns, const int eIns, const int w,
                                                      const int zdrop, const int zmode, const int h0, const int amax,
                                                      int* __restrict__ overflow) {
  if (qLen < 64) return sw_extend_lean1(lane, qLen, tLen, qcode, ts, mat, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax);
  if (!WINDOW || qLen < 128) return sw_extend_lean2(lane, qLen, tLen, qcode, ts, mat, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax);
  // longer flanks (the WINDOW build of the kernel admits them up to 255 bases): the sliding window, when row 0's band
  // [0, min(qLen, w+1)] fits it and as long as no later row outgrows it -- else *overflow = 1 and the task goes to the full kernel
  if (min(qLen, w + 1) > 127) {
    *overflow = 1;
    return ExtRes{0, 0, 0, 0, 0, 0};
  }
  return sw_extend_leanS(lane, qLen, tLen, qcode, ts, mat, oDel, eDel, oIns, eIns, w, zdrop, zmode, h0, amax, nullptr, nullptr, overflow);
}

// Wave-level dequeue: lane 0 alone performs one returning atomic add, the result is broadcast.  Written as
// a single asm statement so that the compiler sees no lane-dependent branch here: with a C-level
// `if (lane == 0) atomicAdd(...)` hipcc threaded that branch together with the lane-0 result store at the end
// of the previous iteration and peeled the other 63 lanes out of the loop, which breaks every cross-lane
// operation of the row sweep.
__device__ __forceinline__ int dequeue_task(int* counter, const int count = 1) {
  int v = count;
  unsigned long long saved;
  asm volatile(
      "s_mov_b64 %1, exec\n\t"
      "s_mov_b64 exec, 1\n\t"
      "global_atomic_add %0, %2, %0, off sc0\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_mov_b64 exec, %1"
      : "+v"(v), "=&s"(saved)
      : "v"(counter)
      : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}

// Stores by lane 0 alone, as single asm statements for the same reason as dequeue_task: a C-level `if (lane == 0) store`
// in the task loop invites the compiler to thread lane 0 and the other 63 lanes through different copies of the loop.
__device__ __forceinline__ void store_lane0_b8(uint8_t* p, int v) {
  unsigned long long saved;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_mov_b64 exec, 1\n\t"
      "global_store_byte %1, %2, off\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(saved)
      : "v"(p), "v"(v)
      : "memory");
}
typedef unsigned int bpsw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_lane0_b128(uint4* p, uint4 val) {
  const bpsw_u32x4 v = {val.x, val.y, val.z, val.w};
  unsigned long long saved;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_mov_b64 exec, 1\n\t"
      "global_store_dwordx4 %1, %2, off\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(saved)
      : "v"(p), "v"(v)
      : "memory");
}

}  // namespace
}  // namespace bpsw
