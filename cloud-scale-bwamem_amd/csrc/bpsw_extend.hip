// bpsw_extend.hip -- banded affine-gap seed extension for gfx950 (boundary 2).
//
// What it computes: for every task of a wire batch (MemChainToAlignBatched.scala:76-172) the result
// of the Scala extension() (MemChainToAlignBatched.scala:789-883): left then right SWExtend
// (SWUtil.scala:61-230) with up to MAX_BAND_TRY=2 band widths, bit-exact.
//
// How: one task per 64-lane wavefront, four independent waves per workgroup, no workgroup barrier.
// The DP is row-synchronous because the band [beg,end) of row i+1 is derived from the finished row i
// (SWUtil.scala:201-214).  A row is swept in 64-column chunks:
//   a(j)   = max(H(i-1,j-1) + S(i,j), E(i,j))                      per lane
//   F(i,j) = max(0, max_{k<j}(a(k) - oeIns - (j-1-k)*eIns))        wave max-plus prefix scan (DPP)
//   H(i,j) = max(a(j), F(i,j)),  E(i+1,j) = max(E(i,j)-eDel, H(i,j)-oeDel, 0)
// (valid because oIns >= 0, checked on the host).  The (H,E) row lives in LDS as int2 per column, the
// 5 x qLen query profile as int8, the unpacked target as bytes.  Row maximum + LAST arg-max
// (SWUtil.scala:158-161) come from a wave max-reduce + ballot; the band trimming loops
// (SWUtil.scala:202-214) are evaluated on 64-bit zero masks held in SGPRs, so all row control is scalar.
#include <stdlib.h>

#include "bpsw_internal.h"
#include "bpsw_wave.h"

namespace bpsw {
namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int EXT_CHUNK = 1;  // tasks per dequeue (larger chunks measured slower: the tail grows faster than the atomic traffic shrinks)

// base k (0-based) of a task's nibble stream: 8 nibbles per word, first base in the top nibble
__device__ __forceinline__ int nibble_at(const uint32_t* __restrict__ words, int k) {
  const uint32_t w = words[k >> 3];
  const int c = (int)((w >> (28 - 4 * (k & 7))) & 0xFu);
  return c > 4 ? 4 : c;  // codes are 0..4 (LocusEncode); never index the matrix out of bounds
}

struct ExtRes {
  int max, qle, tle, gtle, gscore, max_off;
};

// One SWExtend call (SWUtil.scala:61-230) executed by a whole wave.  All scalar state is wave-uniform.
__device__ ExtRes sw_extend_wave(const int lane, const int qLen, const int tLen, int2* __restrict__ eh,
                                 const int8_t* __restrict__ qp, const uint8_t* __restrict__ ts, const int oDel,
                                 const int eDel, const int oIns, const int eIns, const int w, const int zdrop,
                                 const int zmode, const int h0) {
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
// cycles; mixed streams from several waves reach about one instruction per 2.4 cycles, and once the GPU is saturated
// the kernel time follows the TOTAL instruction count (1.15 ns x (VALU + SALU) / SIMD), not the split.  Two builds:
//   BPSW_EXT_VECTOR_CONTROL 1 (default): uniform values kept in VGPRs through an opaque asm -> 355 M VALU + 200 M SALU
//                                        per 30 k-task batch, 0.713 ms stand-alone
//   BPSW_EXT_VECTOR_CONTROL 0          : control on the scalar pipe -> 239 M VALU + 323 M SALU, 0.748 ms stand-alone
// Both give 2.95 ms per bench step when the step's batches overlap on the device.
#ifndef BPSW_EXT_VECTOR_CONTROL
#define BPSW_EXT_VECTOR_CONTROL 1
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

template <int S>
__device__ ExtRes sw_extend_reg(const int lane, const int qLen, const int tLen, const uint32_t* __restrict__ words,
                                const int qStart, const uint8_t* __restrict__ ts, const MatRows& mat, const int oDel,
                                const int eDel, const int oIns, const int eIns, const int w, const int zdrop,
                                const int zmode, const int h0) {
  const int oeDel = oDel + eDel, oeIns = oIns + eIns;
  int Hs[S], Es[S], As[S], plo[S], phi[S], jE[S], c2[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int j = 64 * s + lane;
    const int code = j < qLen ? nibble_at(words, qStart + j) : 4;
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

  for (int i = 0; i < tLen; ++i, iv += 1) {
    const int tsv = ts[i];  // 8 * target base, same in every lane
    const bool isN = tsv == 32;
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
      const int sc = isN ? phi[s] : __builtin_amdgcn_sbfe(plo[s], (unsigned)tsv, 8u);
      const int a = act ? max(Hs[s] + sc, Es[s]) : NEG;
      As[s] = a;
      int Pg = a + jE[s];
      scan_a = a;
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
      zm[s] = __builtin_amdgcn_ballot_w64((act ? H : -1) == 0);
      const int En = act ? max3i(Es[s] - eDel, H - oeDel, 0) : 0;  // E(i+1,j); eh[end].e = 0
      int hsh = wave_shr1(hl_prev, H);                             // H(i,j-1)
      if (S > 1) hl_prev = __builtin_amdgcn_readlane(H, 63);
      hsh = rel == 0u ? h1 : hsh;                                  // eh[beg].h = h1, SWUtil.scala:153
      Hs[s] = upd ? hsh : Hs[s];
      Es[s] = upd ? En : Es[s];
    }
    const int m = max(0, S > 1 ? carry_a : __builtin_amdgcn_readlane(scan_a, 63));  // scalar

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
      bm = 63 - s_lead_zeros(__builtin_amdgcn_ballot_w64(As[0] == m));
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
      const int di = iv - max_i, dj = mj - max_j;
      const bool A = di > dj;
      const bool B = mx - m - (di - dj) * eDel > zdrop;
      const bool C = mx - m - (dj - di) * eIns > zdrop;
      const bool stop = zmode == BPSW_ZDROP_SCALA ? (A && (B || C)) : (A ? B : C);
      if (any_lane(stop)) break;
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
      lzc = s_lead_zeros(zm[0] & s_below_mask(bm));
      fo = s_first_one((zm[0] >> bm) >> 1);
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

// stage the target of one side in LDS as 8*code bytes (the shift the register path feeds to v_bfe)
__device__ void load_target_shifts(const int lane, const uint32_t* __restrict__ words, const int rStart, const int rLen,
                                   uint8_t* __restrict__ ts) {
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < rLen; i += 64) ts[i] = (uint8_t)(8 * nibble_at(words, rStart + i));
  __builtin_amdgcn_wave_barrier();
}

// Unpack one side of a task into LDS: the 5 x qLen query profile and the target bytes.
__device__ void load_side(const int lane, const uint32_t* __restrict__ words, const int qStart, const int qLen,
                          const int rStart, const int rLen, const MatRows& mat, int8_t* __restrict__ qp,
                          uint8_t* __restrict__ ts) {
  for (int j = lane; j < qLen; j += 64) {
    const int c = nibble_at(words, qStart + j);
#pragma unroll
    for (int k = 0; k < 5; ++k) qp[k * qLen + j] = (int8_t)((mat.row[k] >> (8 * c)) & 0xff);
  }
  for (int i = lane; i < rLen; i += 64) ts[i] = (uint8_t)nibble_at(words, rStart + i);
  __builtin_amdgcn_wave_barrier();
}

// Wave-level dequeue: lane 0 alone performs one returning atomic add, the result is broadcast.  Written as
// a single asm statement so that the compiler sees no lane-dependent branch here: with a C-level
// `if (lane == 0) atomicAdd(...)` hipcc threaded that branch together with the lane-0 result store at the end
// of the previous iteration and peeled the other 63 lanes out of the loop, which breaks every cross-lane
// operation of the row sweep.
__device__ __forceinline__ int dequeue_task(int* counter) {
  int v = 1;
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

__device__ __forceinline__ int lo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t v) { return (int)(int16_t)(v >> 16); }

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void ext_kernel(const uint32_t* __restrict__ wire, const int n_tasks,
                                                                     int16_t* __restrict__ out, const ExtScoring sc,
                                                                     const int qcap, const int rcap,
                                                                     const int lds_per_wave,
                                                                     int* __restrict__ next_task,
                                                                     const int* __restrict__ task_list) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  unsigned char* base = smem + (size_t)wave * lds_per_wave;
  int2* eh = reinterpret_cast<int2*>(base);
  int8_t* qp = reinterpret_cast<int8_t*>(base + 8 * (size_t)(qcap + 2));
  uint8_t* ts = reinterpret_cast<uint8_t*>(qp + 5 * (size_t)qcap);

  // header, MemChainToAlignBatched.scala:78-84 (signed bytes)
  const uint32_t hdr0 = wire[0], hdr1 = wire[1];
  const int oDel = (int8_t)(hdr0 & 0xff), eDel = (int8_t)((hdr0 >> 8) & 0xff);
  const int oIns = (int8_t)((hdr0 >> 16) & 0xff), eIns = (int8_t)((hdr0 >> 24) & 0xff);
  const int penClip5 = (int8_t)(hdr1 & 0xff), penClip3 = (int8_t)((hdr1 >> 8) & 0xff);
  const int wBand = (int8_t)((hdr1 >> 16) & 0xff);

  // Tasks differ in cost by an order of magnitude, so waves pull them from a shared counter instead of
  // striding: a wave takes the next task when it finishes one and leaves when the counter passes n_tasks.
  // One counter word sustains only ~90 returning atomics per microsecond chip-wide (MI355X_MICROARCH.md, row
  // "dequeue"), which at one atomic per task would cap a 30 k-task batch near 0.4 ms; waves therefore take tickets
  // in chunks of EXT_CHUNK consecutive tasks.
  int ticket = 0, ticket_end = 0;
  for (;;) {
    if (ticket == ticket_end) {
      ticket = dequeue_task(next_task) * EXT_CHUNK;
      ticket_end = min(ticket + EXT_CHUNK, n_tasks);
      if (ticket >= n_tasks) break;
    }
    const int task = task_list ? uni(task_list[ticket]) : ticket;  // n_tasks counts the entries of task_list when given
    ++ticket;
    const uint32_t* rec = wire + 8 + 8 * (size_t)task;  // MemChainToAlignBatched.scala:95-117
    const uint32_t r0 = rec[0], r1 = rec[1], r3 = rec[3], r4 = rec[4], r5 = rec[5], r6 = rec[6];
    const int lq = uni(lo16(r0)), lr = uni(hi16(r0)), rq = uni(lo16(r1)), rr = uni(hi16(r1));
    const uint32_t* words = wire + (size_t)uni((int)rec[2]);
    const int regScore0 = uni(lo16(r3)), qBeg = uni(hi16(r3)), h0 = uni(lo16(r4));
    const int lMaxIns = max(1, uni(lo16(r5))), lMaxDel = max(1, uni(hi16(r5)));  // SWUtil.scala:110-115
    const int rMaxIns = max(1, uni(lo16(r6))), rMaxDel = max(1, uni(hi16(r6)));
    const int idx = uni((int)rec[7]);

    // extension(), MemChainToAlignBatched.scala:789-883: side 0 = left (penClip5), side 1 = right (penClip3)
    int aw[2] = {wBand, wBand};
    int regScore = regScore0;
    int outQBeg = 0, outRBeg = 0, outQEnd = rq, outREnd = 0, trueScore = regScore0, score = -1;
    for (int side = 0; side < 2; ++side) {
      const int qLen = side ? rq : lq, rLen = side ? rr : lr;
      if (qLen <= 0) continue;
      const int qStart = side ? lq : 0, rStart = side ? lq + rq + lr : lq + rq;
      const int maxIns = side ? rMaxIns : lMaxIns, maxDel = side ? rMaxDel : lMaxDel;
      const int penClip = side ? penClip3 : penClip5;
      const int hInit = side ? regScore : h0;  // the right extension starts from the score after the left one
      const int sc0 = regScore;
      // register path: needs one lane per column 0..qLen and oeIns > 0 (see sw_extend_reg)
      const bool reg_path = qLen <= 255 && oIns + eIns > 0;
      if (reg_path) load_target_shifts(lane, words, rStart, rLen, ts);
      else load_side(lane, words, qStart, qLen, rStart, rLen, sc.mat, qp, ts);
      ExtRes r = {0, 0, 0, 0, 0, 0};
      for (int i = 0; i < 2; ++i) {  // MAX_BAND_TRY
        const int prev = regScore;
        aw[side] = wBand << i;
        const int w = min(min(aw[side], maxIns), maxDel);
        if (reg_path) {
          switch ((qLen + 64) >> 6) {
            case 1: r = sw_extend_reg<1>(lane, qLen, rLen, words, qStart, ts, sc.mat, oDel, eDel, oIns, eIns, w, sc.zdrop, sc.zdrop_mode, hInit); break;
            case 2: r = sw_extend_reg<2>(lane, qLen, rLen, words, qStart, ts, sc.mat, oDel, eDel, oIns, eIns, w, sc.zdrop, sc.zdrop_mode, hInit); break;
            case 3: r = sw_extend_reg<3>(lane, qLen, rLen, words, qStart, ts, sc.mat, oDel, eDel, oIns, eIns, w, sc.zdrop, sc.zdrop_mode, hInit); break;
            default: r = sw_extend_reg<4>(lane, qLen, rLen, words, qStart, ts, sc.mat, oDel, eDel, oIns, eIns, w, sc.zdrop, sc.zdrop_mode, hInit); break;
          }
        } else {
          r = sw_extend_wave(lane, qLen, rLen, eh, qp, ts, oDel, eDel, oIns, eIns, w, sc.zdrop, sc.zdrop_mode, hInit);
        }
        regScore = r.max;
        if (regScore == prev || r.max_off < (aw[side] >> 1) + (aw[side] >> 2)) break;
      }
      score = regScore;
      const bool local = r.gscore <= 0 || r.gscore <= regScore - penClip;  // local extension vs reaching the query end
      if (side == 0) {
        outQBeg = local ? qBeg - r.qle : 0;
        outRBeg = local ? -r.tle : -r.gtle;
        trueScore = local ? regScore : r.gscore;
      } else {
        outQEnd = local ? r.qle : rq;
        outREnd = local ? r.tle : r.gtle;
        trueScore += (local ? regScore : r.gscore) - sc0;
      }
    }
    const int aw0 = aw[0], aw1 = aw[1];
    const int width = aw0 > aw1 ? aw0 : aw1;
    if (lane == 0) {  // MemChainToAlignBatched.scala:181-188: 10 int16 per task
      uint32_t* o = reinterpret_cast<uint32_t*>(out + 10 * (size_t)task);
      o[0] = (uint32_t)idx;
      o[1] = ((uint32_t)outQBeg & 0xffffu) | ((uint32_t)outQEnd << 16);
      o[2] = ((uint32_t)outRBeg & 0xffffu) | ((uint32_t)outREnd << 16);
      o[3] = ((uint32_t)score & 0xffffu) | ((uint32_t)trueScore << 16);
      o[4] = (uint32_t)width & 0xffffu;
    }
  }
}

// ---- table scan: validates the batch and finds the LDS capacities the main launch needs --------
__global__ void ext_prepass_kernel(const uint32_t* __restrict__ wire, const unsigned long long wire_words,
                                   const int n_tasks, ExtPrepass* __restrict__ pre) {
  int mq = 0, mr = 0, err = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if ((int)wire[2] != n_tasks) err = 1;
    const uint32_t hdr0 = wire[0];
    const int oIns = (int8_t)((hdr0 >> 16) & 0xff), eIns = (int8_t)((hdr0 >> 24) & 0xff);
    const int oDel = (int8_t)(hdr0 & 0xff), eDel = (int8_t)((hdr0 >> 8) & 0xff);
    if (oIns < 0 || eIns < 0 || oDel < 0 || eDel < 0) err = 2;  // the prefix-scan form of F needs oIns >= 0
    pre->reserved = oIns + eIns > 0 ? 1 : 0;                    // quad-task kernels are usable
  }
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_tasks; t += gridDim.x * blockDim.x) {
    const uint32_t* rec = wire + 8 + 8 * (size_t)t;
    const int lq = lo16(rec[0]), lr = hi16(rec[0]), rq = lo16(rec[1]), rr = hi16(rec[1]);
    const long long pos = (int)rec[2];
    if (lq < 0 || lr < 0 || rq < 0 || rr < 0) { err = 3; continue; }
    const long long words = ((long long)lq + lr + rq + rr + 7) / 8;
    if (pos < 8 + 8ll * n_tasks || (unsigned long long)(pos + words) > wire_words) err = 4;
    mq = max(mq, max(lq, rq));
    mr = max(mr, max(lr, rr));
  }
  if (mq) atomicMax(&pre->max_qlen, mq);
  if (mr) atomicMax(&pre->max_rlen, mr);
  if (err) atomicMax(&pre->error, err);
}

}  // namespace

size_t ext_lds_per_wave(int qcap, int rcap) {
  size_t b = 8 * (size_t)(qcap + 2) + 5 * (size_t)qcap + (size_t)rcap;
  return (b + 15) & ~(size_t)15;
}

void launch_ext_prepass(const uint32_t* d_wire, size_t wire_words, int n_tasks, ExtPrepass* d_pre, hipStream_t s) {
  const int threads = 256;
  int blocks = (n_tasks + threads - 1) / threads;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(ext_prepass_kernel, dim3(blocks), dim3(threads), 0, s, d_wire, (unsigned long long)wire_words,
                     n_tasks, d_pre);
}

hipError_t launch_ext_kernel(const uint32_t* d_wire, int n_tasks, int16_t* d_out, const ExtScoring& sc, int qcap,
                             int rcap, int num_cu, int* d_counter, const int* d_task_list, hipStream_t s) {
  if (n_tasks <= 0) return hipSuccess;
  // round the capacities so that a handful of LDS configurations cover all batches
  qcap = (qcap + 31) & ~31;
  rcap = (rcap + 63) & ~63;
  const size_t per_wave = ext_lds_per_wave(qcap, rcap);
  const size_t lds = per_wave * WAVES_PER_BLOCK;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static thread_local size_t attr_set = 0;
  if (lds > 64 * 1024 && lds > attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ext_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = lds;
  }
  // resident workgroups per CU: 8 waves/SIMD = 8 blocks of 4 waves, capped by LDS
  int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
  static const int cap_per_cu = getenv("BPSW_EXT_BLOCKS_PER_CU") ? atoi(getenv("BPSW_EXT_BLOCKS_PER_CU")) : 8;
  per_cu = per_cu > cap_per_cu ? cap_per_cu : (per_cu < 1 ? 1 : per_cu);
  int blocks = (n_tasks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  const int max_blocks = num_cu * per_cu;
  if (blocks > max_blocks) blocks = max_blocks;
  hipError_t me = hipMemsetAsync(d_counter, 0, sizeof(int), s);
  if (me != hipSuccess) return me;
  hipLaunchKernelGGL(ext_kernel, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), lds, s, d_wire, n_tasks, d_out, sc, qcap,
                     rcap, (int)per_wave, d_counter, d_task_list);
  return hipGetLastError();
}

}  // namespace bpsw
