// bpsw_extend_sift.hip -- the exact shortcuts of the extension (bpsw_extend_core.h: flank_closed_form with its single-gap
// certificate and two-gap-open tests, flank_start_gap_form), evaluated one TASK per LANE in front of ext_kernel.
//
// Why: three quarters of the tasks of low-error reads are resolved by the shortcuts on both sides, and ext_kernel spends a whole
// wavefront on each of them -- 64 lanes for a flank of ~65 bases, two DPP scans per 64 columns and shift of the certificate, several
// hundred wave-instructions per task that is never swept.  The shortcuts decide on the main diagonal's few mismatches, so they are
// word-parallel work on the nibble stream itself: eight bases per XOR.  Here a wavefront copies the nibble streams of 64 consecutive
// tasks into LDS with coalesced loads (they lie back to back in the wire batch; a batch laid out otherwise is left to ext_kernel),
// every lane takes one task and runs the same decisions in 32-bit integer arithmetic on funnel-shifted words of its stream.
//
// What it hands on (per task, `flag`):  1 = both sides resolved, the result record is written, ext_kernel skips the task;
// 2 = a record per side for ext_kernel: "no form holds" (sweep at once) or "resolved when hInit >= hmin, with these results
// relative to hInit" -- every form is linear in the start score, and the right side's start score is the left side's result, which
// only ext_kernel knows when the left side needs the DP;  0 = not examined: ext_kernel does it all.
// A side that this kernel does not judge (an N in either flank) is marked "not examined" and evaluated by ext_kernel's own
// wave-wide code: the decisions are the same either way, never a guess.
// Restrictions (the launch checks them): wire format 1, a matrix whose sixteen base-vs-base entries are a on the diagonal and one
// value a - dm off it (sc.exact_a > 0 says the first, `dm` is passed in), flanks up to 127 bases.
//
// The single-gap certificate without scans.  With one mismatch score every term of bpsw_extend_core.h's conditions is a multiple of
// dm: along a shifted diagonal a column GAINS dm where the main diagonal mismatches and the shifted one matches, LOSES dm where the
// main one matches and the shifted one does not, and G(y) - min_{z<y} G(z) = dm * r(y) with r(y) = c(y) + max(0, r(y-1)).  So
// W(y) = max(0, r(y)) is a counter that goes up at a gain and down (not below 0) at a loss, gains are among the <= 3 deficit columns
// of the flank, and the conditions can only fail at a handful of columns: at a gain (insertion: dm (W + 1) >= T), at the end of the
// range, and -- deletion, where the comparison is with the main diagonal d rows further down, tail(x) = A(min(x+d, n-1)) - A(x) --
// wherever dm W(x) - tail(x) can rise: at x = 0, at a deficit column p, at p - d (p enters the tail), and in the last d columns (the
// tail gets shorter).  Between such columns only the number of losses matters, and only up to W: a popcount over words, nearly
// always one word.  Same decisions as the scans (tests/test_extend_gpu.py, tools/soak_cert2.py compare the verdicts side by side).
// Cites: the forms follow SWUtil.scala:61-230 / MemChainToAlignBatched.scala:789-883 exactly as bpsw_extend_core.h derives them;
// the chaining of the two sides mirrors ext_kernel (bpsw_extend.hip).
#include "bpsw_extend_core.h"
#include "bpsw_extend_sift_core.h"

namespace bpsw {
namespace {
using namespace sift;

constexpr int SIFT_RAW_WORDS = 4096;  // nibble words of 64 consecutive tasks a wavefront stages (16 KB; 2x150 bp tasks average ~41)

__device__ __forceinline__ int s_lo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int s_hi16(uint32_t v) { return (int)(int16_t)(v >> 16); }

// the eight bases k .. k+7 of the 2-bit reference (four bases per byte, first base in the top bits: util/BNTSeqUtil.scala:56-73) as 16
// bits, first base on top; the buffer is 256-byte aligned and padded (bpsw_ref_load), bases outside 0 .. l_pac-1 read as anything
__device__ __forceinline__ uint32_t pac_bases8(const uint8_t* __restrict__ pac, const long long k) {
  const long long A = k >> 4;
  const uint32_t* __restrict__ p32 = reinterpret_cast<const uint32_t*>(pac);
  const uint32_t d0 = A >= 0 ? p32[A] : 0u, d1 = A + 1 >= 0 ? p32[A + 1] : 0u;
  const unsigned long long v = ((unsigned long long)__builtin_bswap32(d0) << 32) | __builtin_bswap32(d1);
  return (uint32_t)((v << (((int)k & 15) << 1)) >> 48);
}
// 16 bits of eight 2-bit codes -> eight nibbles, order kept
__device__ __forceinline__ uint32_t spread_codes(uint32_t x) {
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  return (x | (x << 2)) & 0x33333333u;
}
// the order of eight 2-bit codes reversed
__device__ __forceinline__ uint32_t reverse_codes(uint32_t x) {
  const uint32_t y = __brev(x) >> 16;
  return ((y & 0x5555u) << 1) | ((y >> 1) & 0x5555u);
}
// the target flank of a coordinate task (bnsGetSeq of MemChainToAlignBatched.scala:363 with the left flank reversed, :511-517): base i
// is position pos + step * i of the doubled reference, positions >= l_pac being the reverse strand, complemented.  A flank lies on
// one strand (scan_wire), so the 2-bit stream is read forwards or backwards, eight bases at a time.
__device__ __forceinline__ void sift_stage_pac(uint32_t* __restrict__ dst, const uint8_t* __restrict__ pac, const long long l_pac,
                                               const long long pos, const int step, const int nbases) {
  const bool rev = pos >= l_pac;
  const long long k0 = rev ? (l_pac << 1) - 1 - pos : pos;
  const bool up = rev ? step < 0 : step > 0;
  const int nw = (nbases + 7) >> 3;
  for (int w = 0; w < nw; ++w) {
    uint32_t x = up ? pac_bases8(pac, k0 + 8 * w) : reverse_codes(pac_bases8(pac, k0 - 8 * w - 7));
    if (rev) x ^= 0xFFFFu;
    dst[w] = spread_codes(x);
  }
  dst[nw] = 0u;
}

__device__ __forceinline__ uint4 pack_rec(const SideRec& r) {
  return make_uint4(((uint32_t)r.kind & 0xffu) | ((uint32_t)r.hmin << 8), ((uint32_t)r.max_rel & 0xffffu) | ((uint32_t)r.g_rel << 16),
                    ((uint32_t)r.qle & 0xffffu) | ((uint32_t)r.tle << 16), ((uint32_t)r.gtle & 0xffffu) | ((uint32_t)r.max_off << 16));
}

constexpr int SIFT_T_WORDS = 21;  // a coordinate task's staged target flank: 127 + SIFT_MAX_SHIFTS + 4 bases and a spare word, odd (LDS banks)

// one wavefront per workgroup, 64 consecutive tasks
template <bool COORD>
__global__ __launch_bounds__(64) void ext_sift_kernel(const uint32_t* __restrict__ wire, const int n_tasks, int16_t* __restrict__ out,
                                                      const ExtScoring sc, const int dm, const int qmax, uint8_t* __restrict__ flag,
                                                      uint4* __restrict__ recs, const ExtPrepass* __restrict__ pre,
                                                      int* __restrict__ todo_count, int* __restrict__ todo_list, const int heavy_min) {
  // asynchronous entry (bpsw_extend_batch_device): the table scan ran just before on the same stream and nobody has read it back
  // yet -- a malformed batch is left untouched (ext_kernel behind this launch does the same and never looks at the flags)
  if (pre && pre->error != 0) return;
  // (wave priorities: bpsw_swalign.hip, swp_kernel)
#ifndef BPSW_SIFT_PRIO
#define BPSW_SIFT_PRIO 2
#endif
  if (BPSW_SIFT_PRIO) __builtin_amdgcn_s_setprio(BPSW_SIFT_PRIO);
  __shared__ uint32_t raw[SIFT_RAW_WORDS + 4 + (COORD ? 2 * 64 * SIFT_T_WORDS : 0)];
  constexpr int T_BASE = SIFT_RAW_WORDS + 4;  // COORD: the target flank of (side, lane) at T_BASE + (side * 64 + lane) * SIFT_T_WORDS
  __shared__ int items[128 * 8];   // the flanks whose closed form waits for its certificate: query stream, qs | ts << 8, target stream, n | tLen << 8, k, p0, p1, p2
  __shared__ int item_fail[128];
  const int lane = threadIdx.x;
  const int task = (int)blockIdx.x * 64 + lane;
  const bool live = task < n_tasks;
  const uint32_t hdr0 = wire[0], hdr1 = wire[1];
  SiftParams P;
  P.oDel = (int8_t)(hdr0 & 0xff); P.eDel = (int8_t)((hdr0 >> 8) & 0xff);
  P.oIns = (int8_t)((hdr0 >> 16) & 0xff); P.eIns = (int8_t)((hdr0 >> 24) & 0xff);
  const int penClip5 = (int8_t)(hdr1 & 0xff), penClip3 = (int8_t)((hdr1 >> 8) & 0xff);
  P.wBand = (int8_t)((hdr1 >> 16) & 0xff);
  P.zdrop = sc.zdrop; P.certify = sc.certify; P.dm = dm;
  const int oe_min = min(P.oIns + P.eIns, P.oDel + P.eDel);
  P.a = (oe_min > 0 && P.wBand >= 2) ? sc.exact_a : 0;

  const uint32_t* rec = wire + 8 + (COORD ? 10 : 8) * (size_t)(live ? task : n_tasks - 1);
  const uint32_t r0 = rec[0], r1 = rec[1], r3 = rec[3], r4 = rec[4];
  const int lq = s_lo16(r0), lr = s_hi16(r0), rq = s_lo16(r1), rr = s_hi16(r1);
  const int pos = (int)rec[2];
  const int nwords = COORD ? (lq + rq + 7) >> 3 : (lq + lr + rq + rr + 7) >> 3;  // a coordinate task's stream holds the query flanks only
  // the streams of the wave's tasks lie back to back in the batch (MemChainToAlignBatched.scala:125-170 appends them in task order):
  // one coalesced copy brings them into LDS.  Anything else -- or longer tasks than the buffer holds -- is left to ext_kernel.
  const int base = uni(pos);
  const int span = wave_max(pos + nwords) - base;
  const bool fits = __builtin_amdgcn_ballot_w64(pos < base) == 0ull && span <= SIFT_RAW_WORDS;
  // The tasks this kernel leaves to ext_kernel go on its to-do list (one atomic per wave and class): todo_count[0] of them from
  // the front of todo_list, and -- the ones with the longest sweeps ahead, which ext_kernel takes first: a launch ends with the
  // last task taken -- todo_count[1] from its back.
  const auto todo = [&](const bool mine_todo, const bool heavy) {
    if (!todo_list) return;
#pragma unroll
    for (int cls = 0; cls < 2; ++cls) {
      const bool in = mine_todo && (heavy == (cls == 1));
      const unsigned long long v = __builtin_amdgcn_ballot_w64(in);
      if (v) {
        const int first = __builtin_ctzll(v);
        int at = 0;
        if (lane == first) at = atomicAdd(todo_count + cls, __popcll(v));
        at = __builtin_amdgcn_readlane(at, first) + __popcll(v & ((1ull << lane) - 1ull));
        if (in) todo_list[cls ? n_tasks - 1 - at : at] = task;
      }
    }
  };
  if (P.a <= 0 || !fits) {
    if (live) flag[task] = 0;
    todo(live, false);
    return;
  }
  uint32_t n_codes = 0u;  // codes above 3 (N) anywhere in the wave's streams?  (mostly none: then no lane looks for them again)
  {
    const uint32_t* __restrict__ src = wire + (size_t)base;
    int i = lane;
    for (; i + 192 < span; i += 256) {  // four loads in flight per lane
      const uint32_t v0 = src[i], v1 = src[i + 64], v2 = src[i + 128], v3 = src[i + 192];
      raw[i] = v0; raw[i + 64] = v1; raw[i + 128] = v2; raw[i + 192] = v3;
      n_codes |= (v0 | v1 | v2 | v3) & 0xCCCCCCCCu;
    }
    for (; i < span; i += 64) { const uint32_t v = src[i]; raw[i] = v; n_codes |= v & 0xCCCCCCCCu; }
  }
  if (lane < 4) raw[span + lane] = 0u;
  item_fail[lane] = 0; item_fail[64 + lane] = 0;
  const bool any_n = __builtin_amdgcn_ballot_w64(n_codes != 0u) != 0ull;
  __syncthreads();
  const bool mine = live && lq <= qmax && rq <= qmax;  // else not a task the short build of ext_kernel takes
  const int regScore0 = s_lo16(r3), qBeg = s_hi16(r3), h0 = s_lo16(r4);
  const int idx = (int)rec[7];
  const uint32_t* my_raw = raw + (pos - base);
  // what a side's forms see of its target flank: all of it, or -- a coordinate batch -- what ext_kernel stages (bpsw_extend.hip)
  const auto t_len = [&](const int side) {
    const int qLen = side ? rq : lq, rLen = side ? rr : lr;
    return COORD ? min(rLen, qLen + (P.wBand << 1) + 2) : rLen;
  };
  const auto side_seq = [&](const int side) {
    if constexpr (COORD) return SiftSeq{my_raw, side ? lq : 0, raw + T_BASE + (side * 64 + lane) * SIFT_T_WORDS, 0};
    else return SiftSeq{my_raw, side ? lq : 0, my_raw, side ? lq + rq + lr : lq + rq};
  };
  if constexpr (COORD) {  // the target flanks from the device-resident reference (the 2-bit reference holds no N)
    const long long seedRb = (long long)(((unsigned long long)rec[9] << 32) | rec[8]);
    const int seedLen = s_hi16(r4);
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int qLen = side ? rq : lq;
      if (mine && qLen > 0)
        sift_stage_pac(raw + T_BASE + (side * 64 + lane) * SIFT_T_WORDS, sc.pac, sc.l_pac, side ? seedRb + seedLen : seedRb - 1, side ? 1 : -1,
                       min(t_len(side), qLen + SIFT_MAX_SHIFTS + 4));
    }
  }

  // ---- 1: the closed form of both sides up to its certificate; the flanks that need one queue up in LDS ---------------------------
  int n_items = 0, max_di = 0, max_dd = 0;
  SideRec sr0 = {SIFT_UNSEEN, 0, 0, 0, 0, 0, 0, 0}, sr1 = sr0;
  int st0 = CF_UNSEEN, st1 = CF_UNSEEN, item0 = -1, item1 = -1;  // CF_* per side
  const auto closed = [&](const int side, SideRec* r, int* st, int* item) {
    const int qLen = side ? rq : lq, rLen = t_len(side);
    const SiftSeq s = side_seq(side);
    int k = 0, p[3] = {0, 0, 0}, dI = 0, dD = 0;
    *st = CF_UNSEEN;
    if (mine && qLen > 0) {
      uint32_t n_seen = 0u;
      if (any_n) {  // a code above 3 (N) anywhere in the two flanks: the side is left to ext_kernel
        for (int j = 0; j < qLen; j += 8) n_seen |= s.q8(j) & top_nibbles(qLen - j) & 0xCCCCCCCCu;
        if (!COORD) for (int j = 0; j < rLen; j += 8) n_seen |= s.t8(j) & top_nibbles(rLen - j) & 0xCCCCCCCCu;
      }
      if (!n_seen) *st = sift_closed_form(s, qLen, rLen, P, r, &k, p, &dI, &dD);
    }
    const bool need = *st == CF_IF_CERTIFIED;
    const unsigned long long needs = __builtin_amdgcn_ballot_w64(need);
    if (need) {
      const int it = n_items + __popcll(needs & ((1ull << lane) - 1ull));
      *item = it;
      int* e = items + 8 * it;
      e[0] = (int)(s.qraw - raw); e[1] = s.qs | (s.ts << 8); e[2] = (int)(s.traw - raw); e[3] = qLen | (rLen << 8); e[4] = k; e[5] = p[0]; e[6] = p[1]; e[7] = p[2];
    }
    n_items += __popcll(needs);
    max_di = max(max_di, wave_max(need ? dI : 0));
    max_dd = max(max_dd, wave_max(need ? dD : 0));
  };
  closed(0, &sr0, &st0, &item0);
  closed(1, &sr1, &st1, &item1);
  __syncthreads();

  // ---- 2: the certificates, one (flank, shift) pair per lane; the insertion shifts first, then the deletion shifts (the two
  // kinds share no code: mixed in one round every lane would wait through both) ------------------------------------------------
#pragma unroll
  for (int kind = 0; kind < 2; ++kind) {
    const int per = kind ? max_dd : max_di;  // shifts per flank in this pass (the largest any queued flank needs)
    for (int slot0 = 0; slot0 < n_items * per; slot0 += 64) {
      const int slot = slot0 + lane;
      if (slot < n_items * per) {
        const int it = slot / per, d = slot - it * per + 1;
        const int* e = items + 8 * it;
        const int k = e[4], D = k * dm;
        const int mine_d = kind ? max(0, (D - P.oDel) / P.eDel) : max(0, (D - P.oIns) / P.eIns);
        if (d <= mine_d) {
          const SiftSeq s = {raw + e[0], e[1] & 0xff, raw + e[2], e[1] >> 8};
          if (!sift_certificate_shift(s, e[3] & 0xff, e[3] >> 8, P, k, e[5], e[6], e[7], kind == 0, d)) item_fail[it] = 1;
        }
      }
    }
  }
  __syncthreads();
  // ---- 3: the start-gap form where the closed form does not hold; extension() over the sides that are resolved ---------------------
  const auto settle = [&](const int side, SideRec* r, int st, const int item) {
    const int qLen = side ? rq : lq, rLen = t_len(side);
    if (qLen <= 0) return;
    if (st == CF_IF_CERTIFIED) st = item_fail[item] ? CF_FAILS : CF_HOLDS;
    if (st == CF_FAILS) {
      const SiftSeq s = side_seq(side);
      r->kind = SIFT_FAIL;
      (void)sift_start_gap_form(s, qLen, rLen, P, r);
    } else if (st == CF_UNSEEN) {
      r->kind = SIFT_UNSEEN;
    }
  };
  int fl = 0;          // what ext_kernel finds in flag[task]: 0 nothing, 1 the record is written, 2 a verdict per side in recs
  bool heavy = false;  // a long sweep ahead
  if (!mine) {
    heavy = live && heavy_min > 0;      // a flank of 128 bases or more, left to ext_kernel whole
  } else {
    settle(0, &sr0, st0, item0);
    settle(1, &sr1, st1, item1);
    // extension() over the sides that are resolved (sift_chain, bpsw_extend_sift_core.h)
    const SiftTask T = {lq, rq, regScore0, qBeg, h0, idx, penClip5, penClip3, P.wBand};
    uint32_t rec_out[5];
    if (sift_chain(T, sr0, sr1, rec_out)) {
      uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)sc.out_stride * (size_t)task);
#pragma unroll
      for (int w = 0; w < 5; ++w) o[w] = rec_out[w];
      if (sc.side_how) {
        if (lq > 0) sc.side_how[2 * (size_t)task] = 1;
        if (rq > 0) sc.side_how[2 * (size_t)task + 1] = 1;
      }
      fl = 1;
    } else {
      // a form that did not hold for the start score this kernel could see is judged again by ext_kernel (another form may hold)
      recs[2 * (size_t)task] = pack_rec(sr0);
      recs[2 * (size_t)task + 1] = pack_rec(sr1);
      fl = 2;
      // the rows its sweeps may take: the bases of the flanks no form resolved
      heavy = heavy_min > 0 && (sr0.kind != SIFT_FORM ? lq : 0) + (sr1.kind != SIFT_FORM ? rq : 0) >= heavy_min;
    }
  }
  todo(live && fl != 1, heavy);
  if (live) flag[task] = (uint8_t)fl;
}

}  // namespace

hipError_t launch_ext_sift_kernel(const uint32_t* d_wire, int n_tasks, int16_t* d_out, const ExtScoring& sc, int dm, int qmax,
                                  uint8_t* d_flag, uint4* d_recs, hipStream_t s, KernelEvents kev, const ExtPrepass* d_pre_check,
                                  int* d_todo_count, int* d_todo_list, int heavy_min) {
  if (n_tasks <= 0) return hipSuccess;
  const int blocks = (n_tasks + 63) / 64;
  if (sc.pac) BPSW_LAUNCH(kev, ext_sift_kernel<true>, dim3(blocks), dim3(64), 0, s, d_wire, n_tasks, d_out, sc, dm, qmax, d_flag, d_recs, d_pre_check, d_todo_count, d_todo_list, heavy_min);
  else BPSW_LAUNCH(kev, ext_sift_kernel<false>, dim3(blocks), dim3(64), 0, s, d_wire, n_tasks, d_out, sc, dm, qmax, d_flag, d_recs, d_pre_check, d_todo_count, d_todo_list, heavy_min);
  return hipGetLastError();
}

}  // namespace bpsw
