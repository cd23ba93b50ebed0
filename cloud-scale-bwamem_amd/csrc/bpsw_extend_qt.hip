// bpsw_extend_qt.hip -- "quad-task" banded extension: FOUR extension tasks per 64-lane wavefront, gfx950.
//
// Same computation as ext_kernel (bpsw_extend.hip): extension() of MemChainToAlignBatched.scala:789-883 over
// SWExtend (SWUtil.scala:61-230), bit-exact.  Why a second kernel: measured on MI355X both SW kernels issue about
// one instruction per ~2.8 cycles per SIMD whatever its type, and a typical extension row touches only ~40 DP
// cells, so the per-row control (band clamp, break tests, z-drop, band trimming) dominated the instruction count.
// Here every 16-lane DPP row of the wave runs its own task, so each control instruction serves four tasks:
//   * lane l of a group owns the S consecutive query columns [l*S, l*S+S)  (S = 4: sides <= 63 bp, S = 9: <= 143 bp);
//   * F needs ONE exclusive max-plus scan per row: the lane aggregate max_c g(col) goes through a 4-step
//     row_shr DPP scan, then each lane walks its S columns sequentially;
//   * the row maximum and its LAST arg-max come from one row_ror all-reduce of key = a<<8 | col;
//   * band trimming (SWUtil.scala:202-214) is two more all-reduces: unsigned max / min of (zero column - mj);
//   * all per-task state (band, maxima, phase, side, try) is per-lane data, identical inside a group, so the four
//     groups advance through rows, band retries, sides and tasks independently; a group that finishes a task pulls
//     the next one from the shared counter.
// Tasks whose sides are too long for S = 9 (or whose reference flank exceeds the LDS window) stay on ext_kernel.
#include <stdlib.h>

#include "bpsw_internal.h"
#include "bpsw_wave.h"

namespace bpsw {
namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int QT_TS_CAP = 448;  // reference bases staged per group (multiple of 8; eligibility: max(lr, rr) <= QT_TS_CAP)
constexpr int QT_CHUNK = 1;      // tickets per dequeue and group (3 measured 2x slower: single-task granularity is what balances the waves)
constexpr int NOZ = -(1 << 20); // "no zero in this column" marker for the trimming reductions

__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ int nib(const uint32_t* __restrict__ words, int k) {
  const uint32_t w = words[k >> 3];
  const int c = (int)((w >> (28 - 4 * (k & 7))) & 0xFu);
  return c > 4 ? 4 : c;
}
__device__ __forceinline__ int lo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t v) { return (int)(int16_t)(v >> 16); }

// lane l of every 16-lane row <- lane l-1 of the same row; lane 0 of the row keeps `old`
__device__ __forceinline__ int row_shr1(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, 0x111, 0xf, 0xf, false);
}
// inclusive max-scan of g inside each 16-lane row, and max all-reduce of k inside each row (interleaved so every
// DPP read happens two wait states after the write of its operand)
__device__ __forceinline__ void row_scan_and_allreduce(int& g, int& k) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(g), "+v"(k));
}
// unsigned max all-reduce of a and unsigned min all-reduce of b inside each 16-lane row
__device__ __forceinline__ void row_allreduce_umax_umin(unsigned& a, unsigned& b) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_u32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_u32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_u32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_u32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(a), "+v"(b));
}
__device__ __forceinline__ int row_allreduce_max(int v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}

// Leader lanes named by `mask` each take one ticket from the counter (one returning atomic per lane); a single asm
// statement so the compiler sees no lane-dependent branch (see dequeue_task in bpsw_extend.hip).
__device__ __forceinline__ int dequeue_lanes(int* counter, unsigned long long mask) {
  int v = 1;
  unsigned long long saved;
  asm volatile(
      "s_mov_b64 %1, exec\n\t"
      "s_mov_b64 exec, %3\n\t"
      "global_atomic_add %0, %2, %0, off sc0\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_mov_b64 exec, %1"
      : "+v"(v), "=&s"(saved)
      : "v"(counter), "s"(mask)
      : "memory");
  return v;
}

enum { PH_NEED_TASK = 0, PH_NEED_CALL = 1, PH_ROWS = 2, PH_DONE = 3 };

template <int S>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void ext_qt_kernel(const uint32_t* __restrict__ wire,
                                                                        const int* __restrict__ task_list, const int n_list,
                                                                        int16_t* __restrict__ out, const ExtScoring sc,
                                                                        int* __restrict__ next_task,
                                                                        const unsigned wire_words) {
  __shared__ __align__(16) uint8_t ts_all[WAVES_PER_BLOCK][4][QT_TS_CAP];
  __shared__ int prof_all[WAVES_PER_BLOCK][16];
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const int grp = lane >> 4, l = lane & 15;
  const int colbase = l * S;
  uint8_t* ts = ts_all[wave][grp];
  int* prof = prof_all[wave];
  if (lane < 5) {  // profile words per query base: bytes 0..3 = scores against target A,C,G,T; [8+c] = against N
    const int sh = 8 * lane;
    prof[lane] = (int)(((sc.mat.row[0] >> sh) & 0xff) | (((sc.mat.row[1] >> sh) & 0xff) << 8) |
                       (((sc.mat.row[2] >> sh) & 0xff) << 16) | (((sc.mat.row[3] >> sh) & 0xff) << 24));
    prof[8 + lane] = (int)(int8_t)((sc.mat.row[4] >> sh) & 0xff);
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
  const int gbase = colbase * eIns - oeIns;  // g(col) = a + col*eIns - oeIns = a + gbase + c*eIns
  const int xbase = (colbase - 1) * eIns;    // F(col) = max(0, P - (col-1)*eIns) = max(0, P - xbase - c*eIns)

  // ---- per-group state (identical in the 16 lanes of a group) ----
  int phase = PH_NEED_TASK, task = 0, ticket = 0, ticket_end = 0;
  int side = 0, tryi = 0, regScore = 0, prev = 0, sc0 = 0, aw0 = wBand, aw1 = wBand;
  int outQBeg = 0, outRBeg = 0, outQEnd = 0, outREnd = 0, trueScore = 0, score = -1;
  int qBegT = 0, rqT = 0;
  uint32_t r0 = 0, r1 = 0, r2 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;  // the task record, fetched once per task
  int qLen = 0, tLen = 0, w = 0, hInit = 0, penClip = 0, awCur = 0;
  int i = 0, beg = 0, end = 0, mx = 0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, h1raw = 0;
  int Hs[S], Es[S], plo[S], phi[S];
#pragma unroll
  for (int c = 0; c < S; ++c) { Hs[c] = 0; Es[c] = 0; plo[c] = 0; phi[c] = 0; }

  for (;;) {
    // ------------------------------------------------------------------ groups without a task take one
    if (any_lane(phase == PH_NEED_TASK)) {
      // tickets are taken QT_CHUNK at a time: one counter word only sustains ~90 returning atomics per microsecond
      // chip-wide (MI355X_MICROARCH.md "dequeue"), which one atomic per task would approach at 30 k tasks per batch
      const bool refill = phase == PH_NEED_TASK && ticket == ticket_end;
      if (any_lane(refill)) {
        const unsigned long long leaders = __builtin_amdgcn_ballot_w64(refill && l == 0);
        int t0 = dequeue_lanes(next_task, leaders);
        t0 = __builtin_amdgcn_ds_swizzle(t0, 0x0010);  // broadcast lane 0 of each 16-lane row to the row
        if (refill) {
          ticket = t0 * QT_CHUNK;
          ticket_end = ticket + QT_CHUNK;
        }
      }
      if (phase == PH_NEED_TASK) {
        if (ticket >= n_list) {
          phase = PH_DONE;
        } else {
          task = task_list ? task_list[ticket] : ticket;
          ticket += 1;
          const uint32_t* rec = wire + 8 + 8 * (size_t)task;  // MemChainToAlignBatched.scala:95-117
          const uint4 ra4 = *reinterpret_cast<const uint4*>(rec), rb4 = *reinterpret_cast<const uint4*>(rec + 4);
          r0 = ra4.x; r1 = ra4.y; r2 = ra4.z; r4 = rb4.x; r5 = rb4.y; r6 = rb4.z; r7 = rb4.w;
          const uint32_t r3 = ra4.w;
          rqT = lo16(r1);
          regScore = lo16(r3);
          qBegT = hi16(r3);
          // extension() defaults, MemChainToAlignBatched.scala:790-807
          aw0 = wBand; aw1 = wBand;
          outQBeg = 0; outRBeg = 0; outQEnd = rqT; outREnd = 0; trueScore = regScore; score = -1;
          side = 0; tryi = 0;
          phase = PH_NEED_CALL;
        }
      }
    }
    // ------------------------------------------------------------------ set up the next SWExtend call of a task
    if (any_lane(phase == PH_NEED_CALL)) {
      const bool need = phase == PH_NEED_CALL;
      const int lq = lo16(r0), lr = hi16(r0), rq = lo16(r1), rr = hi16(r1);
      if (need && side == 0 && lq <= 0) side = 1;  // MemChainToAlignBatched.scala:809 / :844
      if (need && side == 1 && rq <= 0) side = 2;
      if (need && side == 2) {  // task complete: 10 int16, MemChainToAlignBatched.scala:181-188, :877-879
        if (l == 0) {
          uint32_t* o = reinterpret_cast<uint32_t*>(out + 10 * (size_t)task);
          const int width = aw0 > aw1 ? aw0 : aw1;
          o[0] = r7;
          o[1] = ((uint32_t)outQBeg & 0xffffu) | ((uint32_t)outQEnd << 16);
          o[2] = ((uint32_t)outRBeg & 0xffffu) | ((uint32_t)outREnd << 16);
          o[3] = ((uint32_t)score & 0xffffu) | ((uint32_t)trueScore << 16);
          o[4] = (uint32_t)width & 0xffffu;
        }
        phase = PH_NEED_TASK;
      }
      const bool setup = need && side < 2;
      if (any_lane(setup)) {
        const uint32_t* words = wire + (size_t)(setup ? (int)r2 : 0);
        const int sq = side ? rq : lq, sr = side ? rr : lr;
        const int qStart = side ? lq : 0, rStart = side ? lq + rq + lr : lq + rq;
        const int maxIns = max(1, side ? lo16(r6) : lo16(r5)), maxDel = max(1, side ? hi16(r6) : hi16(r5));  // SWUtil.scala:110-115
        if (setup) {
          qLen = sq; tLen = sr;
          penClip = side ? penClip3 : penClip5;
          if (tryi == 0) sc0 = regScore;               // MemChainToAlignBatched.scala:847
          hInit = side ? sc0 : lo16(r4);               // left: h0; right: the score after the left extension
          prev = regScore;
          awCur = wBand << tryi;
          if (side) aw1 = awCur; else aw0 = awCur;
          w = min(min(awCur, maxIns), maxDel);
          // query profile words and row -1 (SWUtil.scala:83-104) for this lane's S columns; the loads are
          // unconditional (index clamped into the side) so all S of them are in flight together
          int codes[S];
#pragma unroll
          for (int c = 0; c < S; ++c) codes[c] = nib(words, qStart + min(colbase + c, sq - 1));
#pragma unroll
          for (int c = 0; c < S; ++c) {
            const int col = colbase + c;
            const int code = col < sq ? codes[c] : 4;
            plo[c] = prof[code];
            phi[c] = prof[8 + code];
            Hs[c] = col == 0 ? hInit : max(0, hInit - oeIns - (col - 1) * eIns);
            Es[c] = 0;
          }
          i = 0; beg = 0; end = sq;
          mx = hInit; max_i = -1; max_j = -1; max_ie = -1; gscore = -1; max_off = 0;  // SWUtil.scala:118-125
          h1raw = hInit - oDel;
          phase = PH_ROWS;
        }
        // stage the target as 8*code bytes (the shift fed to v_bfe_i32): 8 bases (two nibble words) per lane and pass
        for (int k = 8 * l; any_lane(setup && k < sr); k += 128)
          if (setup && k < sr) {
            const unsigned b0 = (unsigned)(rStart + k);
            const unsigned wbase = (unsigned)(words - wire) + (b0 >> 3);
            const unsigned long long hi = wire[min(wbase, wire_words - 1u)], lo = wire[min(wbase + 1u, wire_words - 1u)];
            const unsigned long long both = (hi << 32) | lo;  // first base in the most significant nibble
            const int o = (int)(b0 & 7u);
            unsigned w0 = 0, w1 = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              int code = (int)((both >> (60 - 4 * (o + t))) & 0xFull);
              code = code > 4 ? 4 : code;
              if (t < 4) w0 |= (unsigned)(8 * code) << (8 * t);
              else w1 |= (unsigned)(8 * code) << (8 * (t - 4));
            }
            *reinterpret_cast<uint2*>(ts + k) = make_uint2(w0, w1);  // bytes past sr are never read (k + 8 <= QT_TS_CAP)
          }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (!any_lane(phase != PH_DONE)) break;
    if (!any_lane(phase == PH_ROWS)) continue;

    // ------------------------------------------------------------------ one DP row for every group in PH_ROWS
    const bool rows = phase == PH_ROWS;
    const bool ra = rows && i < tLen;  // groups that still have a target row to sweep
    const int tsv = ts[ra ? i : 0];
    const bool isN = tsv == 32;
    if (ra) {
      h1raw -= eDel;
      beg = max(beg, i - w);                 // SWUtil.scala:140-142
      end = min(min(end, i + w + 1), qLen);
    }
    const int h1 = max(0, h1raw);            // SWUtil.scala:137-138
    const int span = end - beg;
    const unsigned spanA = ra ? (unsigned)max(span, 0) : 0u;      // columns beg <= col <  end
    const unsigned spanU = ra ? (unsigned)max(span + 1, 0) : 0u;  // columns beg <= col <= end
    const int relbase = colbase - beg;

    int a[S];
    int G = NEG, K = -(1 << 30);
#pragma unroll
    for (int c = 0; c < S; ++c) {  // pass 1: a(col), the lane's best g and best (a, col) key
      const bool act = (unsigned)(relbase + c) < spanA;
      const int s = isN ? phi[c] : __builtin_amdgcn_sbfe(plo[c], (unsigned)tsv, 8u);
      a[c] = act ? max(Hs[c] + s, Es[c]) : -1;
      G = max(G, a[c] + gbase + c * eIns);
      K = max(K, (a[c] << 8) + c);
    }
    K += colbase;                    // key = a<<8 | col  (col <= 143; a inactive = -1 -> negative key)
    row_scan_and_allreduce(G, K);    // G: inclusive prefix max over lanes; K: row maximum in every lane
    int P = row_shr1(NEG, G);        // exclusive: everything left of this lane's first column
    const int m = max(K >> 8, 0);    // row maximum (0 for an empty band)
    const int mj = K & 0xff;         // LAST column whose a == m (SWUtil.scala:158-161); meaningful when m > 0

    unsigned Lu = 0u, Ru = 0xffffffffu;  // max of negative / min of positive (zero column - mj), as unsigned
    int Hprev = 0;
#pragma unroll
    for (int c = 0; c < S; ++c) {  // pass 2: F, H, E for the lane's columns, left to right
      const unsigned rel = (unsigned)(relbase + c);
      const bool act = rel < spanA, upd = rel < spanU;
      const int H = max3i(a[c], P - xbase - c * eIns, 0);  // F(i,col) = max(0, P - (col-1)*eIns)
      P = max(P, a[c] + gbase + c * eIns);
      const int En = act ? max3i(Es[c] - eDel, H - oeDel, 0) : 0;  // E(i+1,col); eh[end].e = 0
      if (c > 0) {
        const int hsh = rel == 0u ? h1 : Hprev;  // eh[col].h = H(i,col-1); eh[beg].h = h1 (SWUtil.scala:153)
        Hs[c] = upd ? hsh : Hs[c];
      }
      Es[c] = upd ? En : Es[c];
      Hprev = H;
      const int t = (act && H == 0) ? colbase + c - mj : NOZ;  // zero of H inside the band, relative to mj
      Lu = max(Lu, (unsigned)t);
      Ru = min(Ru, (unsigned)t);
    }
    {  // column 0 of the lane takes H(i, col-1) from the left neighbour's last column
      const unsigned rel = (unsigned)relbase;
      const int hleft = row_shr1(0, Hprev);
      const int hsh = rel == 0u ? h1 : hleft;
      Hs[0] = rel < spanU ? hsh : Hs[0];
    }

    // SWUtil.scala:177-182: j after the column loop is end (or beg for an empty band); h1 there is eh[end].h
    const bool at_qend = ra && (span > 0 ? end : beg) == qLen;
    if (any_lane(at_qend)) {
      int mine = NEG;  // the lane that owns column `end` contributes eh[end].h
#pragma unroll
      for (int c = 0; c < S; ++c) mine = (colbase + c == end) ? Hs[c] : mine;
      int hlast = row_allreduce_max(mine);
      hlast = span > 0 ? hlast : h1;
      const bool better = at_qend && gscore <= hlast;
      max_ie = better ? i : max_ie;
      gscore = better ? hlast : gscore;
    }

    bool brk = ra && m == 0;  // SWUtil.scala:184-185
    const bool improved = ra && m > mx;
    if (zdrop > 0 && any_lane(ra && !brk && !improved)) {  // SWUtil.scala:194-199 (Scala) / native/ksw.c:455-461 (BWA)
      const int di = i - max_i, dj = mj - max_j;
      const bool A = di > dj;
      const bool B = mx - m - (di - dj) * eDel > zdrop;
      const bool C = mx - m - (dj - di) * eIns > zdrop;
      const bool stop = zmode == BPSW_ZDROP_SCALA ? (A && (B || C)) : (A ? B : C);
      brk = brk || (ra && !improved && m > 0 && stop);
    }
    {  // SWUtil.scala:187-193
      const int d = mj - i;
      const int off = max3i(max_off, d, -d);
      mx = improved ? m : mx;
      max_i = improved ? i : max_i;
      max_j = improved ? mj : max_j;
      max_off = improved ? off : max_off;
    }
    // band trimming, SWUtil.scala:202-214: last zero of H left of mj, first zero right of mj
    row_allreduce_umax_umin(Lu, Ru);
    const bool cont = ra && !brk;
    if (cont) {
      const int lrel = (int)Lu, rrel = (int)Ru;
      const int nb = (lrel < 0 && lrel > NOZ / 2) ? mj + lrel + 2 : beg + (h1 == 0 ? 1 : 0);
      const int ne = (rrel > 0) ? mj + rrel + 1 : end + 1;
      beg = nb;
      end = ne;
      i += 1;
    }
    // ------------------------------------------------------------------ a call ends: band retry / next side
    const bool fin = rows && (!ra || brk || i >= tLen);
    if (any_lane(fin)) {
      if (fin) {
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
            outQEnd = local ? qle : rqT;
            outREnd = local ? tle : gtle;
            trueScore += (local ? regScore : gscore) - sc0;
          }
          side += 1;
          tryi = 0;
        }
        phase = PH_NEED_CALL;
      }
    }
  }
}

// ---- binning: which kernel handles which task ------------------------------------------------------------------
__global__ void ext_bin_kernel(const uint32_t* __restrict__ wire, const int n_tasks, int* __restrict__ lists,
                               int* __restrict__ counts, const int max_qt_side) {
  const int lane = threadIdx.x & 63;
  const int nround = (n_tasks + 63) & ~63;  // every lane of a wave runs the same number of iterations (ballots below)
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nround; t += gridDim.x * blockDim.x) {
    int bin = -1;
    if (t < n_tasks) {
      const uint32_t* rec = wire + 8 + 8 * (size_t)t;
      const int lq = lo16(rec[0]), lr = hi16(rec[0]), rq = lo16(rec[1]), rr = hi16(rec[1]);
      const int mq = max(lq, rq), mr = max(lr, rr);
      bin = 2;
      if (mr <= QT_TS_CAP && mq <= max_qt_side) bin = mq <= 16 * 4 - 1 ? 0 : (mq <= 16 * 9 - 1 ? 1 : 2);
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {  // one atomic per wave and bin instead of one per task
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(bin == b);
      if (mask == 0ull) continue;
      int base = 0;
      if (lane == (int)__builtin_ctzll(mask)) base = atomicAdd(&counts[b], (int)__builtin_popcountll(mask));
      base = __builtin_amdgcn_readlane(base, (int)__builtin_ctzll(mask));
      if (bin == b) lists[(size_t)b * n_tasks + base + (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = t;
    }
  }
}

}  // namespace

void launch_ext_bin(const uint32_t* d_wire, int n_tasks, int* d_lists, int* d_counts, hipStream_t s) {
  const int threads = 256;
  int blocks = (n_tasks + threads - 1) / threads;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  // BPSW_EXT_QT=2: only sides <= 63 bp go to the quad-task kernel (S = 4); =1: also S = 9
  static const int max_side = (getenv("BPSW_EXT_QT") && atoi(getenv("BPSW_EXT_QT")) == 2) ? 63 : 143;
  hipLaunchKernelGGL(ext_bin_kernel, dim3(blocks), dim3(threads), 0, s, d_wire, n_tasks, d_lists, d_counts, max_side);
}

hipError_t launch_ext_qt_kernel(int s_cols, const uint32_t* d_wire, size_t wire_words, const int* d_list, int n_list,
                                int16_t* d_out, const ExtScoring& sc, int num_cu, int* d_counter, hipStream_t s) {
  if (n_list <= 0) return hipSuccess;
  // Each group should run several tasks back to back (dynamic queue) so a wave is not held by its longest task;
  // ~3 tasks per group also leaves 2-3 resident waves per SIMD at 32 k-read batches, enough to hide the DPP chains.
  static const int tpg = getenv("BPSW_QT_TPG") ? atoi(getenv("BPSW_QT_TPG")) : 3;
  const int tasks_per_block = 4 * WAVES_PER_BLOCK * (tpg > 0 ? tpg : 1);
  int blocks = (n_list + tasks_per_block - 1) / tasks_per_block;
  const int max_blocks = num_cu * 8;
  if (blocks > max_blocks) blocks = max_blocks;
  hipError_t me = hipMemsetAsync(d_counter, 0, sizeof(int), s);
  if (me != hipSuccess) return me;
  if (s_cols == 4)
    hipLaunchKernelGGL(ext_qt_kernel<4>, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, s, d_wire, d_list, n_list, d_out, sc, d_counter, (unsigned)wire_words);
  else
    hipLaunchKernelGGL(ext_qt_kernel<9>, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), 0, s, d_wire, d_list, n_list, d_out, sc, d_counter, (unsigned)wire_words);
  return hipGetLastError();
}

// QT kernels need oeIns > 0 (their row maximum is read off `a`); the header is the same for the whole batch, so the
// caller decides `use_qt` once.  Bin 2 (and everything when !use_qt) goes to the one-task-per-wave kernel.
// The three launches are independent (disjoint task lists and result rows), so they run on three streams and
// overlap their tails: fork from `s`, join back into `s`.
hipError_t launch_ext_all(const uint32_t* d_wire, size_t wire_words, int n_tasks, int16_t* d_out, const ExtScoring& sc,
                          int qcap, int rcap, int num_cu, int* d_counters, const int* d_lists, const int h_counts[3],
                          bool use_qt, const ExtStreams& aux, hipStream_t s) {
  if (!use_qt) return launch_ext_kernel(d_wire, n_tasks, d_out, sc, qcap, rcap, num_cu, d_counters, nullptr, s);
  static const bool serial = getenv("BPSW_QT_SERIAL") != nullptr;  // diagnostic: all launches on the caller's stream
  if (serial) {
    hipError_t e2 = launch_ext_qt_kernel(9, d_wire, wire_words, d_lists + (size_t)n_tasks, h_counts[1], d_out, sc, num_cu, d_counters + 1, s);
    if (e2 == hipSuccess) e2 = launch_ext_qt_kernel(4, d_wire, wire_words, d_lists, h_counts[0], d_out, sc, num_cu, d_counters, s);
    if (e2 == hipSuccess && h_counts[2] > 0)
      e2 = launch_ext_kernel(d_wire, h_counts[2], d_out, sc, qcap, rcap, num_cu, d_counters + 2, d_lists + 2 * (size_t)n_tasks, s);
    return e2;
  }
  hipError_t e = hipEventRecord(aux.fork, s);
  if (e != hipSuccess) return e;
  for (int k = 0; k < 2; ++k) {
    e = hipStreamWaitEvent(aux.stream[k], aux.fork, 0);
    if (e != hipSuccess) return e;
  }
  // longest work first on the caller's stream: the S = 9 bin; S = 4 and the fallback on the auxiliary streams
  e = launch_ext_qt_kernel(9, d_wire, wire_words, d_lists + (size_t)n_tasks, h_counts[1], d_out, sc, num_cu, d_counters + 1, s);
  if (e != hipSuccess) return e;
  e = launch_ext_qt_kernel(4, d_wire, wire_words, d_lists, h_counts[0], d_out, sc, num_cu, d_counters, aux.stream[0]);
  if (e != hipSuccess) return e;
  if (h_counts[2] > 0) {
    e = launch_ext_kernel(d_wire, h_counts[2], d_out, sc, qcap, rcap, num_cu, d_counters + 2, d_lists + 2 * (size_t)n_tasks,
                          aux.stream[1]);
    if (e != hipSuccess) return e;
  }
  for (int k = 0; k < 2; ++k) {
    e = hipEventRecord(aux.join[k], aux.stream[k]);
    if (e != hipSuccess) return e;
    e = hipStreamWaitEvent(s, aux.join[k], 0);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace bpsw
