// bpsw_extend_quad.hip -- the DP of the extension flanks that no exact shortcut resolves, FOUR flanks per 64-lane wavefront.
//
// What it computes: SWExtend (SWUtil.scala:61-230) under the band-retry / clip logic of extension()
// (MemChainToAlignBatched.scala:789-883) for the tasks ext_kernel (bpsw_extend.hip) hands over: ext_kernel evaluates the exact
// shortcuts of every flank (closed forms, certificates: bpsw_extend_core.h) and, at the first flank of a task that needs the DP,
// appends the task to a list together with what it has computed so far (the result of a left flank it resolved).  This kernel
// takes the list: it runs the remaining flank(s) of each task and writes the task's result record.  Bit-exact by construction --
// the same recurrences, the same order of evaluation, the same trimming as the one-task-per-wave sweeps.
//
// Why: a DP row of a flank touches ~45 cells, and in the one-task-per-wave sweeps it costs 110-140 instructions -- the scans
// over 64 lanes, and above all the row-synchronous control (band clamp, break tests, z-drop, trimming: SWUtil.scala:140-214).
// Here each 16-lane DPP row of the wave is a GROUP with a task of its own:
//   * lane l of a group owns S consecutive columns (S = 4: flanks up to 64 bases, S = 8: up to 128), right-aligned so that the
//     last query column is the last column of its lane;
//   * all per-task control state (band, maxima, phase, side, try) is per-lane data, identical inside a group: one vector
//     instruction advances the control of four tasks, and there is no scalar control flow in a row at all;
//   * F needs ONE exclusive max-plus scan per row (4 DPP steps inside the 16-lane row), after the lane has folded its S columns;
//     the row maximum and its LAST arg-max (SWUtil.scala:158-161) ride on a row_ror all-reduce of (a << 7 | column); the trimming
//     (SWUtil.scala:202-214) is one more pair of all-reduces;
//   * groups advance independently: a group that finishes a flank sets up its next one (or takes the next task from the queue)
//     while the others keep sweeping rows.
#include <stdlib.h>

#include "bpsw_extend_core.h"

namespace bpsw {
namespace {

constexpr int QD_WAVES_PER_BLOCK = 4;
#ifndef QD_CHUNK_DEF
#define QD_CHUNK_DEF 16
#endif
constexpr int QD_CHUNK = QD_CHUNK_DEF;  // tasks a wave takes per dequeue: with a third of them flagged, one or two per group
constexpr int QD_TS_CAP = 400;            // target rows staged per group: a flank's rows are at most qLen + 2*w + 2 <= 128 + 254 + 2
constexpr int NOZ = -(1 << 20);           // "no zero in this column" marker of the trimming reductions
constexpr int QD_NEG_A = -(1 << 20);      // "no cell": below every score, and << 7 does not overflow

__device__ __forceinline__ int qlo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int qhi16(uint32_t v) { return (int)(int16_t)(v >> 16); }

// lane l of every 16-lane row <- lane l-1 of the same row; lane 0 of the row keeps `old`
__device__ __forceinline__ int row_shr1(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x111, 0xf, 0xf, false); }

// inclusive max-scan of g inside each 16-lane row, and max all-reduce of k inside each row (interleaved so that every DPP read
// happens two wait states after the write of its operand: the hazard hipcc does not see inside an asm statement)
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

// Leader lanes named by `mask` each take one ticket from the counter (one returning atomic per lane); a single asm statement so
// that the compiler sees no lane-dependent branch (see dequeue_task in bpsw_extend_core.h).
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

enum { QP_NEED_TASK = 0, QP_NEED_CALL = 1, QP_ROWS = 2, QP_DONE = 3 };

#ifndef BPSW_QUAD_WAVES_PER_SIMD
#define BPSW_QUAD_WAVES_PER_SIMD 4
#endif

// qflag: one byte per task of the batch, written by ext_kernel: 0 = not handed over, 1 + side = the first flank that needs the DP
// (a plain store per task: a shared list would cost one returning atomic on one word per task, and that word sustains only ~90 of
// them per microsecond chip-wide).  The waves of this kernel take the batch in chunks of QD_CHUNK tasks (one atomic per chunk and
// wave), read their flags with one load, put them back to zero for the next batch, and hand the flagged tasks to their groups.
// carry: per task (indexed by the task number), for tasks with side == 1: what extension() has computed when its left flank is
// done -- x = regScore (low 16) | outQBeg (high 16), y = outRBeg | trueScore << 16, z = the band the left flank tried last.
template <int S, bool COORD>
__global__ __launch_bounds__(64 * QD_WAVES_PER_BLOCK, BPSW_QUAD_WAVES_PER_SIMD) void ext_quad_kernel(
    const uint32_t* __restrict__ wire, uint8_t* __restrict__ qflag, const uint4* __restrict__ carry, int16_t* __restrict__ out,
    const ExtScoring sc, int* __restrict__ next_task, const unsigned wire_words, const int n_tasks) {
  __shared__ __align__(16) uint8_t ts_all[QD_WAVES_PER_BLOCK][4][QD_TS_CAP];
  __shared__ int prof_all[QD_WAVES_PER_BLOCK][16];
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const int grp = lane >> 4, l = lane & 15;
  const int pos0 = l * S;  // the lane's first column, counted from the window's origin
  uint8_t* ts = ts_all[wave][grp];
  int* prof = prof_all[wave];
  if (lane < 5) {  // profile words per query base: bytes 0..3 = scores against target A,C,G,T; [8 + c] = against a target N
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
  const int amax = sc.tail_bound ? sc.mat_max : 0;  // rows past the query end that cannot matter (tail_row_bound)
  // F(i,j) = max_{k<j} (a(k) + k*e - oe) - (j-1)*e is invariant under a shift of the column origin, so the per-lane constants
  // count columns from the window's origin: g(pos) = a + pos*e - oe, F(pos) = P - (pos-1)*e.  With xbase = (pos0-1)*e the
  // lane-local form is gx(c) = g - xbase = a + (c+1)*e - oe  (no per-lane term at all) and F(c) = (P - xbase) - c*e.
  const int xbase = (pos0 - 1) * eIns;
  const int kG = eIns - oeIns;

  // ---- per-group state (identical in the 16 lanes of a group) ----
  int phase = QP_NEED_TASK, task = 0;
  int side = 0, tryi = 0, regScore = 0, prev = 0, sc0 = 0, awCur = wBand, awMax = wBand;
  int outQBeg = 0, outRBeg = 0, outQEnd = 0, outREnd = 0, trueScore = 0, score = -1;
  int qBegT = 0, rqT = 0;
  uint32_t r0 = 0, r1 = 0, r2 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;  // the task record, fetched once per task
  long long seedRb = 0;
  int seedLen = 0;
  int qLen = 0, tLen = 1, w = 0, hInit = 0, penClip = 0, base = 0, i_tail = 0x7fffffff;
  int i = 0, beg = 0, end = 0, mx = 0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, h1raw = 0;
  int addr_q = 0;  // ds_bpermute address of the lane that owns the last query column (its last column)
  int Hs[S], Es[S], plo[S];
  unsigned phiN[(S + 3) / 4];  // scores against a target N, one byte per column
#pragma unroll
  for (int c = 0; c < S; ++c) { Hs[c] = 0; Es[c] = 0; plo[c] = 0; }
#pragma unroll
  for (int k = 0; k < (S + 3) / 4; ++k) phiN[k] = 0;

  // the wave's chunk of the batch: QD_CHUNK consecutive tasks, their flags (one per lane) and which of them are still to be handed out
  int chunk_base = 0, chunk_flags = 0;
  unsigned long long avail = 0ull;
  bool exhausted = false;
  for (;;) {
    // ------------------------------------------------------------------ groups without a task take one
    if (any_lane(phase == QP_NEED_TASK)) {
      const unsigned long long needy = __builtin_amdgcn_ballot_w64(phase == QP_NEED_TASK);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (!((needy >> (16 * g)) & 1ull)) continue;  // wave-uniform: group g has a task
        while (avail == 0ull && !exhausted) {  // the next chunk of 64 tasks with at least one flagged task
          chunk_base = QD_CHUNK * dequeue_task(next_task);
          if (chunk_base >= n_tasks) {
            exhausted = true;
          } else {
            const int t = chunk_base + lane;
            chunk_flags = (lane < QD_CHUNK && t < n_tasks) ? (int)qflag[t] : 0;
            if (chunk_flags) qflag[t] = 0;  // the flags are zero again when the next batch arrives
            avail = __builtin_amdgcn_ballot_w64(chunk_flags != 0);
          }
        }
        const bool mine = grp == g;
        if (avail == 0ull) {
          if (mine) phase = QP_DONE;
          continue;
        }
        const int b = (int)__builtin_ctzll(avail);
        avail &= avail - 1ull;
        const int tk = chunk_base + b;
        const int sd = __builtin_amdgcn_readlane(chunk_flags, b) - 1;
        if (mine) {
          task = tk;
          side = sd;
          const uint32_t* rec = wire + 8 + (COORD ? 10 : 8) * (size_t)task;  // MemChainToAlignBatched.scala:95-117
          const uint4 ra4 = *reinterpret_cast<const uint4*>(rec), rb4 = *reinterpret_cast<const uint4*>(rec + 4);
          r0 = ra4.x; r1 = ra4.y; r2 = ra4.z; r4 = rb4.x; r5 = rb4.y; r6 = rb4.z; r7 = rb4.w;
          const uint32_t r3 = ra4.w;
          if (COORD) {
            seedRb = (long long)(((unsigned long long)rec[9] << 32) | rec[8]);
            seedLen = qhi16(r4);
          }
          rqT = qlo16(r1);
          regScore = qlo16(r3);
          qBegT = qhi16(r3);
          // extension() defaults, MemChainToAlignBatched.scala:790-807
          awMax = wBand;
          outQBeg = 0; outRBeg = 0; outQEnd = rqT; outREnd = 0; trueScore = regScore; score = -1;
          if (side == 1) {  // the left flank is done (ext_kernel resolved it): continue from its result
            const uint4 cw = carry[task];
            regScore = qlo16(cw.x); outQBeg = qhi16(cw.x);
            outRBeg = qlo16(cw.y); trueScore = qhi16(cw.y);
            awMax = (int)cw.z;
            score = regScore;
          }
          tryi = 0;
          phase = QP_NEED_CALL;
        }
      }
    }
    // ------------------------------------------------------------------ set up the next SWExtend call of a task
    if (any_lane(phase == QP_NEED_CALL)) {
      const bool need = phase == QP_NEED_CALL;
      const int lq = qlo16(r0), lr = qhi16(r0), rq = qlo16(r1), rr = qhi16(r1);
      if (need && side == 0 && lq <= 0) side = 1;  // MemChainToAlignBatched.scala:809 / :844
      if (need && side == 1 && rq <= 0) side = 2;
      if (need && side == 2) {  // task complete: 10 int16, MemChainToAlignBatched.scala:181-188, :877-879
        if (l == 0) {
          uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)sc.out_stride * (size_t)task);
          o[0] = r7;
          o[1] = ((uint32_t)outQBeg & 0xffffu) | ((uint32_t)outQEnd << 16);
          o[2] = ((uint32_t)outRBeg & 0xffffu) | ((uint32_t)outREnd << 16);
          o[3] = ((uint32_t)score & 0xffffu) | ((uint32_t)trueScore << 16);
          o[4] = (uint32_t)awMax & 0xffffu;
        }
        phase = QP_NEED_TASK;
      }
      const bool setup = need && side < 2;
      if (any_lane(setup)) {
        const uint32_t* words = wire + (size_t)(setup ? (int)r2 : 0);
        const int sq = side ? rq : lq, sr = side ? rr : lr;
        const int qStart = side ? lq : 0, rStart = side ? lq + rq + lr : lq + rq;
        const int maxIns = max(1, side ? qlo16(r6) : qlo16(r5)), maxDel = max(1, side ? qhi16(r6) : qhi16(r5));  // SWUtil.scala:110-115
        int tstage = 0;
        if (setup) {
          qLen = sq;
          penClip = side ? penClip3 : penClip5;
          if (tryi == 0) sc0 = regScore;               // MemChainToAlignBatched.scala:847
          hInit = side ? sc0 : qlo16(r4);              // left: h0; right: the score after the left extension
          prev = regScore;
          awCur = wBand << tryi;
          w = min(min(awCur, maxIns), maxDel);
          // Row i needs i - w <= qLen, so at most qLen + w + 1 rows of a flank are ever swept (the row at i = qLen + w has an
          // empty band and ends the call): only those are staged
          tLen = min(sr, min(qLen + w + 2, QD_TS_CAP));
          tstage = tLen;
          base = -((S - (qLen % S)) % S);  // right-aligned: column qLen-1 is the last column of its lane
          addr_q = ((lane & 48) + ((qLen - base) / S - 1)) << 2;
          i_tail = amax > 0 ? qLen : 0x7fffffff;
          const int col0 = base + pos0;
          int codes[S];
#pragma unroll
          for (int c = 0; c < S; ++c) codes[c] = nibble_at(words, qStart + min(max(col0 + c, 0), sq - 1));
#pragma unroll
          for (int k = 0; k < (S + 3) / 4; ++k) phiN[k] = 0;
#pragma unroll
          for (int c = 0; c < S; ++c) {
            const int col = col0 + c;
            const int code = (col >= 0 && col < sq) ? codes[c] : 4;
            plo[c] = prof[code];
            phiN[c >> 2] |= (unsigned)(prof[8 + code] & 0xff) << (8 * (c & 3));
            Hs[c] = col == 0 ? hInit : max(0, hInit - oeIns - (col - 1) * eIns);  // row -1, SWUtil.scala:97-104
            Es[c] = 0;
          }
          i = 0; beg = 0; end = sq;
          mx = hInit; max_i = -1; max_j = -1; max_ie = -1; gscore = -1; max_off = 0;  // SWUtil.scala:118-125
          h1raw = hInit - oDel;
          phase = QP_ROWS;
          if (sc.side_how && l == 0) sc.side_how[2 * (size_t)task + side] = 2;  // diagnostics only
        }
        // stage the target as 8*code bytes (the shift fed to v_bfe_i32)
        if constexpr (COORD) {
          const PacT tpac = {sc.pac, sc.l_pac, side ? seedRb + seedLen : seedRb - 1, side ? 1 : -1};
          for (int k = l; any_lane(setup && k < tstage); k += 16)
            if (setup && k < tstage) ts[k] = (uint8_t)(8 * tpac(k));
        } else {
          // 8 bases (two nibble words) per lane and pass
          for (int k = 8 * l; any_lane(setup && k < tstage); k += 128)
            if (setup && k < tstage) {
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
              *reinterpret_cast<uint2*>(ts + k) = make_uint2(w0, w1);  // bytes past tstage are never read (k + 8 <= QD_TS_CAP)
            }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (!any_lane(phase != QP_DONE)) break;
    // a call without a single target row (SWExtend's loop does not run): finish it with the initial state
    const bool rows = phase == QP_ROWS;
    bool fin = rows && tLen <= 0;

    if (any_lane(rows && !fin)) {
      // ---------------------------------------------------------------- one DP row for every group in QP_ROWS
      // (groups that are done run along on whatever their registers hold: nothing of theirs is read again)
      const int tsv = ts[i];
      h1raw -= eDel;
      const int h1 = max(0, h1raw);            // SWUtil.scala:137-138
      beg = max(beg, i - w);                   // SWUtil.scala:140-142
      end = min(min(end, i + w + 1), qLen);
      const int span = end - beg;
      const unsigned spanA = (unsigned)max(span, 0);  // columns beg <= col < end
      const int relbase = base + pos0 - beg;

      int sc_c[S];
#pragma unroll
      for (int c = 0; c < S; ++c) sc_c[c] = __builtin_amdgcn_sbfe(plo[c], (unsigned)tsv, 8u);
      if (any_lane(tsv == 32)) {  // a target N somewhere in the wave (rare): those groups score against the N row
        const bool isN = tsv == 32;
#pragma unroll
        for (int c = 0; c < S; ++c) sc_c[c] = isN ? __builtin_amdgcn_sbfe((int)phiN[c >> 2], 8u * (c & 3), 8u) : sc_c[c];
      }

      int a[S], gx[S];
      bool act[S];
      int G = NEG, K = -(1 << 30);
#pragma unroll
      for (int c = 0; c < S; ++c) {  // pass 1: a(col), the lane's best g and best (a, col) key
        act[c] = (unsigned)(relbase + c) < spanA;
        const int v = max(Hs[c] + sc_c[c], Es[c]);  // >= 0: E never goes below 0
        a[c] = act[c] ? v : QD_NEG_A;
        gx[c] = a[c] + (kG + c * eIns);
        G = max(G, gx[c]);
        K = max(K, (a[c] << 7) + c);
      }
      G += xbase;
      K += pos0;                       // key = a << 7 | position in the window (<= 127)
      row_scan_and_allreduce(G, K);    // G: inclusive prefix max over the lanes of the group; K: row maximum in every lane
      int Pm = row_shr1(NEG, G) - xbase;  // exclusive prefix, in the lane's own frame
      const int m = max(K >> 7, 0);       // row maximum (0 for an empty band)
      const int mj = base + (K & 127);    // LAST column whose a == m (SWUtil.scala:158-161); meaningful when m > 0

      unsigned Lu = 0u, Ru = 0xffffffffu;  // max of negative / min of positive (zero column - mj), as unsigned
      int H[S];
      const int cm = base + pos0 - mj;
#pragma unroll
      for (int c = 0; c < S; ++c) {  // pass 2: F, H, E for the lane's columns, left to right
        const int Hraw = max3i(a[c], Pm - c * eIns, 0);  // F(i,col) = max(0, P - (col-1)*eIns)
        Pm = max(Pm, gx[c]);
        // a column outside the band holds h1: the one left of `beg` hands it to column beg as eh[beg].h (SWUtil.scala:153);
        // the others are never read before the band has rewritten them (see sw_extend_il2)
        H[c] = act[c] ? Hraw : h1;
        const int En = max3i(Es[c] - eDel, Hraw - oeDel, 0);  // E(i+1,col)
        Es[c] = act[c] ? En : 0;                               // eh[end].e = 0
        const int t = (act[c] && Hraw == 0) ? cm + c : NOZ;    // zero of H inside the band, relative to mj
        Lu = max(Lu, (unsigned)t);
        Ru = min(Ru, (unsigned)t);
      }
#pragma unroll
      for (int c = S - 1; c > 0; --c) Hs[c] = H[c - 1];  // eh[col].h = H(i,col-1)
      Hs[0] = row_shr1(h1, H[S - 1]);                    // from the left neighbour; the window's first column takes h1

      // SWUtil.scala:177-182: j after the column loop is end (or beg for an empty band); h1 there is eh[end].h = H(i, qLen-1)
      const bool at_qend = rows && (span > 0 ? end : beg) == qLen;
      if (any_lane(at_qend)) {
        int hlast = __builtin_amdgcn_ds_bpermute(addr_q, H[S - 1]);
        hlast = span > 0 ? hlast : h1;
        const bool better = at_qend && gscore <= hlast;
        max_ie = better ? i : max_ie;
        gscore = better ? hlast : gscore;
      }

      bool brk = m == 0;  // SWUtil.scala:184-185
      const bool improved = m > mx;
      if (zdrop > 0 && any_lane(rows && !brk && !improved)) {  // SWUtil.scala:194-199 (Scala) / native/ksw.c:455-461 (BWA)
        const bool stop = zdrop_stop((i - max_i) - (mj - max_j), mx - m, eDel, eIns, zdrop, zmode);
        brk = brk || (!improved && stop);
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
      {
        const int lrel = (int)Lu, rrel = (int)Ru;
        beg = (lrel < 0 && lrel > NOZ / 2) ? mj + lrel + 2 : beg + (h1 == 0 ? 1 : 0);
        end = (rrel > 0) ? mj + rrel + 1 : end + 1;
      }
      // rows past the query end that cannot change the result (tail_row_bound): the reference would sweep them, none of them
      // moves max / gscore, so the call may end here -- checked for the NEXT row, as the one-task sweeps do at their loop top
      bool tstop = false;
      if (any_lane(rows && i + 1 >= i_tail)) {
        const int U = tail_row_bound(qLen, i + 1, hInit, amax, oDel, eDel);
        tstop = i + 1 >= i_tail && U <= mx && U < gscore;
      }
      const bool cont = rows && !brk;
      i += cont ? 1 : 0;
      fin = rows && (brk || tstop || i >= tLen);
    }
    // ------------------------------------------------------------------ a call ends: band retry / next side
    if (any_lane(fin)) {
      if (fin) {
        if (tLen <= 0) {  // no target row at all: SWExtend's loop did not run (SWUtil.scala:118-125 are the result)
          mx = hInit; max_i = -1; max_j = -1; max_ie = -1; gscore = -1; max_off = 0;
        }
        regScore = mx;
        const int qle = max_j + 1, tle = max_i + 1, gtle = max_ie + 1;  // SWUtil.scala:222-227
        bool again = tryi == 0 && !(regScore == prev || max_off < (awCur >> 1) + (awCur >> 2));  // C2AB:821,858
        if (again) {
          // the retry doubles the band; when the effective band min(w << 1, maxIns, maxDel) is the one just swept, the sweep
          // would repeat itself row by row: only the reported width changes
          const int maxIns = max(1, side ? qlo16(r6) : qlo16(r5)), maxDel = max(1, side ? qhi16(r6) : qhi16(r5));
          if (min(min(wBand << 1, maxIns), maxDel) == w) {
            awCur = wBand << 1;
            again = false;
          }
        }
        if (again) {
          tryi = 1;
        } else {
          score = regScore;
          awMax = max(awMax, awCur);
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
        phase = QP_NEED_CALL;
      }
    }
  }
  // The last wave to leave puts the queue back to zero for the next launch on this context: next_task[1] counts the waves that
  // have left (every group of a wave has seen the end of the list by then, so nobody takes another ticket).
  const int gone = dequeue_task(next_task + 1);
  if (gone == (int)gridDim.x * QD_WAVES_PER_BLOCK - 1 && lane == 0) {
    next_task[0] = 0;
    next_task[1] = 0;
  }
}

}  // namespace

// The flanks ext_kernel handed over (d_qflag: one byte per task, 1 + side), four per wavefront.  n_hint sizes the grid (the host
// does not know how many tasks are flagged; surplus waves leave at once).  d_counter: two device ints, zero between launches (the
// kernel's last wave puts them back); the kernel also puts the flags it consumes back to zero.
hipError_t launch_ext_quad_kernel(int s_cols, const uint32_t* d_wire, size_t wire_words, uint8_t* d_qflag, const uint4* d_carry,
                                  int n_hint, int n_tasks, int16_t* d_out, const ExtScoring& sc, int num_cu, int* d_counter, hipStream_t s,
                                  KernelEvents kev) {
  if (n_hint <= 0) return hipSuccess;
  const bool coord = sc.pac != nullptr;
  // groups per launch: each group runs several tasks back to back (dynamic queue), so a wave is not held by its longest task
  static const int tpg = getenv("BPSW_QUAD_TPG") ? atoi(getenv("BPSW_QUAD_TPG")) : 2;
  static const double blocks_per_cu = getenv("BPSW_QUAD_BLOCKS_PER_CU") ? atof(getenv("BPSW_QUAD_BLOCKS_PER_CU")) : 1.0;
  const int tasks_per_block = 4 * QD_WAVES_PER_BLOCK * (tpg > 0 ? tpg : 1);
  int blocks = (n_hint + tasks_per_block - 1) / tasks_per_block;
  const int max_blocks = (int)(num_cu * blocks_per_cu) > 0 ? (int)(num_cu * blocks_per_cu) : 1;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks < 1) blocks = 1;
#define BPSW_QUAD_GO(SC, CO)                                                                                              \
  BPSW_LAUNCH(kev, (ext_quad_kernel<SC, CO>), dim3(blocks), dim3(64 * QD_WAVES_PER_BLOCK), 0, s, d_wire, d_qflag, d_carry, d_out, sc, \
              d_counter, (unsigned)wire_words, n_tasks)
  if (s_cols == 4) {
    if (coord) BPSW_QUAD_GO(4, true); else BPSW_QUAD_GO(4, false);
  } else {
    if (coord) BPSW_QUAD_GO(8, true); else BPSW_QUAD_GO(8, false);
  }
#undef BPSW_QUAD_GO
  return hipGetLastError();
}

}  // namespace bpsw
