// bpsw_extend.hip -- banded affine-gap seed extension for gfx950 (boundary 2).
//
// What it computes: for every task of a wire batch (MemChainToAlignBatched.scala:76-172) the result
// of the Scala extension() (MemChainToAlignBatched.scala:789-883): left then right SWExtend
// (SWUtil.scala:61-230) with up to MAX_BAND_TRY=2 band widths, bit-exact.
//
// How: one task per 64-lane wavefront, four independent waves per workgroup, no workgroup barrier, tasks pulled from a
// self-resetting work queue by persistent workgroups.  The DP is row-synchronous because the band [beg,end) of row i+1 is
// derived from the finished row i (SWUtil.scala:201-214).  A row lives in registers, one, two or four columns per lane on a window that
// follows the band (bpsw_extend_rows.h: sw_extend_adaptive, row loops in GCN assembly; the full kernel: the slot sweeps of
// bpsw_extend_core.h):
//   a(j)   = max(H(i-1,j-1) + S(i,j), E(i,j))                      per lane
//   F(i,j) = max(0, max_{k<j}(a(k) - oeIns - (j-1-k)*eIns))        wave max-plus prefix scan (DPP)
//   H(i,j) = max(a(j), F(i,j)),  E(i+1,j) = max(E(i,j)-eDel, H(i,j)-oeDel, 0)
// (valid because oIns >= 0, checked on the host).  Row maximum + LAST arg-max (SWUtil.scala:158-161) ride on a second scan; the
// band trimming loops (SWUtil.scala:202-214) are evaluated on 64-bit zero masks held in SGPRs, so all row control is scalar.
// Before any DP a flank goes through the exact shortcuts (closed forms, certificates: bpsw_extend_core.h), which resolve most
// flanks of low-error reads -- for large batches of short flanks in a kernel of their own in front of this one (bpsw_extend_sift.hip:
// one task per lane), which leaves a flag and a verdict per side here (sift_flag, sift_recs).  Two builds (ext_kernel<COORD, SHORT>):
// SHORT = 1, the 64-VGPR short kernel at eight waves per SIMD for flanks up to 255 bases (the adaptive sweep of bpsw_extend_rows.h: one,
// two or four columns per lane on a window that follows the band); SHORT = 0, the full kernel (slot sweeps, an LDS-row sweep for
// flanks above 255 bases) for what the host lists.  A task's body is ext_do_task, shared with ext_resident_kernel: the
// short kernel's resident form behind the device's extension ring (bpsw_ring.h), which takes the small batches without a launch.
#include <stdlib.h>

#include <atomic>

#include "bpsw_extend_core.h"
#include "bpsw_extend_rows.h"
#include "bpsw_ring_dev.h"

#include "bpsw_diag_waves.h"
BPSW_DIAG_WAVES_DEFINE(ext)

namespace bpsw {
namespace {

constexpr int WAVES_PER_BLOCK = 4;

// stage the target of one side in LDS as 8*code bytes (the shift the register path feeds to v_bfe)
template <class T>
__device__ void load_target_shifts(const int lane, const T& tsrc, const int rLen, uint8_t* __restrict__ ts) {
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < rLen; i += 64) ts[i] = (uint8_t)(8 * tsrc(i));
  __builtin_amdgcn_wave_barrier();
}

// Unpack one side of a task into LDS: the 5 x qLen query profile and the target bytes.
template <class T>
__device__ void load_side(const int lane, const uint32_t* __restrict__ words, const int qStart, const int qLen,
                          const T& tsrc, const int rLen, const MatRows& mat, int8_t* __restrict__ qp,
                          uint8_t* __restrict__ ts) {
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < qLen; j += 64) {
    const int c = nibble_at(words, qStart + j);
#pragma unroll
    for (int k = 0; k < 5; ++k) qp[k * qLen + j] = (int8_t)((mat.row[k] >> (8 * c)) & 0xff);
  }
  for (int i = lane; i < rLen; i += 64) ts[i] = (uint8_t)tsrc(i);
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int lo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t v) { return (int)(int16_t)(v >> 16); }

// One task of a batch: both sides' shortcuts and sweeps, the 10-int16 record (extension(), MemChainToAlignBatched.scala:789-883).  Shared
// by the launched kernels (ext_kernel) and the resident one (ext_resident_kernel: RING, the record is stored for a host that is
// looking at it while the kernel lives).  eh / qp: the full kernel's LDS rows; ts / pl: the target bytes and the query profile.
template <bool COORD, int SHORT, bool RING>
__device__ __forceinline__ void ext_do_task(const int task, const int sifted, const int lane, const uint32_t* __restrict__ wire,
                                            int16_t* __restrict__ out, const ExtScoring& sc, const MatRows& mat, const int oDel, const int eDel, const int oIns,
                                            const int eIns, const int penClip5, const int penClip3, const int wBand, const int exact_a,
                                            const int amax, int2* eh, int8_t* qp, uint8_t* ts, const ProfLds& pl, int* __restrict__ defer,
                                            const int short_qmax, const uint4* __restrict__ sift_recs, const int inject_defer) {
  const uint32_t* rec = wire + 8 + (COORD ? 10 : 8) * (size_t)task;  // MemChainToAlignBatched.scala:95-117
  const uint32_t r0 = rec[0], r1 = rec[1], r3 = rec[3], r4 = rec[4], r5 = rec[5], r6 = rec[6];
  // coordinate batch: the seed's start in the doubled reference and its length (in the slot of the redundant 16-bit idx)
  const long long seedRb = COORD ? (long long)(((unsigned long long)uni((int)rec[9]) << 32) | (unsigned)uni((int)rec[8])) : 0ll;
  const int seedLen = COORD ? uni(hi16(r4)) : 0;
  const int lq = uni(lo16(r0)), lr = uni(hi16(r0)), rq = uni(lo16(r1)), rr = uni(hi16(r1));
  // short_qmax < 0 (the asynchronous device entry, where no host has seen the records): nobody has listed the tasks this build
  // cannot take -- a flank above -short_qmax bases, or any task when the gap costs rule out the register sweeps -- so it defers
  // them itself; short_qmax > 0: the host has listed them for the full kernel
  bool deferred = false;
  // (test hook, BPSW_EXT_INJECT_DEFER=k: every k-th task of a deferring launch takes the way of a band that outgrew the window)
  if (SHORT && inject_defer > 0 && task % inject_defer == 0) deferred = true;
  if (SHORT) {
    const int qm = short_qmax < 0 ? -short_qmax : short_qmax;
    const bool too_long = lq > qm || rq > qm || (short_qmax < 0 && oIns + eIns <= 0);
    if (too_long) {
      if (short_qmax < 0) {
        if (lane == 0) defer[1 + atomicAdd(defer, 1)] = task;
      }
      return;
    }
  }
  const uint32_t* words = wire + (size_t)uni((int)rec[2]);
  const int regScore0 = uni(lo16(r3)), qBeg = uni(hi16(r3)), h0 = uni(lo16(r4));
  const int lMaxIns = max(1, uni(lo16(r5))), lMaxDel = max(1, uni(hi16(r5)));  // SWUtil.scala:110-115
  const int rMaxIns = max(1, uni(lo16(r6))), rMaxDel = max(1, uni(hi16(r6)));
  const int idx = uni((int)rec[7]);

  // extension(), MemChainToAlignBatched.scala:789-883: side 0 = left (penClip5), side 1 = right (penClip3)
  int awSide = wBand, awMax = wBand;  // the band tried last on this side / the widest over both sides (no array: a dynamically indexed one lives in scratch memory)
  int regScore = regScore0;
  int outQBeg = 0, outRBeg = 0, outQEnd = rq, outREnd = 0, trueScore = regScore0, score = -1;
  for (int side = 0; side < 2; ++side) {
    const int qLen = side ? rq : lq, rLen = side ? rr : lr;
    if (qLen <= 0) continue;
    // (SHORT: a side derives its per-lane values from an opaque copy of the lane number -- qStart + lane and the like, computed for
    // both sides at the top of the task, were VGPRs the 64-register build had to spill: bpsw_extend_rows.h, rows_opaque)
    const int lane_s = SHORT ? rows_opaque(lane) : lane;
    const int qStart = side ? lq : 0, rStart = side ? lq + rq + lr : lq + rq;
    const int maxIns = side ? rMaxIns : lMaxIns, maxDel = side ? rMaxDel : lMaxDel;
    const int penClip = side ? penClip3 : penClip5;
    const int hInit = uni(side ? regScore : h0);  // the right extension starts from the score after the left one (uni: see `exact` below)
    const int sc0 = regScore;
    // register path: needs one lane per column 0..qLen and oeIns > 0 (see sw_extend_reg)
    const bool reg_path = qLen <= 255 && oIns + eIns > 0;
    ExtRes r = {0, 0, 0, 0, 0, 0};
    const NibbleT tnib = {words, rStart};
    const PacT tpac = {sc.pac, sc.l_pac, side ? seedRb + seedLen : seedRb - 1, side ? 1 : -1};
    // Row i needs i - w <= qLen, so at most qLen + w + 1 rows of a side are ever swept (the row at i = qLen + w has an empty
    // band and ends the call): a coordinate batch stages only those, before the shortcuts, which then read LDS too.
    const int tstage = COORD ? min(rLen, qLen + (wBand << 1) + 2) : rLen;
    if (COORD && reg_path) load_target_shifts(lane_s, tpac, tstage, ts);
    // near-exact flank: the DP result is known (flank_closed_form); the retry loop would stop after its first try
    const auto shortcuts = [&](const auto& tsrc, const int tl) {
      return exact_a > 0 &&
             ((rLen >= qLen && tl >= qLen && flank_closed_form(lane_s, qLen, tl, NibbleQ{words, qStart}, tsrc, mat,
                                                              hInit, exact_a, oDel, eDel, oIns, eIns, sc.zdrop, sc.certify, &r)) ||
              (sc.certify >= 3 && flank_start_gap_form(lane_s, qLen, tl, NibbleQ{words, qStart}, tsrc, mat,
                                                       hInit, exact_a, oDel, eDel, oIns, eIns, sc.zdrop, wBand, &r)));
    };
    bool exact_v;
    int judged = 0;  // 1: the sift kernel found that no form holds, 2: that one does for this start score
    if (SHORT && sifted == 2) {
      const uint4 sr = sift_recs[2 * (size_t)task + side];
      const int kind = uni((int)(sr.x & 0xffu)), hmin = uni((int)sr.x >> 8);
      if (kind == 1) judged = 1;
      else if (kind == 2 && hInit >= hmin) {
        judged = 2;
        r.max = hInit + uni(lo16(sr.y)); r.gscore = hInit + uni(hi16(sr.y));
        r.qle = uni(lo16(sr.z)); r.tle = uni(hi16(sr.z)); r.gtle = uni(lo16(sr.w)); r.max_off = uni(hi16(sr.w));
      }
    }
    if (judged) exact_v = judged == 2;
    else if constexpr (COORD) exact_v = reg_path && shortcuts(LdsShiftT{ts}, tstage);
    else exact_v = shortcuts(tnib, rLen);
    // wave-uniform by construction (the shortcuts decide on wave reductions), but not to the compiler: without this the DP below
    // sits in what it takes for divergent control flow and its whole scalar state is kept in vector registers
    const bool exact = uni(exact_v ? 1 : 0) != 0;
    if (sc.side_how && lane == 0) sc.side_how[2 * (size_t)task + side] = exact ? 1 : 2;  // diagnostics only
    if (exact) {
      awSide = wBand;
      regScore = uni(r.max);
    } else if (reg_path) {
      if (!COORD) load_target_shifts(lane_s, tnib, rLen, ts);
    } else if constexpr (SHORT) {
      // (never: the host sends a batch whose gap costs rule the register path out to the full kernel)
    } else if constexpr (COORD) {
      load_side(lane, words, qStart, qLen, tpac, tstage, mat, qp, ts);
    } else {
      load_side(lane, words, qStart, qLen, tnib, rLen, mat, qp, ts);
    }
    for (int i = 0; i < 2 && !exact; ++i) {  // MAX_BAND_TRY
      const int prev = regScore;
      awSide = wBand << i;
      const int w = uni(min(min(awSide, maxIns), maxDel));
      // the retry doubles the band; when the EFFECTIVE band min(w << 1, maxIns, maxDel) is the one just swept, the sweep would
      // repeat itself row by row (SWUtil.scala:110-115: w is all of the band the call sees): only the reported width changes
      if (i == 1 && w == uni(min(min(wBand, maxIns), maxDel))) break;
      if constexpr (SHORT) {
        int oInsT = oIns, eInsT = eIns;  // opaque copies, as below
        asm volatile("" : "+s"(oInsT), "+s"(eInsT));
        int ov = 0;
#if BPSW_EXT_ADAPTIVE
        r = sw_extend_adaptive(lane_s, qLen, COORD ? min(rLen, qLen + w + 2) : rLen, NibbleQ{words, qStart}, ts, pl, mat, oDel, eDel, oInsT, eInsT, w, sc.zdrop, sc.zdrop_mode, hInit, amax, &ov);
#else
        r = sw_extend_reg_short<true>(lane, qLen, COORD ? min(rLen, qLen + w + 2) : rLen, NibbleQ{words, qStart}, ts, mat, oDel, eDel, oInsT, eInsT, w, sc.zdrop, sc.zdrop_mode, hInit, amax, &ov);
#endif
        if (uni(ov)) {  // (never since round 5: sw_extend_adaptive holds every band of a flank this kernel takes)
          deferred = true;
          break;
        }
      } else if (reg_path) {
        // opaque copies: otherwise the per-lane column constants of every slot count (j*eIns - oeIns, (j-1)*eIns) are hoisted
        // out of the task loop and sit in ~20 VGPRs for the whole kernel, which no longer fits five waves per SIMD
        int oInsT = oIns, eInsT = eIns;
        asm volatile("" : "+s"(oInsT), "+s"(eInsT));
        r = sw_extend_reg_any(lane, qLen, COORD ? min(rLen, qLen + w + 2) : rLen, NibbleQ{words, qStart}, ts, mat, oDel, eDel, oInsT, eInsT, w, sc.zdrop, sc.zdrop_mode, hInit, amax, eh);
      } else {
        r = sw_extend_wave(lane, qLen, COORD ? min(rLen, qLen + w + 2) : rLen, eh, qp, ts, oDel, eDel, oIns, eIns, w, sc.zdrop, sc.zdrop_mode, hInit, amax);
      }
      regScore = uni(r.max);
      if (regScore == prev || r.max_off < (awSide >> 1) + (awSide >> 2)) break;
    }
    if (SHORT && deferred) break;
    score = regScore;
    awMax = max(awMax, awSide);
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
  if (SHORT && deferred) {  // one atomic per deferred task (rare): its slot in the full kernel's list
    // (every deferring launch has a list: launch_ext_kernel refuses one without.  A batch whose flanks all have at most 127 bases
    // is not expected to defer anything -- their bands fit the 128-column window -- but if a row loop ever said otherwise, the
    // task goes to the full kernel like any other deferred one instead of trapping the executor's process: round 4 trapped)
    if (!defer) return;
    if (lane == 0) defer[1 + atomicAdd(defer, 1)] = task;
    return;
  }
  const int width = awMax;
  // MemChainToAlignBatched.scala:181-188: 10 int16 per task
  uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)sc.out_stride * (size_t)task);
  const uint32_t o1 = ((uint32_t)outQBeg & 0xffffu) | ((uint32_t)outQEnd << 16), o2 = ((uint32_t)outRBeg & 0xffffu) | ((uint32_t)outREnd << 16);
  const uint32_t o3 = ((uint32_t)score & 0xffffu) | ((uint32_t)trueScore << 16), o4 = (uint32_t)width & 0xffffu;
  if constexpr (RING) {
    // under the resident kernel nothing ends with a kernel's end: the record goes out with system-scope (write-through) stores, ONE
    // wave instruction for its five words (a word per lane; word by word from one lane they are five round trips to the point of
    // coherence: bpsw_swalign.hip, swp_do_duo, tells what fourteen cost)
    const uint32_t v = lane == 0 ? (uint32_t)idx : lane == 1 ? o1 : lane == 2 ? o2 : lane == 3 ? o3 : o4;
    if (lane < 5) __hip_atomic_store(o + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else if (lane == 0) {
    o[0] = (uint32_t)idx; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = o4;
  }
}

#ifndef BPSW_EXT_WAVES_PER_SIMD
// register budget.  Five waves per SIMD: 96 VGPRs and no scratch.  Six (80 VGPRs) ran the bench step at the same rate, but three
// dwords per lane spilled -- one scratch store per task, 1.4 MB of the launch's 2.4 MB of WRITE_SIZE (tools/pmc_traffic_quick.sh).
#define BPSW_EXT_WAVES_PER_SIMD 5
#endif
// COORD: a coordinate batch (include/bpsw.h, "wire format 2") -- 40-byte task records, query flanks only, the target flanks are
// read from the device-resident reference (bnsGetSeq of MemChainToAlignBatched.scala:363 moves to the device).
// SHORT = 1: the kernel for tasks whose two query flanks have at most `short_qmax` bases (255: nearly every task of 2x150 and 2x250 bp
// reads).  It carries only the adaptive sweep (bpsw_extend_rows.h: one / two columns per lane on a window that follows the band,
// row loops in assembly) and the shortcuts, needs 64 VGPRs instead of 87 and little LDS (the target bytes and the call's query
// profile), so eight of its waves share a SIMD.  It skips longer tasks: the host, which has seen every record (scan_wire), lists
// those for the full kernel.  Since round 5 no band of such a flank leaves the adaptive sweep (its four-columns-per-lane phase holds 256
// columns; rounds 3-4 deferred a band wider than 128 columns to the full kernel, or -- a second build, ext_kernel<., 2>, for 2x250 bp
// batches -- swept the side again with the slot sweep): a task is DEFERRED -- appended to the full kernel's list (defer[0] = entries so
// far, defer[1..] = task indices), which the full kernel, launched behind this one on the stream, reads its task count from -- only by
// the asynchronous device entry, where no host has listed the long tasks, and by the test hook below.
// SHORT = 0: the full kernel (slot sweeps for wide bands, an LDS-row sweep for flanks above 255 bases).
template <bool COORD, int SHORT>
#ifndef BPSW_EXT_SHORT_WAVES_PER_SIMD
#define BPSW_EXT_SHORT_WAVES_PER_SIMD 8
#endif
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, SHORT ? BPSW_EXT_SHORT_WAVES_PER_SIMD : BPSW_EXT_WAVES_PER_SIMD) void ext_kernel(const uint32_t* __restrict__ wire, const int n_tasks_arg,
                                                                     int16_t* __restrict__ out, const ExtScoring sc,
                                                                     const int qcap, const int rcap,
                                                                     const int lds_per_wave, const int chunk, const int guide_cap,
                                                                     int* __restrict__ next_task,
                                                                     const int* __restrict__ task_list_arg,
                                                                     const ExtPrepass* __restrict__ pre,
                                                                     int* __restrict__ defer, const int short_qmax,
                                                                     const uint8_t* __restrict__ sift_flag,
                                                                     const uint4* __restrict__ sift_recs,
                                                                     int* __restrict__ defer_post,
                                                                     const int* __restrict__ todo_list, const int inject_defer) {
  extern __shared__ __align__(16) unsigned char smem[];
  BPSW_DIAG_WAVE_BEGIN();
  BPSW_DIAG_TASKS_DECL();
  // the full kernel behind a SHORT launch: its task list and count are what the host listed plus what that launch deferred
  const int n_tasks = (!SHORT && defer) ? uni(defer[0]) : n_tasks_arg;
  const int* __restrict__ task_list = (!SHORT && defer) ? defer + 1 : task_list_arg;
  if (!SHORT && defer && n_tasks == 0) return;  // nothing was listed or deferred: the queue heads stay as they are (zero)
  // asynchronous entry (bpsw_extend_batch_device): the table scan ran just before on the same stream and nobody has read
  // it back yet -- a malformed batch, or one whose tasks outgrow the LDS this launch was sized for, is left untouched
  if (pre && (pre->error != 0 || pre->max_qlen > qcap || pre->max_rlen > rcap)) return;
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  unsigned char* base = smem + (size_t)wave * lds_per_wave;
  int2* eh = reinterpret_cast<int2*>(base);
  int8_t* qp = reinterpret_cast<int8_t*>(base + 8 * (size_t)(qcap + 2));
  uint8_t* ts = SHORT ? base : reinterpret_cast<uint8_t*>(qp + 5 * (size_t)qcap);  // SHORT: the target bytes, then the query profile of the call
  // (SHORT) the query profile of the adaptive sweep (bpsw_extend_rows.h, ProfLds): qcap + 2 words, qcap + 2 bytes behind the target bytes
  int* prof_w = reinterpret_cast<int*>(base + (((size_t)rcap + 15) & ~(size_t)15));
  const ProfLds pl = {prof_w, reinterpret_cast<int8_t*>(prof_w + qcap + 2),
                      (unsigned)(uintptr_t)((__attribute__((address_space(3))) int*)prof_w)};

  // header, MemChainToAlignBatched.scala:78-84 (signed bytes)
  const uint32_t hdr0 = wire[0], hdr1 = wire[1];
  const int oDel = (int8_t)(hdr0 & 0xff), eDel = (int8_t)((hdr0 >> 8) & 0xff);
  const int oIns = (int8_t)((hdr0 >> 16) & 0xff), eIns = (int8_t)((hdr0 >> 24) & 0xff);
  const int penClip5 = (int8_t)(hdr1 & 0xff), penClip3 = (int8_t)((hdr1 >> 8) & 0xff);
  const int wBand = (int8_t)((hdr1 >> 16) & 0xff);
  // closed form for near-exact flanks (bpsw_extend_core.h, flank_closed_form): usable with this batch's band?
  const int oe_min = min(oIns + eIns, oDel + eDel);
  const int exact_a = (oe_min > 0 && wBand >= 2) ? sc.exact_a : 0;
  const int amax = sc.tail_bound ? sc.mat_max : 0;  // rows past the query end that cannot matter (tail_row_bound)

  // Tasks differ in cost by an order of magnitude, so waves pull them from a shared counter instead of
  // striding: a wave takes the next task when it finishes one and leaves when the counter passes n_tasks.
  // One counter word sustains only ~90 returning atomics per microsecond chip-wide (MI355X_MICROARCH.md, row
  // "dequeue"), which at one atomic per task would cap a 30 k-task batch near 0.4 ms; waves therefore take tickets
  // in chunks of EXT_CHUNK consecutive tasks.
  // Guided dequeue (chunk == 0, the default): a wave takes (tasks left) / (2 x waves) tickets at a time, at most guide_cap, and single
  // tasks once the queue runs low -- the tail, where balance matters, is still served task by task, and the early part costs a
  // sixth of the atomics.  Every dequeue is a device-scope atomic that goes to the memory side (the XCDs' L2s are not coherent
  // with each other): at one per task the queue alone was as much fabric traffic as the tasks' own bytes.
  const int total_waves = (int)gridDim.x * WAVES_PER_BLOCK;
  // Behind the sift kernel (SHORT): the tickets are the entries of its to-do list -- the tasks it did not finish, next_task[2] of
  // them from the front of the list and, taken first, next_task[3] from its back: the ones with the longest sweeps ahead.  One
  // ticket at a time: every ticket is a task with a sweep to do (a wave holds the counter for one atomic per ~25 us), and a
  // launch ends with the last tickets taken -- with eight at a time from the whole batch (the guided dequeue below, whose chunk
  // size is one chunk old when it is used), half of the waves had left 155 us before the last one of a 430 us launch
  // (tools/wave_placement.py, profiles/r04_wave_placement.txt).
  const bool listed = SHORT && todo_list != nullptr;
  // (clamped: the counts come from the sift kernel of this call; whatever went wrong before, no ticket indexes outside the list)
  const int n_heavy = listed ? min(uni(next_task[3]), n_tasks) : 0;
  const int n_tix = listed ? min(uni(next_task[2]) + n_heavy, n_tasks) : n_tasks;
  int ticket = 0, ticket_end = 0, take = listed ? 1 : chunk > 0 ? chunk : max(1, min(guide_cap, n_tix / (2 * total_waves)));
  for (;;) {
    if (ticket == ticket_end) {
      ticket = dequeue_task(next_task, take);
      ticket_end = min(ticket + take, n_tix);
      if (chunk == 0 && !listed) take = max(1, min(guide_cap, (n_tix - ticket_end) / (2 * total_waves)));
      if (ticket >= n_tix) {
        // The last wave to leave puts the queue back to zero for the next launch on this context: next_task[1] counts the waves
        // that have taken their last ticket.  Saves the fill kernel that used to zero the head before every launch.
        const int gone = dequeue_task(next_task + 1);
        if (gone == (int)gridDim.x * WAVES_PER_BLOCK - 1) {
          // (SHORT) the host launches the full kernel only when this launch left it something (bpsw_runtime.cpp): the length of
          // the list, next to the results.  Every other wave's pushes returned before it counted itself out above, and
          // the count is read where the atomics were performed (the XCDs' L2s are not coherent with each other).
          if (SHORT && defer_post) {
            const int listed = dequeue_task(defer, 0);
            if (lane == 0) *defer_post = listed;
          }
          if (lane == 0) {
            next_task[0] = 0;
            next_task[1] = 0;
            if (listed) { next_task[2] = 0; next_task[3] = 0; }
          }
        }
        break;
      }
    }
    const int task = listed ? uni(todo_list[ticket < n_heavy ? n_tasks - 1 - ticket : ticket - n_heavy])
                            : task_list ? uni(task_list[ticket]) : ticket;  // n_tasks counts the entries of task_list when given
    ++ticket;
    // what the sift kernel (bpsw_extend_sift.hip: the shortcuts, one task per lane) left for this task: 1 = its record is written,
    // 2 = a record per side in sift_recs, 0 = nothing
    int sifted = 0;
    if (SHORT && sift_flag) {
      sifted = uni((int)sift_flag[task]);
      if (sifted == 1) continue;
    }
    BPSW_DIAG_TASK_BEGIN(task);
    ext_do_task<COORD, SHORT, false>(task, sifted, lane, wire, out, sc, sc.mat, oDel, eDel, oIns, eIns, penClip5, penClip3, wBand, exact_a, amax, eh, qp, ts, pl,
                                     defer, short_qmax, sift_recs, inject_defer);
    BPSW_DIAG_TASK_END();
  }
  BPSW_DIAG_WAVE_END_TASKS(SHORT ? 1 : 0, next_task, lane);
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
    if (oIns < 0 || eIns < 1 || oDel < 0 || eDel < 1) err = 2;  // the prefix-scan form of F needs oIns >= 0; e = 0 divides by zero in SWUtil.scala:110-115
    if ((int8_t)((wire[1] >> 16) & 0xff) < 0) err = 2;         // band width is a signed byte
    if ((wire[1] >> 24) != 0) err = 5;                          // header byte 7: a coordinate batch (format 2) goes through bpsw_extend_batch
    pre->reserved = oIns + eIns > 0 ? 1 : 0;                    // the register sweeps are usable (their prefix-scan form of F needs a positive gap cost)
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
  // one atomic per wavefront and word, not per task: 30 k same-address atomics were most of this kernel's 36 us
  mq = wave_max(mq); mr = wave_max(mr); err = wave_max(err);
  if ((threadIdx.x & 63) == 0) {
    if (mq) atomicMax(&pre->max_qlen, mq);
    if (mr) atomicMax(&pre->max_rlen, mr);
    if (err) atomicMax(&pre->error, err);
  }
}

// The resident form of the short kernel (bpsw_ring.h, class RING_CLASS_EXT): the same tasks, taken a unit at a time from the descriptors
// task threads append to the device's submission ring instead of from one launch's queue.  Wavefront 0 of workgroup 0 is the ring's
// poller; every other wavefront is a worker.  Only what needs neither the sift kernel nor the full kernel comes here (small batches
// of flanks up to 255 bases, of either wire format: the host checks), so a descriptor is ONE phase: its units are its tasks.  LDS geometry fixed for the epoch.
// The batch is read with ordinary (vector) loads from a pointer that arrives in the descriptor -- nothing here is `__restrict__` kernel
// argument memory the compiler may keep in the scalar cache across batches -- and a unit's first act is ring_next_unit's invalidate of
// the vector cache: a caller's staging buffer is reused from batch to batch.
// (register budget: the full kernel's 96 VGPRs.  The epoch's grid is one workgroup per CU -- a wave per SIMD --, so nothing is gained by
// squeezing it into the short kernel's 64, where the ring's own state and the descriptor's fields cost 160 bytes of scratch per lane)
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, BPSW_EXT_WAVES_PER_SIMD) void ext_resident_kernel(const RingArgs A, const int qcap, const int rcap,
                                                                                                        const int lds_per_wave) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ MatRows mat_lds[WAVES_PER_BLOCK];
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  if (blockIdx.x == 0 && wave == 0) {
    ring_poller(A, lane);
    return;
  }
  unsigned char* base = smem + (size_t)wave * lds_per_wave;
  uint8_t* ts = base;
  int* prof_w = reinterpret_cast<int*>(base + (((size_t)rcap + 15) & ~(size_t)15));
  const ProfLds pl = {prof_w, reinterpret_cast<int8_t*>(prof_w + qcap + 2), (unsigned)(uintptr_t)((__attribute__((address_space(3))) int*)prof_w)};
  RingWorker W;
  for (;;) {
    uint32_t unit = 0, word = 0;
    if (!ring_next_unit(A, lane, W, unit, word)) break;
    const auto f32 = [&](const int k) { return (uint32_t)__builtin_amdgcn_readlane((int)word, k); };
    const auto f64 = [&](const int k) { return ((unsigned long long)f32(k + 1) << 32) | (unsigned long long)f32(k); };
    // the payload (ExtRingPayload, words 8..) as the arguments a launch would have got
    const uint32_t* wire = (const uint32_t*)f64(8);
    int16_t* out = (int16_t*)f64(10);
    const int n_tasks = (int)f32(12), per_unit = (int)f32(13);
    ExtScoring sc;
    sc.out_stride = (int)f32(14); sc.zdrop = (int)f32(15); sc.zdrop_mode = (int)f32(16); sc.mat_max = (int)f32(17);
    sc.exact_a = (int)f32(18); sc.tail_bound = (int)f32(19); sc.certify = (int)f32(20);
    const bool coord = f32(21) != 0u;  // a coordinate batch (wire format 2): target flanks from the device-resident reference
    sc.side_how = nullptr; sc.pac = coord ? (const uint8_t*)f64(32) : nullptr; sc.l_pac = coord ? (long long)f64(34) : 0ll;
    // the matrix rows into this wave's LDS copy (lane r: row r, descriptor words 22 + 2r, 23 + 2r)
    {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((22 + 2 * lane) << 2, (int)word), hi = (uint32_t)__builtin_amdgcn_ds_bpermute((23 + 2 * lane) << 2, (int)word);
      if (lane < 5) mat_lds[wave].row[lane] = ((unsigned long long)hi << 32) | (unsigned long long)lo;
    }
    // header, MemChainToAlignBatched.scala:78-84 (signed bytes) -- as ext_kernel reads it
    uint32_t hw = 0;
    if (lane < 2) hw = __hip_atomic_load(const_cast<uint32_t*>(wire) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t hdr0 = (uint32_t)__builtin_amdgcn_readlane((int)hw, 0), hdr1 = (uint32_t)__builtin_amdgcn_readlane((int)hw, 1);
    const int oDel = (int8_t)(hdr0 & 0xff), eDel = (int8_t)((hdr0 >> 8) & 0xff);
    const int oIns = (int8_t)((hdr0 >> 16) & 0xff), eIns = (int8_t)((hdr0 >> 24) & 0xff);
    const int penClip5 = (int8_t)(hdr1 & 0xff), penClip3 = (int8_t)((hdr1 >> 8) & 0xff);
    const int wBand = (int8_t)((hdr1 >> 16) & 0xff);
    const int oe_min = min(oIns + eIns, oDel + eDel);
    const int exact_a = (oe_min > 0 && wBand >= 2) ? sc.exact_a : 0;
    const int amax = sc.tail_bound ? sc.mat_max : 0;
    const int t_end = min(n_tasks, ((int)unit + 1) * per_unit);
    if (coord) {
      for (int task = (int)unit * per_unit; task < t_end; ++task)
        ext_do_task<true, 1, true>(task, 0, lane, wire, out, sc, mat_lds[wave], oDel, eDel, oIns, eIns, penClip5, penClip3, wBand, exact_a, amax, nullptr, nullptr, ts, pl,
                                   nullptr, 255, nullptr, 0);
    } else {
      for (int task = (int)unit * per_unit; task < t_end; ++task)
        ext_do_task<false, 1, true>(task, 0, lane, wire, out, sc, mat_lds[wave], oDel, eDel, oIns, eIns, penClip5, penClip3, wBand, exact_a, amax, nullptr, nullptr, ts, pl,
                                    nullptr, 255, nullptr, 0);
    }
    ring_unit_done(A, lane, W, word);
  }
}

}  // namespace

hipError_t launch_ext_resident(const RingArgs& A, int blocks, hipStream_t s) {
  const size_t per_wave = (((size_t)EXT_RING_RCAP + 15) & ~(size_t)15) + ((5 * ((size_t)EXT_RING_QCAP + 2) + 15) & ~(size_t)15);
  const size_t lds = per_wave * WAVES_PER_BLOCK;
  hipLaunchKernelGGL(ext_resident_kernel, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), lds, s, A, (int)EXT_RING_QCAP, (int)EXT_RING_RCAP, (int)per_wave);
  return hipGetLastError();
}

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
                             int rcap, int num_cu, int* d_counter, const int* d_task_list, hipStream_t s,
                             const ExtPrepass* d_pre_check, bool counter_zeroed, KernelEvents kev, bool short_kernel, int* d_defer,
                             int short_qmax, const uint8_t* d_sift_flag, const uint4* d_sift_recs, int* d_defer_post, const int* d_todo_list) {
  if (n_tasks <= 0) return hipSuccess;  // (the full kernel behind a SHORT launch: n_tasks = the most its device-side list can hold)
  if (short_kernel && !d_defer) return hipErrorInvalidValue;  // the deferring build needs somewhere to defer to
  static const int inject_env = getenv("BPSW_EXT_INJECT_DEFER") ? atoi(getenv("BPSW_EXT_INJECT_DEFER")) : 0;  // test hook (tests/test_extend_gpu.py)
  const int inject_defer = short_kernel ? inject_env : 0;
  const bool coord = sc.pac != nullptr;  // a coordinate batch (the caller sets ExtScoring::pac only for those)
  const int variant = short_kernel ? 1 : 0;
  const void* fn = variant == 0 ? (coord ? reinterpret_cast<const void*>(ext_kernel<true, 0>) : reinterpret_cast<const void*>(ext_kernel<false, 0>))
                                : (coord ? reinterpret_cast<const void*>(ext_kernel<true, 1>) : reinterpret_cast<const void*>(ext_kernel<false, 1>));
  // round the capacities so that a handful of LDS configurations cover all batches
  qcap = (qcap + 31) & ~31;
  rcap = (rcap + 63) & ~63;
  // short kernels: the target bytes, then the call's query profile (qcap + 2 words and bytes: ProfLds, bpsw_extend_rows.h)
  const size_t per_wave = short_kernel ? (((size_t)rcap + 15) & ~(size_t)15) + ((5 * ((size_t)qcap + 2) + 15) & ~(size_t)15)
                                       : ext_lds_per_wave(qcap, rcap);
  const size_t lds = per_wave * WAVES_PER_BLOCK;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  // the opt-in to > 64 KB of dynamic LDS is a property of the function ON A DEVICE: remember the largest size per device
  static std::atomic<size_t> attr_set_v[4][64];
  std::atomic<size_t>* attr_set = attr_set_v[(coord ? 1 : 0) + 2 * variant];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (lds > 64 * 1024 && lds > attr_set[dev].load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute(fn,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    size_t seen = attr_set[dev].load(std::memory_order_relaxed);
    while (seen < lds && !attr_set[dev].compare_exchange_weak(seen, lds, std::memory_order_relaxed)) {}
  }
  // resident workgroups per CU: 8 waves/SIMD = 8 blocks of 4 waves, capped by LDS
  int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
  // How many persistent workgroups a launch gets per CU.  Filling every wave slot with ONE launch is the slowest choice.
  // Measured on MI355X with 16 HIP hardware queues (tools/ext_kernel_time.py n; reads/s in millions for 1 / 2 / 4 / 8 / 16 batches
  // in flight, and the bench step):   2 per CU 72 / 118 / 147 / 189 / 211 / 120.8;   1.5 per CU 73 / 123 / 172 / 204 / 231 / 133.0;
  // 1.25 per CU 73 / 123 / 177 / 215 / 241 / 136.5;   1 per CU 67 / 114 / 169 / 212 / 237 / 141.4;   0.75 per CU 54 / 93 / 149 /
  // 197 / 233 / 135.5.  A launch's persistent waves only leave when its queue is empty, so big grids run one after the other,
  // each with its own tail of long tasks; small grids of several launches (and the rescue kernel of the same step) are
  // resident together and fill each other's tails.
  static const double cap_per_cu = getenv("BPSW_EXT_BLOCKS_PER_CU") ? atof(getenv("BPSW_EXT_BLOCKS_PER_CU")) : 1.0;
  // the short kernel: 1.25 workgroups per CU.  (Round 2, every pass of the bench behind a barrier: reads/s in millions at 1 / 1.5 / 2 / 3
  // per CU 152.5 / 158.1 / 157.5 / 147.2, and two it was.  Round 3, the calls following each other freely and the sift kernel in front:
  // configs[1] 360 / 361 / 326 at 1 / 1.25 / 2, the default workload 165-180 / 179 / 181 at 1 / 0.75 / 2 -- within its noise --,
  // configs[4] 17.9 / 17.7 / 17.7 at 1 / 2 / 3: smaller grids leave wave slots to the one-wave workgroups of the sift kernels.)
  static const double cap_per_cu_short = getenv("BPSW_EXT_SHORT_BLOCKS_PER_CU") ? atof(getenv("BPSW_EXT_SHORT_BLOCKS_PER_CU")) : 1.25;
  double per_cu_f = per_cu < 1 ? 1.0 : (double)per_cu;
  if (per_cu_f > (short_kernel ? cap_per_cu_short : cap_per_cu)) per_cu_f = short_kernel ? cap_per_cu_short : cap_per_cu;
  int blocks = (n_tasks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  int max_blocks = (int)(num_cu * per_cu_f);
  if (max_blocks < 1) max_blocks = 1;
  if (blocks > max_blocks) blocks = max_blocks;
  (void)counter_zeroed;  // the queue head (d_counter[0], [1]) is zero between launches: the kernel's last wave resets it
  // tasks per dequeue: 0 = guided (see the kernel), n > 0 = fixed chunks of n (1 balances a lone launch best)
  static const int chunk = [] { const int v = getenv("BPSW_EXT_CHUNK") ? atoi(getenv("BPSW_EXT_CHUNK")) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }();  // 0: guided
  static const int guide_cap = [] { const int v = getenv("BPSW_EXT_GUIDE_CAP") ? atoi(getenv("BPSW_EXT_GUIDE_CAP")) : 8; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
#define BPSW_EXT_GO(CO, SH)                                                                                                     \
  BPSW_LAUNCH(kev, (ext_kernel<CO, SH>), dim3(blocks), dim3(64 * WAVES_PER_BLOCK), lds, s, d_wire, n_tasks, d_out, sc, qcap, rcap, \
              (int)per_wave, chunk, guide_cap, d_counter, d_task_list, d_pre_check, d_defer, short_qmax, d_sift_flag, d_sift_recs, d_defer_post, d_todo_list, inject_defer)
  if (variant == 1) {
    if (coord) BPSW_EXT_GO(true, 1); else BPSW_EXT_GO(false, 1);
  } else {
    if (coord) BPSW_EXT_GO(true, 0); else BPSW_EXT_GO(false, 0);
  }
#undef BPSW_EXT_GO
  return hipGetLastError();
}

}  // namespace bpsw
