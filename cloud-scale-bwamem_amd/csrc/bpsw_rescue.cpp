// bpsw_rescue.cpp -- host side of boundary 1: the batched pair-end rescue around the SW kernel.
//
// Replaces mem_group_matesw / mem_matesw_precompute (src/main/native/bwamem_pair.c:115-228) and
// mem_sort_and_dedup (src/main/native/bwamem.c:394-435), i.e. what jniNative.so runs under
// MateSWJNI.mateSWJNI; optionally reproduces the pure-Scala path instead
// (MemSamPe.scala:1111-1238,1335-1369 + MemSortAndDedup.scala:33-141).
//
// The reference walks pairs, ends, anchors and orientations sequentially and calls the SW inside the
// walk.  Here the walk is split: the SW result of (pair k, end i, anchor j, orientation r) is a pure
// function of inputs that arrive precomputed (mate bytes, the four windows), and only the skip[] test
// reads the evolving mate list.  So:
//   1. speculate: evaluate skip[] against the INITIAL mate lists, collect every (k,i,j,r) that passes;
//   2. one flat GPU batch computes SWAlign2 for all of them;
//   3. replay the reference's sequential walk consuming the precomputed results.
// A job the replay needs that step 1 did not launch (possible only when a dedup removed the region
// that justified a skip) is collected and computed by ANOTHER GPU round, then the affected pairs are
// replayed again; there is no CPU alignment path.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "bpsw_internal.h"
#include "bpsw_rescue_skip.h"

using namespace bpsw;

namespace {

typedef bpsw_alnreg_t Reg;

// ---- ordering predicates ---------------------------------------------------------------------
struct LtRe {  // alnreg_slt2, native/bwamem.c:385
  bool operator()(const Reg& x, const Reg& y) const { return x.re < y.re; }
};
struct LtScore {  // alnreg_slt, native/bwamem.c:388 == sortBy(-score, rBeg, qBeg), MemSortAndDedup.scala:114
  bool operator()(const Reg& x, const Reg& y) const {
    return x.score > y.score || (x.score == y.score && (x.rb < y.rb || (x.rb == y.rb && x.qb < y.qb)));
  }
};
struct LtReRb {  // sortBy(rEnd, rBeg), MemSortAndDedup.scala:40
  bool operator()(const Reg& x, const Reg& y) const { return x.re < y.re || (x.re == y.re && x.rb < y.rb); }
};

template <class Lt>
void insertion_sort(Reg* first, Reg* last, Lt lt) {  // stable
  for (Reg* i = first + 1; i < last; ++i)
    for (Reg* j = i; j > first && lt(*j, *(j - 1)); --j) std::swap(*j, *(j - 1));
}

template <class Lt>
void comb_sort(size_t n, Reg* a, Lt lt) {  // fallback of klib's introsort, native/ksort.h:154-175
  const double shrink = 1.2473309501039786540366528676643;
  size_t gap = n;
  bool swapped;
  do {
    if (gap > 2) {
      gap = (size_t)(gap / shrink);
      if (gap == 9 || gap == 10) gap = 11;
    }
    swapped = false;
    for (Reg* i = a; i < a + n - gap; ++i)
      if (lt(*(i + gap), *i)) { std::swap(*i, *(i + gap)); swapped = true; }
  } while (swapped || gap > 2);
  if (gap != 1) insertion_sort(a, a + n, lt);
}

// The C library's tie order is a property of klib's ks_introsort (native/ksort.h:176-227): median of
// (first, middle+1, last) as pivot moved to the end, Hoare partition, sub-ranges of <= 16 elements left
// for one final insertion sort, comb sort when the depth budget runs out.  To hand the caller the same
// region order as jniNative.so does, the same sequence of comparisons and swaps is performed here.
template <class Lt>
void klib_order_sort(size_t n, Reg* a, Lt lt) {
  struct Frame { Reg *lo, *hi; int depth; };
  if (n < 1) return;
  if (n == 2) {
    if (lt(a[1], a[0])) std::swap(a[0], a[1]);
    return;
  }
  int d = 2;
  while ((1ul << d) < n) ++d;
  std::vector<Frame> stack;
  stack.reserve(sizeof(size_t) * (size_t)d + 2);
  Reg *s = a, *t = a + (n - 1);
  d <<= 1;
  for (;;) {
    if (s < t) {
      if (--d == 0) { comb_sort((size_t)(t - s) + 1, s, lt); t = s; continue; }
      Reg *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
      if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
      else k = lt(*j, *i) ? i : j;
      const Reg pivot = *k;
      if (k != t) std::swap(*k, *t);
      for (;;) {
        do ++i; while (lt(*i, pivot));
        do --j; while (i <= j && lt(pivot, *j));
        if (j <= i) break;
        std::swap(*i, *j);
      }
      std::swap(*i, *t);
      if (i - s > t - i) {
        if (i - s > 16) stack.push_back({s, i - 1, d});
        s = t - i > 16 ? i + 1 : t;
      } else {
        if (t - i > 16) stack.push_back({i + 1, t, d});
        t = i - s > 16 ? i - 1 : s;
      }
    } else {
      if (stack.empty()) { insertion_sort(a, a + n, lt); return; }
      s = stack.back().lo; t = stack.back().hi; d = stack.back().depth;
      stack.pop_back();
    }
  }
}

// mem_sort_and_dedup (native/bwamem.c:394-435) / memSortAndDedup (MemSortAndDedup.scala:33-141)
int sort_dedup(std::vector<Reg>& v, float mask, int mode) {
  int n = (int)v.size();
  if (n <= 1) return n;
  Reg* a = v.data();
  if (mode == BPSW_RESCUE_C) klib_order_sort((size_t)n, a, LtRe());
  else insertion_sort(a, a + n, LtReRb());
  for (int i = 1; i < n; ++i) {
    Reg& p = a[i];
    if (p.rb >= a[i - 1].re) continue;
    for (int j = i - 1; j >= 0 && p.rb < a[j].re; --j) {
      Reg& q = a[j];
      if (q.qe == q.qb) continue;  // already excluded
      const int64_t orr = q.re - p.rb;
      const int64_t oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
      const int64_t mr = std::min(q.re - q.rb, p.re - p.rb);
      const int64_t mq = std::min<int64_t>(q.qe - q.qb, p.qe - p.qb);
      if ((float)orr > mask * (float)mr && (float)oq > mask * (float)mq) {  // one of the two is redundant
        if (p.score < q.score) { p.qe = p.qb; break; }
        q.qe = q.qb;
      }
    }
  }
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (a[i].qe > a[i].qb) a[m++] = a[i];
  n = m;
  if (mode == BPSW_RESCUE_C) klib_order_sort((size_t)n, a, LtScore());
  else insertion_sort(a, a + n, LtScore());
  for (int i = 1; i < n; ++i)  // identical hits
    if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
  m = n < 1 ? 0 : 1;  // a[0] is never marked
  for (int i = 1; i < n; ++i)
    if (a[i].qe > a[i].qb) a[m++] = a[i];
  v.resize((size_t)m);
  return m;
}

inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t* dist) { return rescue_infer_dir(l_pac, b1, b2, dist); }  // bpsw_rescue_skip.h

// Per-context scratch of bpsw_matesw_group: every vector keeps its capacity between calls, so the steady state allocates nothing.
struct Want { int64_t x; int32_t mate; };  // window index (anchor row * 4 + orientation) and the end (2k + i) whose read is aligned
struct Scratch {
  std::vector<int64_t> t_base;              // per TOUCHED pair, four words: first region of end 0 / end 1, first anchor row of end 0 / end 1
  std::vector<int64_t> job_x;               // launched windows, ascending within a round (round boundaries in job_round)
  std::vector<size_t> job_round;            // first job of every round; job_x[job_round[r] .. job_round[r+1]) is sorted
  std::vector<int32_t> results;             // 7 ints per launched job
  std::vector<uint8_t> used;                // per job: consumed by the replay
  std::vector<Want> want, want_next;
  std::vector<int32_t> touched;             // pairs with at least one window to align, ascending
  std::vector<int64_t> mate_slot;           // per end: offset of the read in the staged mate pool, -1 = not staged
  std::vector<int32_t> staged_mates;        // ends whose mate_slot is set (to reset it)
  std::vector<Reg> arena;                   // final lists of the touched pairs
  std::vector<int64_t> fin_off;             // per touched pair x end: offset into arena
  std::vector<int32_t> fin_cnt;
  std::vector<uint8_t> done;                // per touched pair
  std::vector<Reg> anchors[2], v[2];
};

struct Group {
  const bpsw_opt_t* opt;
  const bpsw_rescue_group_t* g;
  int mode;
  bool pac_mode;
  Scratch* S;
  int failed_mask;  // bit r: pes[r].failed (an orientation without statistics is skipped for every anchor)
  int32_t pes_low[4], pes_high[4];
};

void skip_flags(const Group& G, const Reg& a, const Reg* mates, size_t n_mates, int skip[4]) {
  rescue_skip_flags(G.g->l_pac, G.pes_low, G.pes_high, G.failed_mask, G.mode == BPSW_RESCUE_SCALA, a.rb, mates ? &mates->rb : nullptr, sizeof(Reg), n_mates, skip);  // (no mate regions: no address to form -- UBSan, tests/host_san)
}

// length of window x as the SW sees it: shipped with the bytes, or -- coordinate mode, SURVEY.md 8f.2 -- what bnsGetSeq
// (util/BNTSeqUtil.scala:37-59) would return for (rb, re): swap, clamp to [0, 2 l_pac), nothing when it bridges the strands
inline int64_t win_len(const Group& G, int64_t x) {
  if (!G.pac_mode) return G.g->ref_len[x];
  int64_t b = G.g->ref_rb[x], e = G.g->ref_re[x];
  if (b < 0 && e < 0) return 0;  // (-1,-1): failed orientation, MemSamPe.scala:1863-1868
  if (e < b) std::swap(b, e);
  if (e > (G.g->l_pac << 1)) e = G.g->l_pac << 1;
  if (b < 0) b = 0;
  if (!(b >= G.g->l_pac || e <= G.g->l_pac)) return 0;
  return e - b > 0 ? e - b : 0;
}
inline bool window_ok(const Group& G, int64_t x) { return win_len(G, x) == G.g->ref_re[x] - G.g->ref_rb[x]; }

// region built from an SWAlign2 result: native/bwamem_pair.c:203-212 / MemSamPe.scala:1192-1212
bool make_region(const Group& G, const int32_t aln[7], int r, int l_ms, int64_t x, Reg* out) {
  if (!(aln[0] >= G.opt->min_seed_len && aln[6] >= 0)) return false;
  const bool is_rev = (r >> 1) != (r & 1);
  const int64_t l2 = G.g->l_pac << 1, rbeg = G.g->ref_rb[x];
  Reg b;
  memset(&b, 0, sizeof b);
  if (is_rev) {
    b.qb = l_ms - (aln[2] + 1); b.qe = l_ms - aln[6];
    b.rb = l2 - (rbeg + aln[1] + 1); b.re = l2 - (rbeg + aln[5]);
  } else {
    b.qb = aln[6]; b.qe = aln[2] + 1;
    if (G.mode == BPSW_RESCUE_C) { b.rb = rbeg + aln[5]; b.re = rbeg + aln[1] + 1; }
    else { b.rb = rbeg + aln[1] + 1; b.re = rbeg + aln[1] + 1; }  // MemSamPe.scala:1203-1204
  }
  b.score = aln[0]; b.csub = aln[3]; b.secondary = -1;
  b.seedcov = (int32_t)(std::min<int64_t>(b.re - b.rb, b.qe - b.qb) >> 1);
  *out = b;
  return true;
}

// SW result of window x for the replay: nullptr when it has not been computed yet.  A mate without bases (l_ms < 1, e.g.
// an adapter-trimmed read) never becomes a job: the reference's ksw_align2 returns score 0 for it, which is no region.
static const int32_t kNoHit[7] = {0, -1, -1, -1, -1, -1, -1};
inline const int32_t* result_of(Group& G, int64_t x, int l_ms) {
  if (l_ms < 1) return kNoHit;
  // the launched windows are kept sorted per round (a table over all 4 x anchors windows would have to be cleared per call):
  // binary search, newest round first -- almost always the only one
  const Scratch& S = *G.S;
  for (size_t r = S.job_round.size(); r-- > 0;) {
    const size_t lo = S.job_round[r], hi = r + 1 < S.job_round.size() ? S.job_round[r + 1] : S.job_x.size();
    const int64_t* first = S.job_x.data() + lo;
    const int64_t* last = S.job_x.data() + hi;
    const int64_t* it = std::lower_bound(first, last, x);
    if (it != last && *it == x) {
      const size_t job = (size_t)(it - S.job_x.data());
      G.S->used[job] = 1;
      return &S.results[7 * job];
    }
  }
  return nullptr;
}

// One anchor against the mate list.  Returns false when a needed SW result is missing (the window goes to `missing`).
bool precompute(Group& G, const Reg& a, int l_ms, int mate_end, std::vector<Reg>& ma, int64_t xrow, std::vector<Want>& missing) {
  int skip[4];
  skip_flags(G, a, ma.data(), ma.size(), skip);
  if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return true;
  int n = 0;
  if (G.mode == BPSW_RESCUE_C) {
    for (int r = 0; r < 4; ++r) {
      if (skip[r]) continue;
      const int64_t x = xrow * 4 + r;
      if (window_ok(G, x)) {
        const int32_t* res = result_of(G, x, l_ms);
        if (!res) { missing.push_back({x, mate_end}); return false; }
        Reg b;
        if (make_region(G, res, r, l_ms, x, &b)) {
          // keep the list sorted by score: insert before the first lower score (native/bwamem_pair.c:213-219)
          size_t at = 0;
          while (at < ma.size() && !(ma[at].score < b.score)) ++at;
          ma.insert(ma.begin() + (long)at, b);
        }
        ++n;
      }
      if (n) sort_dedup(ma, G.opt->mask_level_redun, G.mode);
    }
    return true;
  }
  // Scala flavour (MemSamPe.scala:1155-1237): new hits are appended to a copy that is re-sorted by score
  // (ascending, stable) and de-duplicated from scratch after every orientation; the copy itself is never
  // replaced by the dedup result, but the objects the dedup kills stay killed (shared references).
  std::vector<Reg> upd(ma), last;
  for (int r = 0; r < 4; ++r) {
    if (skip[r]) continue;
    const int64_t x = xrow * 4 + r;
    if (window_ok(G, x)) {
      const int32_t* res = result_of(G, x, l_ms);
      if (!res) { missing.push_back({x, mate_end}); return false; }
      Reg b;
      if (make_region(G, res, r, l_ms, x, &b)) upd.push_back(b);
      ++n;
    }
    if (n) {
      std::stable_sort(upd.begin(), upd.end(), [](const Reg& p, const Reg& q) { return p.score < q.score; });
      // identity of each object travels in `hash` while the dedup shuffles the copy
      std::vector<uint64_t> saved(upd.size());
      last = upd;
      for (size_t i = 0; i < last.size(); ++i) { saved[i] = last[i].hash; last[i].hash = i; }
      sort_dedup(last, G.opt->mask_level_redun, G.mode);
      std::vector<uint8_t> alive(upd.size(), 0);
      for (const Reg& s : last) alive[(size_t)s.hash] = 1;
      for (size_t i = 0; i < upd.size(); ++i)
        if (!alive[i]) upd[i].qe = upd[i].qb;
      for (Reg& s : last) s.hash = saved[(size_t)s.hash];
    }
  }
  if (n > 0) ma = last;
  return true;
}

// Replays pair k from its initial state.  Returns false (and fills `missing`) if a result is not available yet.
bool replay_pair(Group& G, int k, size_t ti, std::vector<Reg> v[2], std::vector<Want>& missing) {
  const bpsw_rescue_group_t* g = G.g;
  std::vector<Reg>* tmp = G.S->anchors;
  const int64_t* tb = G.S->t_base.data() + 4 * ti;
  for (int i = 0; i < 2; ++i) {
    const Reg* first = g->regs + tb[i];
    v[i].assign(first, first + g->reg_cnt[2 * k + i]);
    tmp[i].clear();
    for (const Reg& r : v[i])  // anchors: filtered copy taken before any rescue (native/bwamem_pair.c:126-131)
      if (r.score >= v[i][0].score - G.opt->pen_unpaired) tmp[i].push_back(r);
  }
  for (int i = 0; i < 2; ++i) {
    const int na = std::min<int>((int)tmp[i].size(), std::min<int>(G.opt->max_matesw, g->ref_cnt[2 * k + i]));
    for (int j = 0; j < na; ++j)
      if (!precompute(G, tmp[i][(size_t)j], g->seq_len[2 * k + !i], 2 * k + !i, v[!i], tb[2 + i] + j, missing))
        return false;
  }
  return true;
}

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace

namespace bpsw {
int sort_dedup_regs(std::vector<bpsw_alnreg_t>& v, float mask_level_redun, int mode) { return sort_dedup(v, mask_level_redun, mode); }
void rescue_scratch_free(void* p) { delete (Scratch*)p; }
}  // namespace bpsw

// The host layer runs once per group on the calling thread, between the JNI marshalling and the kernel, and at 4 096 pairs
// per group its cost rivals the kernel's: it is written as ONE sequential pass over the group's arrays (prefix sums,
// validation and speculation together; a properly paired end costs two loads and one distance test), the wanted jobs are
// packed straight into the pinned staging block the kernel reads, only the pairs with a job are replayed, and the output is
// assembled with one memcpy per run of untouched pairs.
extern "C" int bpsw_matesw_group(bpsw_ctx_t* c, const bpsw_opt_t* opt, const bpsw_rescue_group_t* g, int mode,
                                 int32_t* out_cnt, bpsw_alnreg_t* out_regs, int64_t out_cap, int64_t* out_total) {
  if (!c || !opt || !g || !out_cnt || !out_total) return fail(BPSW_ERR_ARG, "matesw_group: null argument");
  if (mode != BPSW_RESCUE_C && mode != BPSW_RESCUE_SCALA) return fail(BPSW_ERR_ARG, "matesw_group: bad mode");
  const int G_ = g->group_size;
  if (G_ < 0) return fail(BPSW_ERR_ARG, "matesw_group: negative group size");
  std::lock_guard<std::mutex> lock(c->mu);
  hipError_t he = hipSetDevice(c->device);
  if (he != hipSuccess) return fail(BPSW_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(he));
  { const int prc_ = finish_pending(c); if (prc_ != BPSW_OK) return prc_; }

  const double t_begin = stat_ms();
  double t_pack = 0., t_replay = 0.;
  if (!c->rescue_scratch) c->rescue_scratch = new Scratch();
  Scratch& S = *(Scratch*)c->rescue_scratch;
  Group GR;
  GR.opt = opt; GR.g = g; GR.mode = mode; GR.S = &S;
  // with ref_pool == NULL the windows are named by (rb, re) only and read from the 2-bit reference
  // loaded on the context; their lengths follow bnsGetSeq (win_len)
  GR.pac_mode = g->ref_pool == nullptr;
  const bool rescue_on = (opt->flag & 0x20) == 0;  // MEM_F_NO_RESCUE, native/bwamem.h:18
  static const bool lean = !(getenv("BPSW_RESCUE_LEAN") && atoi(getenv("BPSW_RESCUE_LEAN")) == 0);  // 0: speculate every anchor (A/B)

  // ---- 1. one pass: prefix sums, validation, speculation against the initial lists -----------------------------------
  // (a third of a call's CPU time in round 4, and the calls' CPU time is what bounds the bench step since the rescue kernel is
  // resident: the per-end prefix arrays became four words per TOUCHED pair, the per-pair checks are straight-line on values loaded
  // once, the region records of the pairs ahead are prefetched -- 64 bytes each, two or three per pair, in input order)
  GR.failed_mask = 0;
  for (int r = 0; r < 4; ++r) { GR.failed_mask |= (g->pes[r].failed ? 1 : 0) << r; GR.pes_low[r] = g->pes[r].low; GR.pes_high[r] = g->pes[r].high; }
  S.t_base.clear();
  S.want.clear(); S.touched.clear();
  int64_t nreg = 0, nref = 0;
  {
    const int32_t* reg_cnt = g->reg_cnt;
    const int32_t* ref_cnt = g->ref_cnt;
    const int32_t* seq_len = g->seq_len;
    const int64_t* seq_off = g->seq_off;
    const uint64_t pool_bytes = g->seq_pool_bytes;
    const int max_matesw = opt->max_matesw, pen_unpaired = opt->pen_unpaired;
    for (int k = 0; k < G_; ++k) {
      const int e0 = 2 * k;
      const int rc[2] = {reg_cnt[e0], reg_cnt[e0 + 1]}, fc[2] = {ref_cnt[e0], ref_cnt[e0 + 1]}, sl[2] = {seq_len[e0], seq_len[e0 + 1]};
      if ((rc[0] | rc[1] | fc[0] | fc[1] | sl[0] | sl[1]) < 0) return fail(BPSW_ERR_ARG, "matesw_group: negative count");
      if ((uint64_t)(seq_off[e0] + sl[0]) > pool_bytes || (uint64_t)(seq_off[e0 + 1] + sl[1]) > pool_bytes)
        return fail(BPSW_ERR_ARG, "matesw_group: mate outside seq_pool");
      const int64_t base[2] = {nreg, nreg + rc[0]}, rowb[2] = {nref, nref + fc[0]};
      nreg += (int64_t)rc[0] + rc[1]; nref += (int64_t)fc[0] + fc[1];
      if (!rescue_on) continue;
      __builtin_prefetch(g->regs + nreg + 12);
      __builtin_prefetch(g->regs + nreg + 13);
#ifndef BPSW_PLAN_PF_ROWS
#define BPSW_PLAN_PF_ROWS 16
#endif
      // ... and the window rows of the anchors ahead (ref_rb / ref_re / ref_len: 32 bytes per anchor each).  Only the anchors of pairs
      // that may need a job are looked at -- one pair in ten --, so the hardware sees no stream in them and every look was a miss the
      // loop waited for: with the rows asked for sixteen anchors ahead the plan costs 0.083 ms of CPU per 4 096 pairs instead of 0.146
      // (BPSW_STATS_CLOCK=cpu under the bench; the whole call 0.25 instead of 0.33), at 0.8 MB more read from host memory per call.
      if (BPSW_PLAN_PF_ROWS > 0) {
        __builtin_prefetch(g->ref_rb + 4 * (nref + BPSW_PLAN_PF_ROWS));
        __builtin_prefetch(g->ref_re + 4 * (nref + BPSW_PLAN_PF_ROWS));
        if (!GR.pac_mode) __builtin_prefetch(g->ref_len + 4 * (nref + BPSW_PLAN_PF_ROWS));
      }
      // The common pair -- one hit per end, the two properly paired -- in straight-line code: each end's only anchor (it passes its own
      // score threshold) against the other end's only hit; all four orientations skipped on both sides = nothing to do.  Anything else
      // takes the general walk below.
      if (rc[0] == 1 && rc[1] == 1 && fc[0] >= 1 && fc[1] >= 1 && sl[0] >= 1 && sl[1] >= 1 && max_matesw >= 1 && pen_unpaired >= 0) {
        const int64_t rb0 = g->regs[base[0]].rb, rb1 = g->regs[base[1]].rb;
        int64_t d01, d10;
        const int r01 = infer_dir(g->l_pac, rb0, rb1, &d01), r10 = infer_dir(g->l_pac, rb1, rb0, &d10);
        if (mode == BPSW_RESCUE_SCALA) { d01 = (int64_t)(int32_t)d01; d10 = (int64_t)(int32_t)d10; }
        const int m0 = GR.failed_mask | ((d01 >= GR.pes_low[r01] && d01 <= GR.pes_high[r01]) ? 1 << r01 : 0);
        const int m1 = GR.failed_mask | ((d10 >= GR.pes_low[r10] && d10 <= GR.pes_high[r10]) ? 1 << r10 : 0);
        if (m0 == 15 && m1 == 15) continue;
      }
      bool touched = false;
      for (int i = 0; i < 2; ++i) {
        const int mate = e0 + (i ^ 1);
        if (sl[i ^ 1] < 1 || rc[i] == 0 || fc[i] == 0) continue;
        const Reg* init = g->regs + base[i];
        const Reg* minit = g->regs + base[i ^ 1];
        const int n_init = rc[i], n_mate = rc[i ^ 1];
        const int thr = init[0].score - pen_unpaired;
        int j = 0;
        bool emitted = false;  // an earlier anchor of this end has a job: its hit may make the later anchors' jobs unnecessary
        for (int ai = 0; ai < n_init; ++ai) {
          const Reg& a = init[ai];
          if (!(a.score >= thr)) continue;
          if (j >= max_matesw || j >= fc[i]) break;
          // lean speculation: only the first anchor of an end that has a job is launched now; what the later ones still need once
          // its result is in, the replay asks for (a second, small round: only when that first rescue failed or landed elsewhere) --
          // instead of computing them all and dropping 9 % of the jobs unused (bench step: 467.8 -> 427.3 jobs per group, no
          // second round at all; tests/test_rescue_gpu.py forces one with decoy anchors)
          if (lean && emitted) break;
          int skip[4];
          skip_flags(GR, a, minit, (size_t)n_mate, skip);
          if (skip[0] + skip[1] + skip[2] + skip[3] != 4) {
            const int64_t xrow = rowb[i] + j;
            for (int r = 0; r < 4; ++r)
              if (!skip[r] && window_ok(GR, xrow * 4 + r)) { S.want.push_back({xrow * 4 + r, mate}); touched = true; emitted = true; }
          }
          ++j;
        }
      }
      if (touched) {
        S.touched.push_back(k);
        S.t_base.push_back(base[0]); S.t_base.push_back(base[1]); S.t_base.push_back(rowb[0]); S.t_base.push_back(rowb[1]);
      }
    }
  }
  RefHold ref_hold;  // coordinate mode: the reference stays put until the last round's kernel has been waited for
  if (GR.pac_mode && nref > 0) {
    const uint8_t* d_pac = nullptr;
    long long loaded = 0;
    ref_hold = ref_snapshot(c, &d_pac, &loaded);
    if (loaded <= 0) return fail(BPSW_ERR_ARG, "matesw_group: ref_pool is null and no reference is loaded (bpsw_ref_load)");
    if (loaded != g->l_pac) return fail(BPSW_ERR_ARG, "matesw_group: l_pac differs from the loaded reference");
  }
  S.job_x.clear(); S.job_round.clear();
  S.results.clear(); S.used.clear();
  const double t_planned = stat_ms();

  const int xtra_base = BPSW_KSW_XSUBO | BPSW_KSW_XSTART | (opt->min_seed_len * opt->a);  // KSW_XBYTE is ignored by SWAlign
  const size_t nt = S.touched.size();
  S.arena.clear();
  S.fin_off.assign(2 * nt, 0); S.fin_cnt.assign(2 * nt, 0); S.done.assign(nt, 0);
  S.mate_slot.assign(2 * (size_t)G_, (int64_t)-1);
  S.staged_mates.clear();
  uint64_t rounds = 0;
  const uint64_t speculated = S.want.size();
  size_t n_done = 0;
  for (;;) {
    // ---- 2. one flat GPU batch, packed in place in the pinned staging block ---------------------------------------------
    if (!S.want.empty()) {
      const double t_p0 = stat_ms();
      if (rounds > 0) {  // later rounds collect their windows out of order and possibly twice
        std::sort(S.want.begin(), S.want.end(), [](const Want& p, const Want& q) { return p.x < q.x; });
        S.want.erase(std::unique(S.want.begin(), S.want.end(), [](const Want& p, const Want& q) { return p.x == q.x; }), S.want.end());
      }
      const size_t nj = S.want.size();
      // sizes first: only the mates some job aligns travel to the device (a tenth of the group's reads), each once
      size_t qbytes = 0, tbytes = 0;
      int mq = 0, mt = 0;
      for (const int32_t m : S.staged_mates) S.mate_slot[(size_t)m] = -1;
      S.staged_mates.clear();
      for (size_t t = 0; t < nj; ++t) {
        const Want& w = S.want[t];
        if (S.mate_slot[(size_t)w.mate] < 0) {
          S.mate_slot[(size_t)w.mate] = (int64_t)qbytes;
          S.staged_mates.push_back(w.mate);
          {  // the mate's bases are copied further down: they lie anywhere in the group's read pool, three cache lines nobody has touched
            const uint8_t* mp = g->seq_pool + g->seq_off[w.mate];
            __builtin_prefetch(mp); __builtin_prefetch(mp + 64); __builtin_prefetch(mp + 128);
          }
          qbytes += align16((size_t)g->seq_len[w.mate]);
          mq = std::max(mq, g->seq_len[w.mate]);
        }
        const int64_t len = win_len(GR, w.x);
        if (len > BPSW_SW_MAX_TLEN) return fail(BPSW_ERR_LIMIT, "matesw_group: window longer than the kernel limit");
        if (!GR.pac_mode) {
          if (g->ref_off[w.x] < 0 || (uint64_t)(g->ref_off[w.x] + len) > g->ref_pool_bytes)
            return fail(BPSW_ERR_ARG, "matesw_group: window outside ref_pool");
          tbytes += align16((size_t)len);
#ifndef BPSW_PACK_NO_PREFETCH
          // (the window's bytes are copied further down, from wherever the caller's pool has them: ask for its lines now)
          const uint8_t* wp = g->ref_pool + g->ref_off[w.x];
          for (int64_t o = 0; o < len; o += 64) __builtin_prefetch(wp + o);
#endif
        }
        mt = std::max(mt, (int)len);
      }
      SwStage st;
      int rc = sw_stage_begin(c, (int)nj, qbytes ? qbytes : 16, GR.pac_mode ? 0 : (tbytes ? tbytes : 16), &st);
      if (rc != BPSW_OK) return rc;
      int32_t* q_len = (int32_t*)(st.base + st.o_qlen);
      int32_t* t_len = (int32_t*)(st.base + st.o_tlen);
      int64_t* q_off = (int64_t*)(st.base + st.o_qoff);
      int64_t* t_off = (int64_t*)(st.base + st.o_toff);
      uint8_t* q_rev = st.base + st.o_qrev;
      uint8_t* qpool = st.base + st.o_qpool;
      uint8_t* tpool = st.base + st.o_tpool;
      for (const int32_t m : S.staged_mates) {
        uint8_t* dst = qpool + S.mate_slot[(size_t)m];
        const size_t ln = (size_t)g->seq_len[m];
        memcpy(dst, g->seq_pool + g->seq_off[m], ln);
        memset(dst + ln, 0, align16(ln) - ln);
      }
      size_t at = 0;
      for (size_t t = 0; t < nj; ++t) {
        const Want& w = S.want[t];
        const int r = (int)(w.x & 3);
        q_len[t] = g->seq_len[w.mate];
        q_off[t] = S.mate_slot[(size_t)w.mate];
        q_rev[t] = ((r >> 1) != (r & 1)) ? 1 : 0;  // native/bwamem_pair.c:177
        const int64_t len = win_len(GR, w.x);
        t_len[t] = (int32_t)len;
        if (GR.pac_mode) {  // window_ok held, so the window starts at ref_rb[x] unclamped
          t_off[t] = g->ref_rb[w.x];
          continue;
        }
        t_off[t] = (int64_t)at;
        memcpy(tpool + at, g->ref_pool + g->ref_off[w.x], (size_t)len);
        at += align16((size_t)len);
      }
      const double t_p1 = stat_ms();
      const int32_t* res = nullptr;
      rc = sw_stage_run(c, opt, xtra_base, st, mq, mt, GR.pac_mode, &res);
      if (rc != BPSW_OK) return rc;
      const size_t first = S.results.size() / 7;
      S.results.insert(S.results.end(), res, res + 7 * nj);
      S.used.resize(first + nj, 0);
      S.job_round.push_back(first);
      for (size_t t = 0; t < nj; ++t) S.job_x.push_back(S.want[t].x);   // ascending: the first round is generated in order, later ones are sorted
      S.want.clear();
      t_pack += t_p1 - t_p0;
    }
    // ---- 3. replay the pairs that had a job ------------------------------------------------------------------------
    const double t_r0 = stat_ms();
#ifndef BPSW_REPLAY_NO_PREFETCH
    // (the thread has slept through the device phase and other threads have had its core: the touched pairs' region records and their
    // anchors' window rows are cold again.  Ask for the next pairs' lines while one is replayed.)
    const auto prefetch_pair = [&](const size_t tj) {
      const int64_t* tb = S.t_base.data() + 4 * tj;
      const int kk = S.touched[tj];
      for (int i = 0; i < 2; ++i) {
        const Reg* r0 = g->regs + tb[i];
        for (int j = 0; j < g->reg_cnt[2 * kk + i] && j < 4; ++j) __builtin_prefetch(r0 + j);
        __builtin_prefetch(g->ref_rb + 4 * tb[2 + i]); __builtin_prefetch(g->ref_re + 4 * tb[2 + i]);
        if (!GR.pac_mode) __builtin_prefetch(g->ref_len + 4 * tb[2 + i]);
      }
    };
    for (size_t tj = 0; tj < nt && tj < 8; ++tj) prefetch_pair(tj);
#endif
    for (size_t ti = 0; ti < nt; ++ti) {
#ifndef BPSW_REPLAY_NO_PREFETCH
      if (ti + 8 < nt) prefetch_pair(ti + 8);
#endif
      if (S.done[ti]) continue;
      const int k = S.touched[ti];
      if (!replay_pair(GR, k, ti, S.v, S.want)) continue;
      for (int i = 0; i < 2; ++i) {
        S.fin_off[2 * ti + (size_t)i] = (int64_t)S.arena.size();
        S.fin_cnt[2 * ti + (size_t)i] = (int32_t)S.v[i].size();
        S.arena.insert(S.arena.end(), S.v[i].begin(), S.v[i].end());
      }
      S.done[ti] = 1;
      ++n_done;
    }
    t_replay += stat_ms() - t_r0;
    if (n_done == nt) break;
    ++rounds;
    if (S.want.empty()) return fail(BPSW_ERR_DEVICE, "matesw_group: replay stalled");  // cannot happen
  }
  uint64_t wasted = 0;
  for (uint8_t u : S.used) wasted += u ? 0 : 1;
  c->stats.sw_speculated += speculated; c->stats.sw_replayed_rounds += rounds; c->stats.sw_wasted += wasted;

  // ---- 4. output: the lists of the untouched pairs come out exactly as they went in (every anchor of theirs is skipped or
  // has no usable window, and only an SW result can change a list): one memcpy per run between two touched pairs -----------
  const double t_o0 = stat_ms();
  c->stats.grp_calls++; c->stats.grp_pairs += (uint64_t)G_;
  c->stats.grp_plan_ms += t_planned - t_begin; c->stats.grp_pack_ms += t_pack; c->stats.grp_replay_ms += t_replay;
  int64_t total = nreg;
  for (size_t ti = 0; ti < nt; ++ti) {
    const int k = S.touched[ti];
    total += (int64_t)S.fin_cnt[2 * ti] + S.fin_cnt[2 * ti + 1] - g->reg_cnt[2 * k] - g->reg_cnt[2 * k + 1];
  }
  *out_total = total;
  if (total > out_cap || (total > 0 && !out_regs)) return fail(BPSW_ERR_CAPACITY, "matesw_group: out_regs too small");
  if (G_ > 0) memcpy(out_cnt, g->reg_cnt, sizeof(int32_t) * 2 * (size_t)G_);
  int64_t at = 0, src_from = 0;  // output position; first input region not yet copied
  for (size_t ti = 0; ti <= nt; ++ti) {
    const int k = ti < nt ? S.touched[ti] : G_;
    const int64_t run_end = ti < nt ? S.t_base[4 * ti] : nreg;  // regions before pair k
    if (run_end > src_from) {
      memcpy(out_regs + at, g->regs + src_from, sizeof(Reg) * (size_t)(run_end - src_from));
      at += run_end - src_from;
    }
    if (ti == nt) break;
    for (int i = 0; i < 2; ++i) {
      const int32_t cnt = S.fin_cnt[2 * ti + (size_t)i];
      out_cnt[2 * k + i] = cnt;
      if (cnt) memcpy(out_regs + at, S.arena.data() + S.fin_off[2 * ti + (size_t)i], sizeof(Reg) * (size_t)cnt);
      at += cnt;
    }
    src_from = S.t_base[4 * ti + 1] + g->reg_cnt[2 * k + 1];  // first region of pair k + 1
  }
  c->stats.grp_out_ms += stat_ms() - t_o0;
  return BPSW_OK;
}
