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
#include <string.h>

#include <algorithm>
#include <vector>

#include "bpsw_internal.h"

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

inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t* dist) {  // native/bwamem_pair.c:27-34
  const bool r1 = b1 >= l_pac, r2 = b2 >= l_pac;
  const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
  *dist = p2 > b1 ? p2 - b1 : b1 - p2;
  return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

struct Group {
  const bpsw_opt_t* opt;
  const bpsw_rescue_group_t* g;
  int mode;
  std::vector<int64_t> reg_base, ref_base;  // per (k,i): first region / first anchor row
  std::vector<int32_t> job_of;              // per window x: job index, -1 = not launched
  std::vector<int32_t> results;             // 7 ints per launched job
  std::vector<uint8_t> used;                // per job: consumed by the replay
  const int64_t* ref_len;                   // per window x: g->ref_len, or derived from (rb, re) in coordinate mode
  std::vector<Reg> anchors[2];              // scratch of replay_pair, kept across pairs (no allocation per pair)
};

void skip_flags(const Group& G, const Reg& a, const Reg* mates, size_t n_mates, int skip[4]) {
  for (int r = 0; r < 4; ++r) skip[r] = G.g->pes[r].failed ? 1 : 0;
  for (size_t mi = 0; mi < n_mates; ++mi) {
    const Reg& m = mates[mi];
    int64_t dist;
    const int r = infer_dir(G.g->l_pac, a.rb, m.rb, &dist);
    if (G.mode == BPSW_RESCUE_SCALA) dist = (int64_t)(int32_t)dist;  // MemSamPe.scala:1137-1138 narrows to Int
    if (dist >= G.g->pes[r].low && dist <= G.g->pes[r].high) skip[r] = 1;
  }
}

inline bool window_ok(const Group& G, int64_t x) { return G.ref_len[x] == G.g->ref_re[x] - G.g->ref_rb[x]; }

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

// One anchor against the mate list.  Returns false when a needed SW result is missing (x appended to `missing`).
bool precompute(Group& G, const Reg& a, int l_ms, std::vector<Reg>& ma, int64_t xrow, std::vector<int64_t>& missing) {
  int skip[4];
  skip_flags(G, a, ma.data(), ma.size(), skip);
  if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return true;
  int n = 0;
  if (G.mode == BPSW_RESCUE_C) {
    for (int r = 0; r < 4; ++r) {
      if (skip[r]) continue;
      const int64_t x = xrow * 4 + r;
      if (window_ok(G, x)) {
        const int job = G.job_of[(size_t)x];
        if (job < 0) { missing.push_back(x); return false; }
        G.used[(size_t)job] = 1;
        Reg b;
        if (make_region(G, &G.results[7 * (size_t)job], r, l_ms, x, &b)) {
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
      const int job = G.job_of[(size_t)x];
      if (job < 0) { missing.push_back(x); return false; }
      G.used[(size_t)job] = 1;
      Reg b;
      if (make_region(G, &G.results[7 * (size_t)job], r, l_ms, x, &b)) upd.push_back(b);
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
bool replay_pair(Group& G, int k, std::vector<Reg> v[2], std::vector<int64_t>& missing) {
  const bpsw_rescue_group_t* g = G.g;
  std::vector<Reg>* tmp = G.anchors;
  for (int i = 0; i < 2; ++i) {
    const Reg* first = g->regs + G.reg_base[(size_t)(2 * k + i)];
    v[i].assign(first, first + g->reg_cnt[2 * k + i]);
    tmp[i].clear();
    for (const Reg& r : v[i])  // anchors: filtered copy taken before any rescue (native/bwamem_pair.c:126-131)
      if (r.score >= v[i][0].score - G.opt->pen_unpaired) tmp[i].push_back(r);
  }
  for (int i = 0; i < 2; ++i) {
    const int na = std::min<int>((int)tmp[i].size(), std::min<int>(G.opt->max_matesw, g->ref_cnt[2 * k + i]));
    for (int j = 0; j < na; ++j)
      if (!precompute(G, tmp[i][(size_t)j], g->seq_len[2 * k + !i], v[!i], G.ref_base[(size_t)(2 * k + i)] + j, missing))
        return false;
  }
  return true;
}

}  // namespace

namespace bpsw {
int sort_dedup_regs(std::vector<bpsw_alnreg_t>& v, float mask_level_redun, int mode) { return sort_dedup(v, mask_level_redun, mode); }
}  // namespace bpsw

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

  Group S;
  S.opt = opt; S.g = g; S.mode = mode;
  S.reg_base.resize(2 * (size_t)G_ + 1); S.ref_base.resize(2 * (size_t)G_ + 1);
  int64_t nreg = 0, nref = 0;
  for (int e = 0; e < 2 * G_; ++e) {
    if (g->reg_cnt[e] < 0 || g->ref_cnt[e] < 0 || g->seq_len[e] < 0) return fail(BPSW_ERR_ARG, "matesw_group: negative count");
    if ((uint64_t)(g->seq_off[e] + g->seq_len[e]) > g->seq_pool_bytes) return fail(BPSW_ERR_ARG, "matesw_group: mate outside seq_pool");
    S.reg_base[(size_t)e] = nreg; S.ref_base[(size_t)e] = nref;
    nreg += g->reg_cnt[e]; nref += g->ref_cnt[e];
  }
  // with ref_pool == NULL the windows are named by (rb, re) only and read from the 2-bit reference
  // loaded on the context; their lengths follow bnsGetSeq (util/BNTSeqUtil.scala:37-59)
  const bool pac_mode = g->ref_pool == nullptr && nref > 0;
  std::vector<int64_t> derived_len;
  if (pac_mode) {
    const uint8_t* d_pac = nullptr;
    long long loaded = 0;
    ref_snapshot(c, &d_pac, &loaded);
    if (loaded <= 0) return fail(BPSW_ERR_ARG, "matesw_group: ref_pool is null and no reference is loaded (bpsw_ref_load)");
    if (loaded != g->l_pac) return fail(BPSW_ERR_ARG, "matesw_group: l_pac differs from the loaded reference");
    derived_len.resize((size_t)(4 * nref));
    for (int64_t x = 0; x < 4 * nref; ++x) {
      int64_t b = g->ref_rb[x], e = g->ref_re[x], len = 0;
      if (!(b < 0 && e < 0)) {  // (-1,-1): failed orientation, MemSamPe.scala:1863-1868
        if (e < b) std::swap(b, e);
        if (e > (g->l_pac << 1)) e = g->l_pac << 1;
        if (b < 0) b = 0;
        len = e - b > 0 ? e - b : 0;
        if (!(b >= g->l_pac || e <= g->l_pac)) len = 0;  // bridging the strands: bnsGetSeq returns nothing
      }
      derived_len[(size_t)x] = len;
    }
    S.ref_len = derived_len.data();
  } else {
    for (int64_t x = 0; x < 4 * nref; ++x)
      if (g->ref_len[x] > 0 && (g->ref_off[x] < 0 || (uint64_t)(g->ref_off[x] + g->ref_len[x]) > g->ref_pool_bytes))
        return fail(BPSW_ERR_ARG, "matesw_group: window outside ref_pool");
    S.ref_len = g->ref_len;
  }
  S.job_of.assign((size_t)(4 * nref), -1);

  const bool rescue_on = (opt->flag & 0x20) == 0;  // MEM_F_NO_RESCUE, native/bwamem.h:18
  std::vector<int64_t> want;  // windows whose SW result the next GPU round computes
  // pairs with at least one window to align; the lists of the others come out exactly as they went in (every anchor of
  // theirs is skipped or has no usable window, and only an SW result can change a list)
  std::vector<uint8_t> touched((size_t)G_, 0);
  if (rescue_on) {
    // ---- 1. speculate against the initial lists --------------------------------------------
    for (int k = 0; k < G_; ++k) {
      const Reg* init[2] = {g->regs + S.reg_base[(size_t)(2 * k)], g->regs + S.reg_base[(size_t)(2 * k + 1)]};
      const int n_init[2] = {g->reg_cnt[2 * k], g->reg_cnt[2 * k + 1]};
      for (int i = 0; i < 2; ++i) {
        if (g->seq_len[2 * k + !i] < 1) continue;
        int j = 0;
        for (int ai = 0; ai < n_init[i]; ++ai) {
          const Reg& a = init[i][ai];
          if (!(a.score >= init[i][0].score - opt->pen_unpaired)) continue;
          if (j >= opt->max_matesw || j >= g->ref_cnt[2 * k + i]) break;
          int skip[4];
          skip_flags(S, a, init[!i], (size_t)n_init[!i], skip);
          const int64_t xrow = S.ref_base[(size_t)(2 * k + i)] + j;
          for (int r = 0; r < 4; ++r)
            if (!skip[r] && window_ok(S, xrow * 4 + r)) { want.push_back(xrow * 4 + r); touched[(size_t)k] = 1; }
          ++j;
        }
      }
    }
  }
  // window -> (pair, end): needed to find the mate of a job
  std::vector<int32_t> end_of_row((size_t)nref);
  for (int e = 0; e < 2 * G_; ++e)
    for (int64_t j = 0; j < g->ref_cnt[e]; ++j) end_of_row[(size_t)(S.ref_base[(size_t)e] + j)] = e;

  const int xtra_base = BPSW_KSW_XSUBO | BPSW_KSW_XSTART | (opt->min_seed_len * opt->a);
  // final lists, in the order the pairs complete: one arena, (offset, count) per end
  std::vector<Reg> arena;
  std::vector<int64_t> fin_off(2 * (size_t)G_, -1);  // -1: the input list, untouched
  std::vector<int32_t> fin_cnt(2 * (size_t)G_, 0);
  std::vector<uint8_t> done((size_t)G_, 0);
  std::vector<Reg> v[2];
  uint64_t rounds = 0, speculated = want.size();
  for (;;) {
    // ---- 2. one flat GPU batch ----------------------------------------------------------------
    if (!want.empty()) {
      std::sort(want.begin(), want.end());
      want.erase(std::unique(want.begin(), want.end()), want.end());
      const size_t nj = want.size();
      std::vector<int32_t> q_len(nj), t_len(nj);
      std::vector<int64_t> q_off(nj), t_off(nj);
      std::vector<uint8_t> q_rev(nj);
      // only the mates some job aligns travel to the device (a tenth of the group's reads), not the whole read pool
      std::vector<uint8_t> qpool;
      std::vector<int64_t> mate_slot(2 * (size_t)G_, -1);
      std::vector<uint8_t> tpool;
      size_t tbytes = 0;
      if (!pac_mode)
        for (size_t t = 0; t < nj; ++t) tbytes += ((size_t)g->ref_len[want[t]] + 15) & ~(size_t)15;
      tpool.resize(tbytes ? tbytes : 16);
      size_t at = 0;
      for (size_t t = 0; t < nj; ++t) {
        const int64_t x = want[t];
        const int e = end_of_row[(size_t)(x >> 2)], mate = e ^ 1, r = (int)(x & 3);
        q_len[t] = g->seq_len[mate];
        if (mate_slot[(size_t)mate] < 0) {
          mate_slot[(size_t)mate] = (int64_t)qpool.size();
          qpool.insert(qpool.end(), g->seq_pool + g->seq_off[mate], g->seq_pool + g->seq_off[mate] + g->seq_len[mate]);
          qpool.resize((qpool.size() + 15) & ~(size_t)15, 0);
        }
        q_off[t] = mate_slot[(size_t)mate];
        q_rev[t] = ((r >> 1) != (r & 1)) ? 1 : 0;  // native/bwamem_pair.c:177
        t_len[t] = (int32_t)S.ref_len[x];
        if (pac_mode) {  // window_ok held, so the window starts at ref_rb[x] unclamped
          t_off[t] = g->ref_rb[x];
          continue;
        }
        t_off[t] = (int64_t)at;
        memcpy(tpool.data() + at, g->ref_pool + g->ref_off[x], (size_t)g->ref_len[x]);
        at += ((size_t)g->ref_len[x] + 15) & ~(size_t)15;
      }
      bpsw_sw_jobs_t jobs;
      memset(&jobs, 0, sizeof jobs);
      jobs.n = (int32_t)nj; jobs.xtra = xtra_base;  // KSW_XBYTE is ignored by SWAlign
      jobs.q_len = q_len.data(); jobs.t_len = t_len.data(); jobs.q_off = q_off.data(); jobs.t_off = t_off.data();
      if (qpool.empty()) qpool.resize(16, 0);
      jobs.q_rev = q_rev.data(); jobs.q_pool = qpool.data(); jobs.t_pool = pac_mode ? nullptr : tpool.data();
      jobs.q_pool_bytes = qpool.size(); jobs.t_pool_bytes = pac_mode ? 0 : tpool.size();
      const size_t first = S.results.size() / 7;
      S.results.resize(7 * (first + nj));
      S.used.resize(first + nj, 0);
      int rc = run_sw_jobs_host(c, opt, &jobs, S.results.data() + 7 * first);
      if (rc != BPSW_OK) return rc;
      for (size_t t = 0; t < nj; ++t) S.job_of[(size_t)want[t]] = (int32_t)(first + t);
      want.clear();
    }
    // ---- 3. replay -----------------------------------------------------------------------------
    bool all_done = true;
    for (int k = 0; k < G_; ++k) {
      if (done[(size_t)k]) continue;
      if (!rescue_on || !touched[(size_t)k]) {
        fin_cnt[(size_t)(2 * k)] = g->reg_cnt[2 * k];
        fin_cnt[(size_t)(2 * k + 1)] = g->reg_cnt[2 * k + 1];
        done[(size_t)k] = 1;
        continue;
      }
      if (!replay_pair(S, k, v, want)) {
        all_done = false;
        continue;
      }
      for (int i = 0; i < 2; ++i) {
        fin_off[(size_t)(2 * k + i)] = (int64_t)arena.size();
        fin_cnt[(size_t)(2 * k + i)] = (int32_t)v[i].size();
        arena.insert(arena.end(), v[i].begin(), v[i].end());
      }
      done[(size_t)k] = 1;
    }
    if (all_done) break;
    ++rounds;
    if (want.empty()) return fail(BPSW_ERR_DEVICE, "matesw_group: replay stalled");  // cannot happen
  }
  uint64_t wasted = 0;
  for (uint8_t u : S.used) wasted += u ? 0 : 1;
  c->stats.sw_speculated += speculated; c->stats.sw_replayed_rounds += rounds; c->stats.sw_wasted += wasted;

  int64_t total = 0;
  for (int e = 0; e < 2 * G_; ++e) {
    out_cnt[e] = fin_cnt[(size_t)e];
    total += out_cnt[e];
  }
  *out_total = total;
  if (total > out_cap || (total > 0 && !out_regs)) return fail(BPSW_ERR_CAPACITY, "matesw_group: out_regs too small");
  int64_t at = 0;
  for (int e = 0; e < 2 * G_; ++e) {
    const Reg* src = fin_off[(size_t)e] < 0 ? g->regs + S.reg_base[(size_t)e] : arena.data() + fin_off[(size_t)e];
    if (fin_cnt[(size_t)e]) memcpy(out_regs + at, src, sizeof(Reg) * (size_t)fin_cnt[(size_t)e]);
    at += fin_cnt[(size_t)e];
  }
  return BPSW_OK;
}
