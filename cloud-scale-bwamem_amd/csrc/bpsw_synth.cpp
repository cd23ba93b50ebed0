// bpsw_synth.cpp -- seeded synthetic workloads for the tests and bench.py (libbpsw_synth.so).
//
// No genomes or FASTQ files are available offline, so the BASELINE.json configs are concretised as
// SURVEY.md 8(d) prescribes: xorshift64* PRNG, i.i.d. uniform ACGT reference, reads = substrings with
// substitutions / geometric-length indels / rare N, seeds = longest exact-match runs >= 19 bp (a
// stand-in for SMEM seeding, which is upstream of the SW path), tasks emitted the way
// memChainToAlnBatched does (MemChainToAlignBatched.scala:500-562): left query/reference reversed,
// reference flanks of qlen + calMaxGap(qlen) (MemChainToAlignBatched.scala:625-668).
// Host-only helper: it produces inputs, it is not part of the measured path.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed ? seed : 0x9E3779B97F4A7C15ull) {}
  uint64_t next() {  // xorshift64*
    s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
    return s * 0x2545F4914F6CDD1Dull;
  }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  int below(int n) { return (int)(next() % (uint64_t)n); }
  int base() { return (int)(next() >> 62); }
};

int cal_max_gap(int qlen, int a, int o_del, int e_del, int o_ins, int e_ins, int w) {  // C2AB:625-641
  const int l_del = (int)((double)(qlen * a - o_del) / (double)e_del + 1.0);
  const int l_ins = (int)((double)(qlen * a - o_ins) / (double)e_ins + 1.0);
  int len = l_del > l_ins ? l_del : l_ins;
  if (len <= 1) len = 1;
  const int tmp = w << 1;
  return len < tmp ? len : tmp;
}

// Mutate ref[0..) into a read of exactly `len` bases.  ref_pos[i] = reference index the read base i
// was copied from, or -1 for inserted / substituted / N bases.
void make_read(Rng& g, const uint8_t* ref, int ref_avail, int len, double sub, double indel, double nrate,
               std::vector<uint8_t>& read, std::vector<int>& ref_pos) {
  read.clear(); ref_pos.clear();
  int r = 0;
  while ((int)read.size() < len && r < ref_avail) {
    const double u = g.uni();
    if (u < indel * 0.5) {  // insertion, geometric length p = 0.7
      do { read.push_back((uint8_t)g.base()); ref_pos.push_back(-1); } while (g.uni() > 0.7 && (int)read.size() < len);
    } else if (u < indel) {  // deletion
      do { ++r; } while (g.uni() > 0.7 && r < ref_avail);
    } else if (u < indel + sub) {
      read.push_back((uint8_t)((ref[r] + 1 + g.below(3)) & 3)); ref_pos.push_back(-1); ++r;
    } else if (u < indel + sub + nrate) {
      read.push_back(4); ref_pos.push_back(-1); ++r;
    } else {
      read.push_back(ref[r]); ref_pos.push_back(r); ++r;
    }
  }
  while ((int)read.size() < len) { read.push_back((uint8_t)g.base()); ref_pos.push_back(-1); }
}

struct Seed { int qb, rb, len; };

}  // namespace

extern "C" {

typedef struct {
  uint64_t seed;
  int32_t n_reads, read_len;
  double sub_rate, indel_rate, n_rate;
  double tail_frac, tail_sub_rate, tail_indel_rate;  // a fraction of reads drawn at a higher error rate
  int32_t a, o_del, e_del, o_ins, e_ins, w, min_seed_len;
  int32_t second_seed;  // emit a task for the second-longest seed when the longest covers < 40 % of the read
} bpsw_synth_ext_cfg_t;

// Emits SoA extension tasks (the fields of ExtParam, datatype/ExtensionParameters.scala:21-45).
// Arrays must hold 2*n_reads tasks; pool must hold pool_cap bytes.  Returns the task count or -1.
int bpsw_synth_ext_tasks(const bpsw_synth_ext_cfg_t* c, int32_t* left_qlen, int32_t* left_rlen, int32_t* right_qlen,
                         int32_t* right_rlen, int64_t* left_q_off, int64_t* left_r_off, int64_t* right_q_off,
                         int64_t* right_r_off, int32_t* reg_score, int32_t* q_beg, int32_t* h0, int32_t* idx,
                         uint8_t* pool, size_t pool_cap, size_t* pool_used) {
  Rng g(c->seed);
  const int L = c->read_len;
  const int flank = L + cal_max_gap(L, c->a, c->o_del, c->e_del, c->o_ins, c->e_ins, c->w) + 8;
  std::vector<uint8_t> ref((size_t)L * 3 + 2 * (size_t)flank + 64), read;
  std::vector<int> rpos;
  std::vector<Seed> seeds;
  size_t used = 0;
  int nt = 0;
  for (int rd = 0; rd < c->n_reads; ++rd) {
    for (auto& b : ref) b = (uint8_t)g.base();
    const bool tail = g.uni() < c->tail_frac;
    const double sub = tail ? c->tail_sub_rate : c->sub_rate, ind = tail ? c->tail_indel_rate : c->indel_rate;
    const int origin = flank;  // the read starts at ref[origin]
    make_read(g, ref.data() + origin, (int)ref.size() - origin - flank, L, sub, ind, c->n_rate, read, rpos);
    // exact-match runs: consecutive read bases copied from consecutive reference positions
    seeds.clear();
    for (int i = 0; i < L;) {
      if (rpos[i] < 0) { ++i; continue; }
      int j = i + 1;
      while (j < L && rpos[j] == rpos[j - 1] + 1) ++j;
      if (j - i >= c->min_seed_len) seeds.push_back({i, origin + rpos[i], j - i});
      i = j;
    }
    if (seeds.empty()) continue;  // unmapped: no chain, no task
    std::stable_sort(seeds.begin(), seeds.end(), [](const Seed& x, const Seed& y) { return x.len > y.len; });
    int n_emit = 1;
    if (c->second_seed && seeds.size() > 1 && seeds[0].len * 5 < L * 2) n_emit = 2;
    for (int e = 0; e < n_emit; ++e) {
      const Seed s = seeds[e];
      const int lq = s.qb, rq = L - (s.qb + s.len);
      if (lq == 0 && rq == 0) continue;  // seed spans the read: no ExtParam (C2AB:500)
      int lr = lq ? lq + cal_max_gap(lq, c->a, c->o_del, c->e_del, c->o_ins, c->e_ins, c->w) : 0;
      int rr = rq ? rq + cal_max_gap(rq, c->a, c->o_del, c->e_del, c->o_ins, c->e_ins, c->w) : 0;
      if (lr > s.rb) lr = s.rb;
      if (rr > (int)ref.size() - (s.rb + s.len)) rr = (int)ref.size() - (s.rb + s.len);
      const size_t need = (size_t)lq + lr + rq + rr;
      if (used + need > pool_cap) return -1;
      left_qlen[nt] = lq; left_rlen[nt] = lr; right_qlen[nt] = rq; right_rlen[nt] = rr;
      left_q_off[nt] = (int64_t)used;
      for (int i = 0; i < lq; ++i) pool[used++] = read[lq - 1 - i];  // reversed, C2AB:505-510
      left_r_off[nt] = (int64_t)used;
      for (int i = 0; i < lr; ++i) pool[used++] = ref[s.rb - 1 - i];  // reversed, C2AB:511-517
      right_q_off[nt] = (int64_t)used;
      for (int i = 0; i < rq; ++i) pool[used++] = read[s.qb + s.len + i];
      right_r_off[nt] = (int64_t)used;
      for (int i = 0; i < rr; ++i) pool[used++] = ref[s.rb + s.len + i];
      reg_score[nt] = s.len * c->a; h0[nt] = s.len * c->a; q_beg[nt] = s.qb; idx[nt] = rd;
      ++nt;
    }
  }
  if (pool_used) *pool_used = used;
  return nt;
}

typedef struct {
  uint64_t seed;
  int32_t n_jobs, read_len;
  int32_t win_min, win_max;       // window length = read_len + uniform[win_min, win_max]
  double sub_rate, indel_rate, n_rate;
  double unrelated_frac;          // jobs whose window does not contain the mate
  double decoy_frac;              // jobs with a second partial copy (exercises the second-best logic)
  double rev_frac;                // jobs that ask for the reverse complement of the stored mate
} bpsw_synth_sw_cfg_t;

// Emits SWAlign2 jobs: mate bytes in q_pool (stored in the orientation before the optional
// reverse-complement), window bytes in t_pool.  Returns n_jobs or -1 when a pool is too small.
int bpsw_synth_sw_jobs(const bpsw_synth_sw_cfg_t* c, int32_t* q_len, int32_t* t_len, int64_t* q_off, int64_t* t_off,
                       uint8_t* q_rev, uint8_t* q_pool, size_t q_cap, uint8_t* t_pool, size_t t_cap, size_t* q_used,
                       size_t* t_used) {
  Rng g(c->seed);
  const int L = c->read_len;
  std::vector<uint8_t> win, read, mate(L);
  std::vector<int> rpos;
  size_t qu = 0, tu = 0;
  for (int jb = 0; jb < c->n_jobs; ++jb) {
    const int wl = L + c->win_min + (c->win_max > c->win_min ? g.below(c->win_max - c->win_min + 1) : 0);
    win.resize((size_t)wl);
    for (auto& b : win) b = (uint8_t)g.base();
    const bool unrelated = g.uni() < c->unrelated_frac;
    if (unrelated) {
      read.resize(L);
      for (auto& b : read) b = (uint8_t)g.base();
    } else {
      const int span = L + L / 8 + 4;
      const int pos = wl > span ? g.below(wl - span) : 0;
      make_read(g, win.data() + pos, wl - pos, L, c->sub_rate, c->indel_rate, c->n_rate, read, rpos);
      if (g.uni() < c->decoy_frac) {  // copy a 25..84-base prefix or suffix of the read elsewhere in the window
        int dl = 25 + g.below(60);
        if (dl >= L) dl = L / 2;
        const bool suffix = g.below(2);
        const int dpos = g.below(wl - dl);
        for (int i = 0; i < dl; ++i) win[dpos + i] = suffix ? read[L - dl + i] : read[i];
      }
    }
    // `read` is the sequence the SW sees; store its reverse complement when the job sets q_rev
    const bool rev = g.uni() < c->rev_frac;
    for (int i = 0; i < L; ++i) mate[i] = rev ? (uint8_t)(read[L - 1 - i] < 4 ? 3 - read[L - 1 - i] : 4) : read[i];
    if (qu + (size_t)L > q_cap || tu + (size_t)wl > t_cap) return -1;
    q_len[jb] = L; t_len[jb] = wl; q_off[jb] = (int64_t)qu; t_off[jb] = (int64_t)tu; q_rev[jb] = rev ? 1 : 0;
    memcpy(q_pool + qu, mate.data(), (size_t)L); qu += (size_t)L;
    memcpy(t_pool + tu, win.data(), (size_t)wl); tu += (size_t)wl;
    // keep pool offsets 16-byte aligned so device rows can be fetched with wide loads
    qu = (qu + 15) & ~(size_t)15; tu = (tu + 15) & ~(size_t)15;
  }
  if (q_used) *q_used = qu;
  if (t_used) *t_used = tu;
  return c->n_jobs;
}

}  // extern "C"
