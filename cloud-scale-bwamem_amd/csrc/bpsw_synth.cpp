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

// ---- the same tasks over a REFERENCE ("windows are generated on demand from a counter-based hash") ------------
// bpsw_synth_hash_pac fills a 2-bit .pac (BWA layout: base k = pac[k>>2] >> ((~k&3)<<1) & 3) of l_pac i.i.d. bases from a
// counter hash (splitmix64 of seed + k/32: 32 bases per word), so a reference of any length costs no RNG state and any window
// can be regenerated from its coordinates.
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
void bpsw_synth_hash_pac(int64_t l_pac, uint64_t seed, uint8_t* pac) {
  const int64_t n_bytes = (l_pac + 3) / 4;
  for (int64_t w = 0; 8 * w < n_bytes; ++w) {
    const uint64_t h = splitmix64(seed + (uint64_t)w);
    for (int b = 0; b < 8 && 8 * w + b < n_bytes; ++b) pac[8 * w + b] = (uint8_t)(h >> (8 * b));
  }
  if (l_pac & 3) pac[n_bytes - 1] &= (uint8_t)(0xff << (2 * (4 - (l_pac & 3))));  // bases past l_pac are zero
}
static inline int pac_base2(const uint8_t* pac, int64_t l_pac, int64_t k) {  // doubled coordinates, bnsGetSeq (BNTSeqUtil.scala:56-73)
  const bool rev = k >= l_pac;
  const int64_t f = rev ? 2 * l_pac - 1 - k : k;
  const int b = (pac[f >> 2] >> ((~f & 3) << 1)) & 3;
  return rev ? 3 - b : b;
}

// bpsw_synth_ext_tasks with every read drawn from the reference `pac` (uniform position, either strand): the same seeds, the
// same flanks, emitted BOTH as byte tasks (left_r_off / right_r_off: the target flanks copied out of the reference, what the
// Scala driver ships today) and as coordinates (seed_rbeg in [0, 2*l_pac), seed_len: what a coordinate batch ships instead,
// include/bpsw.h "wire format 2").  A read's window lies on one strand, so no flank bridges the two.
int bpsw_synth_ext_tasks_ref(const bpsw_synth_ext_cfg_t* c, const uint8_t* pac, int64_t l_pac, int32_t* left_qlen, int32_t* left_rlen,
                             int32_t* right_qlen, int32_t* right_rlen, int64_t* left_q_off, int64_t* left_r_off, int64_t* right_q_off,
                             int64_t* right_r_off, int32_t* reg_score, int32_t* q_beg, int32_t* h0, int32_t* idx, int64_t* seed_rbeg,
                             int32_t* seed_len, uint8_t* pool, size_t pool_cap, size_t* pool_used) {
  Rng g(c->seed);
  const int L = c->read_len;
  const int flank = L + cal_max_gap(L, c->a, c->o_del, c->e_del, c->o_ins, c->e_ins, c->w) + 8;
  const size_t win = (size_t)L * 3 + 2 * (size_t)flank + 64;
  if (l_pac < (int64_t)win + 2) return -1;
  std::vector<uint8_t> ref(win), read;
  std::vector<int> rpos;
  std::vector<Seed> seeds;
  size_t used = 0;
  int nt = 0;
  for (int rd = 0; rd < c->n_reads; ++rd) {
    const int64_t strand = (g.next() >> 63) ? l_pac : 0;
    const int64_t w0 = strand + (int64_t)(g.next() % (uint64_t)(l_pac - (int64_t)win));  // the window, in doubled coordinates
    for (size_t k = 0; k < win; ++k) ref[k] = (uint8_t)pac_base2(pac, l_pac, w0 + (int64_t)k);
    const bool tail = g.uni() < c->tail_frac;
    const double sub = tail ? c->tail_sub_rate : c->sub_rate, ind = tail ? c->tail_indel_rate : c->indel_rate;
    const int origin = flank;
    make_read(g, ref.data() + origin, (int)ref.size() - origin - flank, L, sub, ind, c->n_rate, read, rpos);
    seeds.clear();
    for (int i = 0; i < L;) {
      if (rpos[i] < 0) { ++i; continue; }
      int j = i + 1;
      while (j < L && rpos[j] == rpos[j - 1] + 1) ++j;
      if (j - i >= c->min_seed_len) seeds.push_back({i, origin + rpos[i], j - i});
      i = j;
    }
    if (seeds.empty()) continue;
    std::stable_sort(seeds.begin(), seeds.end(), [](const Seed& x, const Seed& y) { return x.len > y.len; });
    int n_emit = 1;
    if (c->second_seed && seeds.size() > 1 && seeds[0].len * 5 < L * 2) n_emit = 2;
    for (int e = 0; e < n_emit; ++e) {
      const Seed s = seeds[e];
      const int lq = s.qb, rq = L - (s.qb + s.len);
      if (lq == 0 && rq == 0) continue;
      int lr = lq ? lq + cal_max_gap(lq, c->a, c->o_del, c->e_del, c->o_ins, c->e_ins, c->w) : 0;
      int rr = rq ? rq + cal_max_gap(rq, c->a, c->o_del, c->e_del, c->o_ins, c->e_ins, c->w) : 0;
      if (lr > s.rb) lr = s.rb;
      if (rr > (int)ref.size() - (s.rb + s.len)) rr = (int)ref.size() - (s.rb + s.len);
      const size_t need = (size_t)lq + lr + rq + rr;
      if (used + need > pool_cap) return -1;
      left_qlen[nt] = lq; left_rlen[nt] = lr; right_qlen[nt] = rq; right_rlen[nt] = rr;
      left_q_off[nt] = (int64_t)used;
      for (int i = 0; i < lq; ++i) pool[used++] = read[lq - 1 - i];
      left_r_off[nt] = (int64_t)used;
      for (int i = 0; i < lr; ++i) pool[used++] = ref[s.rb - 1 - i];
      right_q_off[nt] = (int64_t)used;
      for (int i = 0; i < rq; ++i) pool[used++] = read[s.qb + s.len + i];
      right_r_off[nt] = (int64_t)used;
      for (int i = 0; i < rr; ++i) pool[used++] = ref[s.rb + s.len + i];
      reg_score[nt] = s.len * c->a; h0[nt] = s.len * c->a; q_beg[nt] = s.qb; idx[nt] = rd;
      seed_rbeg[nt] = w0 + s.rb; seed_len[nt] = s.len;
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

// ---- pair-end rescue groups (boundary 1), the flat layout of bpsw_rescue_group_t -------------------------------------
// Same model as bpsw_hip/synth.py::rescue_group without a backing reference (FR library, insert ~ N(400, 50^2), windows of
// random bases with the mutated mate planted where the insert puts it), written here because bench.py needs a million
// distinct pairs per step and the Python loop makes two thousand a second.
namespace {
struct SynReg { int64_t rb, re; int32_t qb, qe, score, truesc, sub, csub, sub_n, w, seedcov, secondary; uint64_t hash; };
static_assert(sizeof(SynReg) == 64, "bpsw_alnreg_t layout");
double gauss(Rng& g) {  // Box-Muller, one value per call
  double u1 = g.uni(), u2 = g.uni();
  if (u1 < 1e-300) u1 = 1e-300;
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
}  // namespace

extern "C" {

typedef struct {
  uint64_t seed;
  int32_t n_pairs, read_len;
  int64_t l_pac;
  double p_resc, sub_rate, indel_rate, p_multi_anchor, p_wrong_mate;
  int32_t max_matesw, pen_unpaired;
  int32_t low, high;  // the FR insert-size bounds (the other three orientations are `failed`)
  double avg, std;
} bpsw_synth_group_cfg_t;

// Capacities the caller must provide for n pairs of read length L:
//   seq_*: 2n, seq_pool: 2n * align16(L); reg_cnt/ref_cnt: 2n; regs: 6n; ref_rb/re/len/off: 16n; ref_pool: 4n * align16(high - low + 2L)
// out_counts[4] = {regs, anchor rows, seq_pool bytes, ref_pool bytes}.  Returns 0, or -1 when a pool is too small.
int bpsw_synth_rescue_group(const bpsw_synth_group_cfg_t* c, int32_t* seq_len, int64_t* seq_off, uint8_t* seq_pool, size_t seq_cap,
                            int32_t* reg_cnt, void* regs_v, size_t regs_cap, int32_t* ref_cnt, int64_t* ref_rb, int64_t* ref_re,
                            int64_t* ref_len, int64_t* ref_off, size_t rows_cap, uint8_t* ref_pool, size_t ref_cap,
                            int64_t out_counts[4]) {
  SynReg* regs = (SynReg*)regs_v;
  Rng g(c->seed);
  const int L = c->read_len;
  const int64_t l2 = c->l_pac << 1;
  std::vector<uint8_t> clean[2], rd[2], tmp;
  std::vector<int> rpos;
  std::vector<uint8_t> src((size_t)L * 2 + 64);
  size_t seq_at = 0, ref_at = 0, nreg = 0, nrow = 0;
  auto mk = [&](int64_t rb, int score, int qb, int qe) {
    SynReg r;
    memset(&r, 0, sizeof r);
    r.rb = rb; r.re = rb + (qe - qb); r.qb = qb; r.qe = qe; r.score = score; r.truesc = score; r.w = 100;
    r.seedcov = (qe - qb) / 2; r.secondary = -1; r.hash = g.next() >> 2;
    return r;
  };
  for (int k = 0; k < c->n_pairs; ++k) {
    const int64_t P = 2000 + (int64_t)(g.next() % (uint64_t)(c->l_pac - 5000));
    int ins = (int)(c->avg + c->std * gauss(g));
    if (ins < c->low + 20) ins = c->low + 20;
    if (ins > c->high - 20) ins = c->high - 20;
    for (int i = 0; i < 2; ++i) {
      // a clean locus with some spare bases behind it for deletions, mutated into a read of exactly L bases
      for (auto& b : src) b = (uint8_t)g.base();
      clean[i].assign(src.begin(), src.begin() + L);
      make_read(g, src.data(), (int)src.size(), L, c->sub_rate, c->indel_rate, 0.0, tmp, rpos);
      rd[i] = tmp;
    }
    for (int i = 0; i < L / 2; ++i) std::swap(rd[1][(size_t)i], rd[1][(size_t)(L - 1 - i)]);  // end 1 is sequenced from the reverse strand
    for (auto& b : rd[1]) b = b < 4 ? (uint8_t)(3 - b) : b;
    const int64_t true_rb[2] = {P, l2 - (P + ins)};
    bool have[2] = {true, true};
    if (g.uni() < c->p_resc) have[g.below(2)] = false;
    SynReg er[2][3];
    int ne[2] = {0, 0};
    for (int i = 0; i < 2; ++i) {
      if (have[i]) {
        er[i][ne[i]++] = mk(true_rb[i], L - g.below(8), 0, L);
        if (g.uni() < c->p_multi_anchor) er[i][ne[i]++] = mk(true_rb[i] + 1 + g.below(3), er[i][0].score - g.below(c->pen_unpaired), 0, L);
        if (g.uni() < 0.15) er[i][ne[i]++] = mk((int64_t)(g.next() % (uint64_t)(l2 - L)), er[i][0].score - c->pen_unpaired - 5, 10, L - 20);
      } else if (g.uni() < c->p_wrong_mate) {
        er[i][ne[i]++] = mk((int64_t)(g.next() % (uint64_t)(l2 - L)), L / 2, 0, L / 2 + 10);
      }
      std::stable_sort(er[i], er[i] + ne[i], [](const SynReg& x, const SynReg& y) { return x.score > y.score; });
    }
    const size_t Lp = ((size_t)L + 15) & ~(size_t)15;
    for (int i = 0; i < 2; ++i) {
      if (seq_at + Lp > seq_cap) return -1;
      seq_len[2 * k + i] = L; seq_off[2 * k + i] = (int64_t)seq_at;
      memcpy(seq_pool + seq_at, rd[i].data(), (size_t)L);
      memset(seq_pool + seq_at + L, 0, Lp - (size_t)L);
      seq_at += Lp;
    }
    for (int i = 0; i < 2; ++i) {
      if (nreg + (size_t)ne[i] > regs_cap) return -1;
      reg_cnt[2 * k + i] = ne[i];
      for (int j = 0; j < ne[i]; ++j) regs[nreg++] = er[i][j];
      int na = 0;
      for (int j = 0; j < ne[i] && na < c->max_matesw; ++j)
        if (er[i][j].score >= er[i][0].score - c->pen_unpaired) ++na;
      ref_cnt[2 * k + i] = na;
      if (nrow + (size_t)na > rows_cap) return -1;
      int ja = 0;
      for (int j = 0; j < ne[i] && ja < na; ++j) {
        if (!(er[i][j].score >= er[i][0].score - c->pen_unpaired)) continue;
        const SynReg& a = er[i][j];
        for (int r = 0; r < 4; ++r) {
          const size_t x = 4 * nrow + (size_t)r;
          if (r != 1) { ref_rb[x] = -1; ref_re[x] = -1; ref_len[x] = 0; ref_off[x] = 0; continue; }  // failed orientations
          // getAlnRegRefJNI, MemSamPe.scala:1810-1878: r = 1 is reversed and "larger"
          int64_t rb = a.rb + c->low - L, re = a.rb + c->high;
          if (rb < 0) rb = 0;
          if (re > l2) re = l2;
          const int64_t n = re > rb ? re - rb : 0;
          const size_t np = ((size_t)n + 15) & ~(size_t)15;
          if (ref_at + np > ref_cap) return -1;
          uint8_t* w = ref_pool + ref_at;
          for (int64_t t = 0; t < n; ++t) w[t] = (uint8_t)g.base();
          memset(w + n, 0, np - (size_t)n);
          const int64_t d = a.rb - true_rb[i];
          if (d > -8 && d < 8 && n >= L) {  // the mate lies `ins - L` past the true anchor start, on the anchor's strand
            const int64_t off = (a.rb - rb) + (ins - L) + (true_rb[i] - a.rb);
            if (off >= 0 && off <= n - L) {
              const std::vector<uint8_t>& cm = clean[1 - i];
              for (int t = 0; t < L; ++t) w[off + t] = i == 0 ? cm[(size_t)t] : (uint8_t)(3 - cm[(size_t)(L - 1 - t)]);
            }
          }
          ref_rb[x] = rb; ref_re[x] = re; ref_len[x] = n; ref_off[x] = (int64_t)ref_at;
          ref_at += np;
        }
        ++nrow; ++ja;
      }
    }
  }
  out_counts[0] = (int64_t)nreg; out_counts[1] = (int64_t)nrow; out_counts[2] = (int64_t)seq_at; out_counts[3] = (int64_t)ref_at;
  return 0;
}

}  // extern "C"
