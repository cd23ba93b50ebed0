// bpsw_tail.cpp -- worker2's tail behind the C ABI: bpsw_reg2aln_batch, bpsw_sam_pe_batch.
//
// memSamPeGroupRest (worker2/MemSamPe.scala:1390-1612 == mem_sam_pe after the rescue, native/bwamem_pair.c:385-452) is
// split in three passes so that every global alignment of a group goes to the GPU in ONE launch:
//   plan    per pair: memMarkPrimarySe, memPair, the paired / unpaired decision and its mapQ arithmetic; every region that
//           the reference would hand to memRegToAln becomes a job (a region that is asked for twice -- h[i] and the first
//           entry of memRegToSAMSe -- is one job: memRegToAln is a pure function of (read, region));
//   device  reg2aln_kernel (bpsw_reg2aln.hip): fix-xref, band inference, global alignments, NM/MD, position, CIGAR;
//   emit    per pair: flag / mapq / score / sub, then the SAM text of memAlnToSAM.
// PE = worker2/MemSamPe.scala, R2S = worker2/MemRegToADAMSAM.scala, MP = worker2/MemMarkPrimarySe.scala.
#include <math.h>
#include <string.h>
#if defined(__x86_64__)
#include <tmmintrin.h>  // pshufb: the base / quality columns of a SAM line, 16 characters at a time (host pass only)
#endif

#include <algorithm>
#include <chrono>
#include <string>
#include <utility>
#include <vector>

#include "bpsw_internal.h"

using namespace bpsw;

namespace {

int hip_fail(hipError_t e, const char* what) { return fail(BPSW_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); }
#define HIP_TRY(expr)                                 \
  do {                                                \
    hipError_t e_ = (expr);                           \
    if (e_ != hipSuccess) return hip_fail(e_, #expr); \
  } while (0)

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// the kernel stages at most this many CIGAR operations (CIG_LDS in bpsw_global_core.h)
constexpr int KERNEL_CIG_CAP = 512;

struct BnsView {
  long long l_pac = 0;
  const uint8_t* d_pac = nullptr;
  int n_seqs = 0;
  const long long* d_off = nullptr;
  const int32_t* d_len = nullptr;
  std::vector<long long> off;
  std::vector<int32_t> len;
  std::vector<std::string> name;
  RefHold hold;  // the reference and the contig table stay put while this view lives
};

int snapshot_bns(const bpsw_ctx* c, BnsView* v) {
  DeviceRef& r = device_ref(c->device);
  v->hold = RefHold(&r.gate);
  std::lock_guard<std::mutex> g(r.mu);
  if (r.l_pac <= 0 || !r.buf.ptr) return fail(BPSW_ERR_ARG, "tail: no reference loaded on this device (bpsw_ref_load)");
  if (r.ann_off.empty() || !r.ann.ptr) return fail(BPSW_ERR_ARG, "tail: no contig table loaded on this device (bpsw_bns_load)");
  v->l_pac = r.l_pac;
  v->d_pac = (const uint8_t*)r.buf.ptr;
  v->n_seqs = (int)r.ann_off.size();
  v->d_off = (const long long*)r.ann.ptr;
  v->d_len = (const int32_t*)((const char*)r.ann.ptr + 8 * r.ann_off.size());
  v->off = r.ann_off; v->len = r.ann_len; v->name = r.ann_name;
  return BPSW_OK;
}

int make_sw_scoring(const bpsw_opt_t* opt, SwScoring* sc) {
  if (!opt) return fail(BPSW_ERR_ARG, "tail: null options");
  if (opt->a < 1 || opt->o_del < 0 || opt->e_del < 1 || opt->o_ins < 0 || opt->e_ins < 1)
    return fail(BPSW_ERR_ARG, "tail: scoring must have a >= 1, non-negative gap opens and gap extensions >= 1");
  sc->mat = pack_mat(opt->mat);
  sc->a = opt->a; sc->b = opt->b;
  sc->o_del = opt->o_del; sc->e_del = opt->e_del; sc->o_ins = opt->o_ins; sc->e_ins = opt->e_ins;
  sc->xtra = 0;
  return BPSW_OK;
}

struct JobResult {
  Reg2AlnOut k;
  size_t cig_at = 0, md_at = 0;  // its CIGAR words / MD bytes in JobResults::cig / ::md (k.n_cigar, k.md_len of them)
};
struct JobResults {  // one allocation per kind and call, not two per job
  std::vector<JobResult> r;
  std::vector<uint32_t> cig;
  std::vector<char> md;
};

// One launch of reg2aln_kernel over `n` mapped jobs (regs[j] with rb, re >= 0); caller holds c->mu and has set the device.
int launch_jobs(bpsw_ctx* c, const SwScoring& sc, const bpsw_opt_t* opt, int flavour, const BnsView& bns, int n, const int32_t* read_len,
                const int64_t* read_off, const uint8_t* read_pool, size_t read_pool_bytes, const bpsw_alnreg_t* regs, int max_cigar,
                int max_md, Reg2AlnOut* out, uint32_t* out_cigar, uint8_t* out_md) {
  if (n == 0) return BPSW_OK;
  int mq = 0, mr = 0;
  size_t mz = 0;
  for (int j = 0; j < n; ++j) {
    const bpsw_alnreg_t& a = regs[j];
    const int lq = read_len[j];
    if (lq < 1 || lq > BPSW_R2A_MAX_QLEN) return fail(BPSW_ERR_LIMIT, "reg2aln: read length outside 1..BPSW_R2A_MAX_QLEN");
    if (read_off[j] < 0 || (unsigned long long)(read_off[j] + lq) > read_pool_bytes) return fail(BPSW_ERR_ARG, "reg2aln: read outside its pool");
    if (a.qb < 0 || a.qe < a.qb || a.qe > lq) return fail(BPSW_ERR_ARG, "reg2aln: region query range outside the read");
    if (a.rb < 0 || a.re < a.rb || a.re > (bns.l_pac << 1)) return fail(BPSW_ERR_ARG, "reg2aln: region reference range outside [0, 2*l_pac]");
    if (a.re - a.rb > BPSW_R2A_MAX_RLEN) return fail(BPSW_ERR_LIMIT, "reg2aln: region longer than BPSW_R2A_MAX_RLEN on the reference");
    const int rl = (int)(a.re - a.rb);
    mq = std::max(mq, lq);
    mr = std::max(mr, rl);
    mz = std::max(mz, (size_t)(a.qe - a.qb) * (size_t)rl);
  }
  const int qcap = (mq + 31) & ~31, rcap = (mr + 31) & ~31;
  const int md_cap = 2 * qcap + rcap + 32;  // every mismatch costs >= 2 bytes, every deleted base 1, plus the counts
  if (reg2aln_lds_per_wave(qcap, rcap, md_cap) * 4 > 64 * 1024) return fail(BPSW_ERR_LIMIT, "reg2aln: sequences too long for the LDS staging");
  const size_t o_len = 0, o_off = align16(4 * (size_t)n), o_regs = align16(o_off + 8 * (size_t)n);
  const size_t o_pool = align16(o_regs + sizeof(bpsw_alnreg_t) * (size_t)n);
  // ship only the bytes the jobs touch: reads are contiguous per job in the caller's pool, so copy the covering span
  long long lo = (long long)read_pool_bytes, hi = 0;
  for (int j = 0; j < n; ++j) { lo = std::min<long long>(lo, read_off[j]); hi = std::max<long long>(hi, read_off[j] + read_len[j]); }
  const size_t span = (size_t)(hi - lo);
  const size_t total = align16(o_pool + span);
  const size_t r_out = 0, r_cig = align16(sizeof(Reg2AlnOut) * (size_t)n), r_md = align16(r_cig + 4 * (size_t)n * (size_t)max_cigar);
  const size_t out_bytes = align16(r_md + (size_t)n * (size_t)max_md);
  const size_t z_per_wave = (mz + 255) & ~(size_t)255;
  HIP_TRY(c->h_stage_in.reserve(total));
  HIP_TRY(c->d_sw_in.reserve(total));
  HIP_TRY(c->h_stage_out.reserve(out_bytes));
  HIP_TRY(c->d_sw_out.reserve(out_bytes));
  HIP_TRY(c->d_gl_z.reserve(z_per_wave * (size_t)reg2aln_resident_waves(c->num_cu, qcap, rcap, md_cap)));
  uint8_t* h = (uint8_t*)c->h_stage_in.ptr;
  memcpy(h + o_len, read_len, 4 * (size_t)n);
  long long* ho = (long long*)(h + o_off);
  for (int j = 0; j < n; ++j) ho[j] = read_off[j] - lo;
  memcpy(h + o_regs, regs, sizeof(bpsw_alnreg_t) * (size_t)n);
  memcpy(h + o_pool, read_pool + lo, span);
  uint8_t* d = (uint8_t*)c->d_sw_in.ptr;
  Reg2AlnDev J;
  J.n = n; J.max_cigar = max_cigar; J.max_md = max_md; J.flavour = flavour; J.opt_w = opt->w; J.a = opt->a;
  J.read_len = (const int32_t*)(d + o_len); J.read_off = (const long long*)(d + o_off); J.read_pool = d + o_pool;
  J.regs = (const bpsw_alnreg_t*)(d + o_regs);
  J.pac = bns.d_pac; J.l_pac = bns.l_pac; J.n_seqs = bns.n_seqs; J.ann_off = bns.d_off; J.ann_len = bns.d_len;
  uint8_t* dout = (uint8_t*)c->d_sw_out.ptr;
  HIP_TRY(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(c->ev[6], c->stream));
  HIP_TRY(launch_reg2aln_kernel(J, sc, qcap, rcap, md_cap, z_per_wave, (Reg2AlnOut*)(dout + r_out), (uint32_t*)(dout + r_cig),
                                dout + r_md, (uint8_t*)c->d_gl_z.ptr, c->num_cu, c->stream));
  HIP_TRY(hipEventRecord(c->ev[7], c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_stage_out.ptr, dout, out_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, c->ev[6], c->ev[7]);
  c->last_tail_ms += ms;  // a call may launch twice (jobs whose CIGAR / MD outgrew the first, small, room)
  c->last_tail_jobs += n;  // distinct jobs = this minus last_tail_resubmitted
  c->have_tail_ev = true;
  const uint8_t* r = (const uint8_t*)c->h_stage_out.ptr;
  memcpy(out, r + r_out, sizeof(Reg2AlnOut) * (size_t)n);
  memcpy(out_cigar, r + r_cig, 4 * (size_t)n * (size_t)max_cigar);
  memcpy(out_md, r + r_md, (size_t)n * (size_t)max_md);
  return BPSW_OK;
}

// All jobs of a call, re-submitting the few whose CIGAR or MD did not fit the first, small, per-job room.
int run_jobs(bpsw_ctx* c, const SwScoring& sc, const bpsw_opt_t* opt, int flavour, const BnsView& bns, const std::vector<int32_t>& read_len,
             const std::vector<int64_t>& read_off, const uint8_t* read_pool, size_t read_pool_bytes,
             const std::vector<bpsw_alnreg_t>& regs, JobResults* res) {
  const int n = (int)regs.size();
  res->r.assign((size_t)n, JobResult());
  res->cig.clear(); res->md.clear();
  res->cig.reserve(4 * (size_t)n); res->md.reserve(16 * (size_t)n);
  std::vector<int> todo((size_t)n);
  for (int j = 0; j < n; ++j) todo[(size_t)j] = j;
  int max_cigar = 16, max_md = 64;
  while (!todo.empty()) {
    const int m = (int)todo.size();
    std::vector<int32_t> rl((size_t)m);
    std::vector<int64_t> ro((size_t)m);
    std::vector<bpsw_alnreg_t> rg((size_t)m);
    for (int t = 0; t < m; ++t) { rl[(size_t)t] = read_len[(size_t)todo[(size_t)t]]; ro[(size_t)t] = read_off[(size_t)todo[(size_t)t]]; rg[(size_t)t] = regs[(size_t)todo[(size_t)t]]; }
    std::vector<Reg2AlnOut> out((size_t)m);
    std::vector<uint32_t> cig((size_t)m * (size_t)max_cigar);
    std::vector<uint8_t> md((size_t)m * (size_t)max_md);
    int rc = launch_jobs(c, sc, opt, flavour, bns, m, rl.data(), ro.data(), read_pool, read_pool_bytes, rg.data(), max_cigar, max_md,
                         out.data(), cig.data(), md.data());
    if (rc != BPSW_OK) return rc;
    std::vector<int> again;
    for (int t = 0; t < m; ++t) {
      JobResult& r = res->r[(size_t)todo[(size_t)t]];
      r.k = out[(size_t)t];
      const bool fits = r.k.n_cigar <= max_cigar && r.k.md_len <= max_md;
      if (!fits && r.k.status == 0 && max_cigar < KERNEL_CIG_CAP + 2) { again.push_back(todo[(size_t)t]); continue; }
      if (!fits && r.k.status == 0) r.k.status = BPSW_ALN_OVERFLOW;
      if (r.k.status == 0 || r.k.status == BPSW_ALN_NOCIGAR) {
        r.cig_at = res->cig.size(); r.md_at = res->md.size();
        res->cig.insert(res->cig.end(), cig.begin() + (size_t)t * (size_t)max_cigar, cig.begin() + (size_t)t * (size_t)max_cigar + (size_t)std::max(r.k.n_cigar, 0));
        res->md.insert(res->md.end(), (const char*)md.data() + (size_t)t * (size_t)max_md, (const char*)md.data() + (size_t)t * (size_t)max_md + (size_t)std::max(r.k.md_len, 0));
      }
    }
    todo.swap(again);
    c->last_tail_resubmitted += (int)todo.size();
    max_cigar = std::min(max_cigar * 8, KERNEL_CIG_CAP + 2);
    max_md = std::min(max_md * 8, 4096 + 2 * BPSW_R2A_MAX_QLEN);
  }
  return BPSW_OK;
}

// ---- the scalar pieces of the tail -----------------------------------------------------------------------------------
inline uint64_t hash64(uint64_t key) {  // MP:111-122
  key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
  key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
  return key;
}

// memMarkPrimarySe, MP:37-109 (C: native/bwamem.c:444-477): sorts `a` and fills sub / sub_n / secondary / hash
void mark_primary(const bpsw_opt_t& o, const bpsw_tail_opt_t& t, std::vector<bpsw_alnreg_t>& a, int64_t id) {
  const int n = (int)a.size();
  if (n == 0) return;
  for (int i = 0; i < n; ++i) { a[(size_t)i].sub = 0; a[(size_t)i].secondary = -1; a[(size_t)i].hash = hash64((uint64_t)(id + i)); }
  const bool signed_hash = t.flavour == BPSW_TAIL_SCALA;  // sortBy(r => (-r.score, r.hash)) orders Longs, MP:62
  std::sort(a.begin(), a.end(), [signed_hash](const bpsw_alnreg_t& x, const bpsw_alnreg_t& y) {
    if (x.score != y.score) return x.score > y.score;
    return signed_hash ? (int64_t)x.hash < (int64_t)y.hash : x.hash < y.hash;  // hash64 is a bijection: no ties
  });
  const int gap = std::max(o.a + o.b, std::max(o.o_del + o.e_del, o.o_ins + o.e_ins));
  static thread_local std::vector<int> prim;
  prim.assign((size_t)n + 1, 0);  // the Scala's zero-filled z array, MP:46
  int np = 0;
  prim[(size_t)np++] = 0;
  for (int i = 1; i < n; ++i) {
    int k = 0;
    for (; k < np; ++k) {
      bpsw_alnreg_t& p = a[(size_t)prim[(size_t)k]];
      const bpsw_alnreg_t& q = a[(size_t)i];
      const int b_max = std::max(p.qb, q.qb), e_min = std::min(p.qe, q.qe);
      if (e_min <= b_max) continue;
      const int min_l = std::min(q.qe - q.qb, p.qe - p.qb);
      if ((float)(e_min - b_max) >= (float)min_l * t.mask_level) {
        if (p.sub == 0) p.sub = q.score;
        if (p.score - q.score <= gap) ++p.sub_n;
        break;
      }
    }
    if (k == np) prim[(size_t)np++] = i;
    else a[(size_t)i].secondary = t.flavour == BPSW_TAIL_C ? prim[(size_t)k] : prim[(size_t)k + 1];  // MP:93-101 reads z(k) after k += 1
  }
}

// log(l) for the alignment lengths memApproxMapqSe meets (one call per alignment: the same doubles as log(), computed once)
inline double log_of_len(int l) {
  static thread_local double tab[1024];
  if (l <= 0 || l >= 1024) return log((double)l);
  if (tab[l] == 0.0 && l != 1) tab[l] = log((double)l);
  return tab[l];
}

// memApproxMapqSe, R2S:568-604 (C: native/bwamem.c:845-872)
int approx_mapq(const bpsw_opt_t& o, const bpsw_tail_opt_t& t, const bpsw_alnreg_t& a) {
  int sub = a.sub > 0 ? a.sub : o.min_seed_len * o.a;
  if (t.flavour == BPSW_TAIL_C) sub = a.sub ? a.sub : o.min_seed_len * o.a;
  if (a.csub > sub) sub = a.csub;
  if (sub >= a.score) return 0;
  const int l = (a.qe - a.qb > a.re - a.rb) ? a.qe - a.qb : (int)(a.re - a.rb);
  const double identity = 1. - (double)(l * o.a - a.score) / (o.a + o.b) / l;
  int mapq;
  if (a.score == 0) {
    mapq = 0;
  } else if (t.mapq_coef_len > 0) {
    double tmp;
    if (t.flavour == BPSW_TAIL_C) tmp = l < t.mapq_coef_len ? 1. : t.mapq_coef_fac / log_of_len(l);
    else tmp = l > t.mapq_coef_len ? t.mapq_coef_fac / log_of_len(l) : 1.;  // R2S:586: differs from the C at l == mapQCoefLen
    tmp *= identity * identity;
    mapq = (int)(6.02 * (a.score - sub) / o.a * tmp * tmp + .499);
  } else {
    mapq = (int)(30.0 * (1. - (double)sub / a.score) * log((double)a.seedcov) + .499);
    if (identity < 0.95) mapq = (int)(mapq * identity * identity + .499);
  }
  if (a.sub_n > 0) mapq -= (int)(4.343 * log((double)(a.sub_n + 1)) + .499);
  return std::min(60, std::max(0, mapq));
}

inline int raw_mapq(int diff, int a) { return (int)(6.02 * diff / a + .499); }

int infer_dir(long long l_pac, long long b1, long long b2, long long* dist) {  // native/bwamem_pair.c:27-34, PE:1575-1594
  const bool r1 = b1 >= l_pac, r2 = b2 >= l_pac;
  const long long p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
  *dist = p2 > b1 ? p2 - b1 : b1 - p2;
  return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

struct PairScore { int score, sub, n_sub, z[2]; };

// The insert-size term of memPair's score depends on the orientation's statistics and the INTEGER distance only: an erfc and a
// log per candidate pair were half of the plan phase.  One table per orientation and calling thread over [low, high] (memPair
// never asks outside it), rebuilt when the statistics change; the same doubles as the expression computes.
inline double pair_penalty(const bpsw_pestat_t& pe, int dir, long long dist) {
  struct Tab { double avg = 0., std = 0.; long long low = 0, high = -1; std::vector<double> val; std::vector<uint8_t> have; };
  static thread_local Tab tabs[4];
  const auto direct = [&]() { const double ns = ((double)dist - pe.avg) / pe.std; return .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)); };
  if (dist < pe.low || dist > pe.high || (long long)pe.high - (long long)pe.low > (1 << 16)) return direct();
  Tab& T = tabs[dir & 3];
  if (T.avg != pe.avg || T.std != pe.std || T.low != pe.low || T.high != pe.high) {
    T.avg = pe.avg; T.std = pe.std; T.low = pe.low; T.high = pe.high;
    T.val.assign((size_t)(pe.high - pe.low + 1), 0.); T.have.assign(T.val.size(), 0);
  }
  const size_t at = (size_t)(dist - pe.low);
  if (!T.have[at]) { T.val[at] = direct(); T.have[at] = 1; }
  return T.val[at];
}

// memPair, PE:462-572 (C: native/bwamem_pair.c:298-357)
PairScore mem_pair(const bpsw_opt_t& o, const bpsw_tail_opt_t& t, long long l_pac, const bpsw_pestat_t pes[4],
                   const std::vector<bpsw_alnreg_t> a[2], int64_t id) {
  typedef std::pair<uint64_t, uint64_t> Key;  // (x, y), ordered like pair64_lt
  static thread_local std::vector<Key> v, u;
  v.clear(); u.clear();
  for (int r = 0; r < 2; ++r)
    for (size_t i = 0; i < a[r].size(); ++i) {
      const bpsw_alnreg_t& e = a[r][i];
      const uint64_t x = (uint64_t)(e.rb < l_pac ? e.rb : (l_pac << 1) - 1 - e.rb);
      const uint64_t y = (uint64_t)e.score << 32 | (uint64_t)(i << 2) | (uint64_t)((e.rb >= l_pac) << 1) | (uint64_t)r;
      v.push_back(Key(x, y));
    }
  std::sort(v.begin(), v.end());
  int last[4] = {-1, -1, -1, -1};
  const uint64_t idsh = t.flavour == BPSW_TAIL_C ? (uint64_t)(int64_t)(int32_t)((uint32_t)id << 8) : (uint64_t)id << 8;  // `int id` in the C
  for (int i = 0; i < (int)v.size(); ++i) {
    for (int r = 0; r < 2; ++r) {
      const int dir = r << 1 | (int)(v[(size_t)i].second >> 1 & 1);
      if (pes[dir].failed) continue;
      const int which = r << 1 | (int)((v[(size_t)i].second & 1) ^ 1);
      for (int k = last[which]; k >= 0; --k) {
        if ((int)(v[(size_t)k].second & 3) != which) continue;
        const long long dist = (long long)v[(size_t)i].first - (long long)v[(size_t)k].first;
        if (dist > pes[dir].high) break;
        if (dist < pes[dir].low) continue;
        const double pen = pair_penalty(pes[dir], dir, dist);  // .721 * log(2 * erfc(|ns| / sqrt 2)), ns = (dist - avg) / std
        int q = (int)((double)((v[(size_t)i].second >> 32) + (v[(size_t)k].second >> 32)) + pen * o.a + .499);
        if (q < 0) q = 0;
        const uint64_t y = (uint64_t)k << 32 | (uint64_t)i;
        u.push_back(Key((uint64_t)q << 32 | (hash64(y ^ idsh) & 0xffffffffULL), y));
      }
    }
    last[v[(size_t)i].second & 3] = i;
  }
  PairScore P;
  P.score = 0; P.sub = 0; P.n_sub = 0; P.z[0] = P.z[1] = -1;
  if (!u.empty()) {
    const int gap = std::max(o.a + o.b, std::max(o.o_del + o.e_del, o.o_ins + o.e_ins));
    std::sort(u.begin(), u.end());
    const Key& best = u.back();
    const int i = (int)(best.second >> 32), k = (int)(best.second & 0xffffffffULL);
    P.z[v[(size_t)i].second & 1] = (int)((v[(size_t)i].second & 0xffffffffULL) >> 2);
    P.z[v[(size_t)k].second & 1] = (int)((v[(size_t)k].second & 0xffffffffULL) >> 2);
    P.score = (int)(best.first >> 32);
    P.sub = u.size() > 1 ? (int)(u[u.size() - 2].first >> 32) : 0;
    for (long ii = (long)u.size() - 2; ii >= 0; --ii)
      if (P.sub - (int)(u[(size_t)ii].first >> 32) <= gap) ++P.n_sub;
  }
  return P;
}

// ---- plan / emit -------------------------------------------------------------------------------------------------------
struct Aln {  // a mem_aln_t under construction
  bpsw_aln_t a;
  const uint32_t* cigar = nullptr;  // a.n_cigar words
  const char* md = nullptr;         // a.md_len bytes
};

struct EndPlan {
  int h_job = -1;                             // the job behind h[i] (-1: the unmapped record)
  std::vector<std::pair<int, int> > se_jobs;  // memRegToSAMSe: (region index k, job)
};
struct PairPlan {
  bool paired = false;
  int z[2] = {0, 0}, q_se[2] = {0, 0}, extra_flag = 1;
  EndPlan end[2];
};

// The text of a call is written straight into the caller's buffer (round 3 built a std::string and copied it: push_back by
// push_back, 2.3 GB/s and a 3 MB memcpy per 4 096 pairs).  Past the capacity it only counts, so that *out_needed comes out right.
struct Text {
  char* buf;
  size_t cap, n = 0;
  Text(char* b, size_t c) : buf(b), cap(b ? c : 0) {}
  size_t size() const { return n; }
  char* grow(size_t len) {  // len more bytes, to be written by the caller; nullptr when they do not fit (they still count)
    char* p = n + len <= cap ? buf + n : nullptr;
    n += len;
    return p;
  }
  void push_back(char c) { if (n < cap) buf[n] = c; ++n; }
  void append(const char* p, size_t len) { char* d = grow(len); if (d) memcpy(d, p, len); }
  Text& operator+=(const char* z) { append(z, strlen(z)); return *this; }
  Text& operator+=(const std::string& z) { append(z.data(), z.size()); return *this; }
};

const char kDigitPairs[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";

void put_num(Text& s, long long v) {
  char b[24];
  int n = 24;
  const bool neg = v < 0;
  unsigned long long x = neg ? (unsigned long long)(-v) : (unsigned long long)v;
  while (x >= 100) { const unsigned r = (unsigned)(x % 100); x /= 100; b[--n] = kDigitPairs[2 * r + 1]; b[--n] = kDigitPairs[2 * r]; }
  if (x >= 10) { b[--n] = kDigitPairs[2 * x + 1]; b[--n] = kDigitPairs[2 * x]; }
  else b[--n] = (char)('0' + x);
  if (neg) b[--n] = '-';
  s.append(b + n, (size_t)(24 - n));
}

int ref_len_of(const Aln& p) {  // getRlen, R2S:146-160
  int l = 0;
  if (p.a.n_cigar > 0 && p.cigar)
    for (int k = 0; k < p.a.n_cigar; ++k) { const int op = (int)(p.cigar[(size_t)k] & 0xf); if (op == 0 || op == 2) l += (int)(p.cigar[(size_t)k] >> 4); }
  return l;
}

void put_contig(Text& s, const BnsView& bns, int rid) {
  if ((size_t)rid < bns.name.size() && !bns.name[(size_t)rid].empty()) s += bns.name[(size_t)rid];
  else { s += "ctg"; put_num(s, rid + 1); }
}

// The SEQ and QUAL columns: 2 x read length of the ~390 bytes of a line.  codes[0..n) -> "ACGTN" (forward) or the reverse
// complement "TGCAN" read backwards; quals copied or reversed.  With SSSE3 sixteen characters per pshufb, else byte by byte.
void put_bases_scalar(char* d, const uint8_t* seq, size_t n, bool rev) {
  if (!rev) for (size_t i = 0; i < n; ++i) d[i] = "ACGTN"[seq[i] > 4 ? 4 : seq[i]];
  else for (size_t i = 0; i < n; ++i) d[i] = "TGCAN"[seq[n - 1 - i] > 4 ? 4 : seq[n - 1 - i]];
}
void put_reversed_scalar(char* d, const uint8_t* q, size_t n) {
  for (size_t i = 0; i < n; ++i) d[i] = (char)q[n - 1 - i];
}
#if defined(__x86_64__)
__attribute__((target("ssse3"))) void put_bases_ssse3(char* d, const uint8_t* seq, size_t n, bool rev) {
  const __m128i four = _mm_set1_epi8(4);
  const __m128i flip = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
  const __m128i tab = rev ? _mm_setr_epi8('T', 'G', 'C', 'A', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N')
                          : _mm_setr_epi8('A', 'C', 'G', 'T', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N', 'N');
  size_t i = 0;
  for (; i + 16 <= n; i += 16) {
    __m128i v = _mm_loadu_si128((const __m128i*)(rev ? seq + (n - 16 - i) : seq + i));
    if (rev) v = _mm_shuffle_epi8(v, flip);
    v = _mm_min_epu8(v, four);  // codes above 4 print as N
    _mm_storeu_si128((__m128i*)(d + i), _mm_shuffle_epi8(tab, v));
  }
  if (!rev) put_bases_scalar(d + i, seq + i, n - i, false);
  else put_bases_scalar(d + i, seq, n - i, true);   // the n - i bases left are the FIRST ones of the read
}
__attribute__((target("ssse3"))) void put_reversed_ssse3(char* d, const uint8_t* q, size_t n) {
  const __m128i flip = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
  size_t i = 0;
  for (; i + 16 <= n; i += 16)
    _mm_storeu_si128((__m128i*)(d + i), _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(q + (n - 16 - i))), flip));
  put_reversed_scalar(d + i, q, n - i);
}
const bool kHaveSsse3 = __builtin_cpu_supports("ssse3");
#else
const bool kHaveSsse3 = false;
#endif
inline void put_bases(char* d, const uint8_t* seq, size_t n, bool rev) {
#if defined(__x86_64__)
  if (kHaveSsse3) return put_bases_ssse3(d, seq, n, rev);
#endif
  put_bases_scalar(d, seq, n, rev);
}
inline void put_reversed(char* d, const uint8_t* q, size_t n) {
#if defined(__x86_64__)
  if (kHaveSsse3) return put_reversed_ssse3(d, q, n);
#endif
  put_reversed_scalar(d, q, n);
}

// memAlnToSAM, R2S:328-560 (C: native/bwamem.c:726-838; the Scala leaves the comment field out, R2S:546-556)
void aln_to_sam(const BnsView& bns, int flavour, Text& s, const char* name, size_t name_len, int l_seq, const uint8_t* seq,
                const uint8_t* qual, const Aln* list, const size_t n_list, int which, const Aln* mate_in, const char* rg_id) {
  Aln p = list[(size_t)which];
  Aln m;
  const bool has_m = mate_in != nullptr;
  if (has_m) m = *mate_in;
  if (has_m) p.a.flag |= 0x1;
  if (p.a.rid < 0) p.a.flag |= 0x4;
  if (has_m && m.a.rid < 0) p.a.flag |= 0x8;
  if (p.a.rid < 0 && has_m && m.a.rid >= 0) { p.a.rid = m.a.rid; p.a.pos = m.a.pos; p.a.is_rev = m.a.is_rev; p.a.n_cigar = 0; }
  if (has_m && m.a.rid < 0 && p.a.rid >= 0) { m.a.rid = p.a.rid; m.a.pos = p.a.pos; m.a.is_rev = p.a.is_rev; m.a.n_cigar = 0; }
  if (p.a.is_rev) p.a.flag |= 0x10;
  if (has_m && m.a.is_rev) p.a.flag |= 0x20;
  s.append(name, name_len); s.push_back('\t');
  const int folded = (p.a.flag & 0xffff) | ((p.a.flag & 0x10000) ? 0x100 : 0);
  if (flavour == BPSW_TAIL_SCALA) p.a.flag = folded;  // R2S:362-363 assigns; native/bwamem.c:746 only prints
  put_num(s, folded); s.push_back('\t');
  if (p.a.rid >= 0) {
    put_contig(s, bns, p.a.rid); s.push_back('\t');
    put_num(s, p.a.pos + 1); s.push_back('\t');
    put_num(s, p.a.mapq); s.push_back('\t');
    if (p.a.n_cigar > 0) {
      for (int i = 0; i < p.a.n_cigar; ++i) {
        int c = (int)(p.cigar[(size_t)i] & 0xf);
        if (c == 3 || c == 4) c = which ? 4 : 3;  // hard clipping for supplementary alignments
        put_num(s, p.cigar[(size_t)i] >> 4); s.push_back("MIDSH"[c]);
      }
    } else s.push_back('*');
  } else s += "*\t0\t0\t*";
  s.push_back('\t');
  if (has_m && m.a.rid >= 0) {
    if (p.a.rid == m.a.rid) s.push_back('='); else put_contig(s, bns, m.a.rid);
    s.push_back('\t');
    put_num(s, m.a.pos + 1); s.push_back('\t');
    if (p.a.rid == m.a.rid) {
      const long long p0 = p.a.pos + (p.a.is_rev ? ref_len_of(p) - 1 : 0);
      const long long p1 = m.a.pos + (m.a.is_rev ? ref_len_of(m) - 1 : 0);
      if (m.a.n_cigar == 0 || p.a.n_cigar == 0) s.push_back('0');
      else put_num(s, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
    } else s.push_back('0');
  } else s += "*\t0\t0";
  s.push_back('\t');
  if (p.a.flag & 0x100) {
    s += "*\t*";
  } else {
    int qb = 0, qe = l_seq;
    const int nc = p.a.n_cigar;
    const bool clip_first = nc > 0 && ((p.cigar[0] & 0xf) == 4 || (p.cigar[0] & 0xf) == 3);
    const bool clip_last = nc > 0 && ((p.cigar[(size_t)nc - 1] & 0xf) == 4 || (p.cigar[(size_t)nc - 1] & 0xf) == 3);
    if (!p.a.is_rev) {
      if (which && clip_first) qb += (int)(p.cigar[0] >> 4);
      if (which && clip_last) qe -= (int)(p.cigar[(size_t)nc - 1] >> 4);
      const size_t n = (size_t)std::max(0, qe - qb);
      char* d = s.grow(n + 1 + (qual ? n : 1));  // bases, tab, qualities: written in place
      if (d) {
        put_bases(d, seq + qb, n, false);
        d[n] = '\t';
        if (qual) memcpy(d + n + 1, qual + qb, n); else d[n + 1] = '*';
      }
    } else {
      if (which && clip_first) qe -= (int)(p.cigar[0] >> 4);
      if (which && clip_last) qb += (int)(p.cigar[(size_t)nc - 1] >> 4);
      const size_t n = (size_t)std::max(0, qe - qb);
      char* d = s.grow(n + 1 + (qual ? n : 1));
      if (d) {
        put_bases(d, seq + qb, n, true);   // bases qe-1 down to qb, complemented
        d[n] = '\t';
        if (qual) put_reversed(d + n + 1, qual + qb, n); else d[n + 1] = '*';
      }
    }
  }
  if (p.a.n_cigar > 0) {
    s += "\tNM:i:"; put_num(s, p.a.NM);
    s += "\tMD:Z:"; if (p.md && p.a.md_len > 0) s.append(p.md, (size_t)p.a.md_len);
  }
  if (p.a.score >= 0) { s += "\tAS:i:"; put_num(s, p.a.score); }
  if (p.a.sub >= 0) { s += "\tXS:i:"; put_num(s, p.a.sub); }
  if (rg_id && rg_id[0]) { s += "\tRG:Z:"; s.append(rg_id, strnlen(rg_id, sizeof(((bpsw_tail_opt_t*)nullptr)->rg_id))); }  // R2S:496-500, native/bwamem.c:815
  if (!(p.a.flag & 0x100)) {
    bool others = false;
    for (size_t i = 0; i < n_list; ++i) if ((int)i != which && !(list[i].a.flag & 0x100)) { others = true; break; }
    if (others) {
      s += "\tSA:Z:";
      for (size_t i = 0; i < n_list; ++i) {
        const Aln& r = list[i];
        if ((int)i == which || (r.a.flag & 0x100)) continue;
        put_contig(s, bns, r.a.rid); s.push_back(',');
        put_num(s, r.a.pos + 1); s.push_back(',');
        s.push_back("+-"[r.a.is_rev ? 1 : 0]); s.push_back(',');
        for (int k = 0; k < r.a.n_cigar; ++k) { put_num(s, r.cigar[(size_t)k] >> 4); s.push_back("MIDSH"[r.cigar[(size_t)k] & 0xf]); }
        s.push_back(','); put_num(s, r.a.mapq);
        s.push_back(','); put_num(s, r.a.NM);
        s.push_back(';');
      }
    }
  }
  s.push_back('\n');
}

const uint32_t kNoCigar[1] = {0};
const char kNoMd[1] = {0};

// the mem_aln_t of memRegToAln: kernel result + the fields that need no sequence (R2S:188-192, :306-310)
Aln make_aln(const bpsw_opt_t& o, const bpsw_tail_opt_t& t, const bpsw_alnreg_t* ar, const JobResult* jr, const JobResults& R) {
  Aln x;
  memset(&x.a, 0, sizeof x.a);
  x.cigar = kNoCigar; x.md = kNoMd;
  if (!ar || ar->rb < 0 || ar->re < 0 || !jr) { x.a.rid = -1; x.a.pos = -1; x.a.flag |= 0x4; return x; }
  x.a.mapq = ar->secondary < 0 ? approx_mapq(o, t, *ar) : 0;
  if (ar->secondary >= 0) x.a.flag |= 0x100;
  x.a.status = jr->k.status;
  if (jr->k.status == BPSW_ALN_XREF || jr->k.status == BPSW_ALN_OVERFLOW) { x.a.rid = -1; x.a.pos = -1; return x; }
  x.a.pos = jr->k.pos; x.a.rid = jr->k.rid; x.a.is_rev = jr->k.is_rev; x.a.NM = jr->k.NM;
  x.a.n_cigar = jr->k.n_cigar; x.a.md_len = jr->k.md_len;
  x.a.score = ar->score;
  x.a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
  x.cigar = R.cig.data() + jr->cig_at; x.md = R.md.data() + jr->md_at;
  return x;
}

}  // namespace

// ---- C ABI --------------------------------------------------------------------------------------------------------------
void bpsw_tail_opt_default(bpsw_tail_opt_t* t) {  // datatype/MemOptType.scala:47-52
  if (!t) return;
  t->mask_level = 0.50f;
  t->mapq_coef_len = 50.f;
  t->mapq_coef_fac = (int)log(50.0);
  t->flavour = BPSW_TAIL_SCALA;
  memset(t->rg_id, 0, sizeof t->rg_id);
}

int bpsw_bns_load(bpsw_ctx_t* c, int32_t n_seqs, const int64_t* offset, const int32_t* len, const char* names) {
  if (!c || n_seqs < 1 || !offset || !len) return fail(BPSW_ERR_ARG, "bns_load: null or empty contig table");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  { const int prc_ = finish_pending(c); if (prc_ != BPSW_OK) return prc_; }
  DeviceRef& r = device_ref(c->device);
  RefWriteHold wr(&r.gate);
  std::lock_guard<std::mutex> gr(r.mu);
  if (r.l_pac <= 0) return fail(BPSW_ERR_ARG, "bns_load: load the reference first (bpsw_ref_load)");
  long long at = 0;
  for (int i = 0; i < n_seqs; ++i) {  // contigs tile [0, l_pac) in order, as bns_restore leaves them
    if (offset[i] != at || len[i] < 1) return fail(BPSW_ERR_ARG, "bns_load: contigs must tile [0, l_pac) in order");
    at += len[i];
  }
  if (at != r.l_pac) return fail(BPSW_ERR_ARG, "bns_load: contig lengths do not add up to l_pac");
  // (as bpsw_ref_load does: an epoch of the extension ring that byte-batch callers keep feeding takes no reference hold and would keep the
  // device-wide wait below from returning)
  struct RingPause { int d; explicit RingPause(int dev) : d(dev) { ring_pause(d); } ~RingPause() { ring_resume(d); } } ring_paused(c->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(r.ann.reserve(12 * (size_t)n_seqs + 16));
  HIP_TRY(hipMemcpy(r.ann.ptr, offset, 8 * (size_t)n_seqs, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy((char*)r.ann.ptr + 8 * (size_t)n_seqs, len, 4 * (size_t)n_seqs, hipMemcpyHostToDevice));
  r.ann_off.assign(offset, offset + n_seqs);
  r.ann_len.assign(len, len + n_seqs);
  r.ann_name.assign((size_t)n_seqs, std::string());
  if (names) {
    const char* p = names;
    for (int i = 0; i < n_seqs; ++i) { r.ann_name[(size_t)i] = p; p += r.ann_name[(size_t)i].size() + 1; }
  }
  return BPSW_OK;
}

int bpsw_reg2aln_batch(bpsw_ctx_t* c, const bpsw_opt_t* opt, const bpsw_tail_opt_t* topt, const bpsw_reg2aln_jobs_t* j, bpsw_aln_t* out,
                       uint32_t* out_cigar, uint8_t* out_md) {
  if (!c || !topt || !j || !out || !out_cigar || !out_md) return fail(BPSW_ERR_ARG, "reg2aln: null argument");
  SwScoring sc;
  int rc = make_sw_scoring(opt, &sc);
  if (rc != BPSW_OK) return rc;
  const int n = j->n;
  if (n == 0) return BPSW_OK;
  if (n < 0 || !j->read_len || !j->read_off || !j->read_pool || !j->regs) return fail(BPSW_ERR_ARG, "reg2aln: null job arrays");
  if (j->max_cigar < 1 || j->max_md < 1) return fail(BPSW_ERR_ARG, "reg2aln: max_cigar and max_md must be positive");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  { const int prc_ = finish_pending(c); if (prc_ != BPSW_OK) return prc_; }
  BnsView bns;
  rc = snapshot_bns(c, &bns);
  if (rc != BPSW_OK) return rc;
  // mapped jobs go to the device in one launch; the unmapped record needs no sequence
  std::vector<int> mapped;
  std::vector<int32_t> rl;
  std::vector<int64_t> ro;
  std::vector<bpsw_alnreg_t> rg;
  for (int t = 0; t < n; ++t)
    if (j->regs[t].rb >= 0 && j->regs[t].re >= 0) { mapped.push_back(t); rl.push_back(j->read_len[t]); ro.push_back(j->read_off[t]); rg.push_back(j->regs[t]); }
  const int m = (int)mapped.size();
  c->last_tail_ms = 0.f; c->last_tail_jobs = 0; c->last_tail_resubmitted = 0; c->tail_host_ms[0] = c->tail_host_ms[1] = c->tail_host_ms[2] = 0.;
  std::vector<Reg2AlnOut> k((size_t)m);
  std::vector<uint32_t> cig((size_t)m * (size_t)j->max_cigar);
  std::vector<uint8_t> md((size_t)m * (size_t)j->max_md);
  rc = launch_jobs(c, sc, opt, topt->flavour, bns, m, rl.data(), ro.data(), j->read_pool, j->read_pool_bytes, rg.data(), j->max_cigar,
                   j->max_md, k.data(), cig.data(), md.data());
  if (rc != BPSW_OK) return rc;
  memset(out_cigar, 0, 4 * (size_t)n * (size_t)j->max_cigar);
  memset(out_md, 0, (size_t)n * (size_t)j->max_md);
  const JobResults none;
  for (int t = 0; t < n; ++t) { Aln u = make_aln(*opt, *topt, nullptr, nullptr, none); out[t] = u.a; }
  for (int x = 0; x < m; ++x) {
    const int t = mapped[(size_t)x];
    JobResult jr;
    jr.k = k[(size_t)x];
    Aln u = make_aln(*opt, *topt, &j->regs[t], &jr, none);  // (the fields only: CIGAR and MD are copied from the launch's arrays below)
    out[t] = u.a;
    if (jr.k.status == BPSW_ALN_XREF) continue;
    if (jr.k.n_cigar <= j->max_cigar) memcpy(out_cigar + (size_t)t * (size_t)j->max_cigar, cig.data() + (size_t)x * (size_t)j->max_cigar, 4 * (size_t)std::max(jr.k.n_cigar, 0));
    memcpy(out_md + (size_t)t * (size_t)j->max_md, md.data() + (size_t)x * (size_t)j->max_md, (size_t)std::min(std::max(jr.k.md_len, 0), j->max_md));
  }
  return BPSW_OK;
}

int bpsw_sam_pe_batch(bpsw_ctx_t* c, const bpsw_opt_t* opt, const bpsw_tail_opt_t* topt, const bpsw_pairs_t* g, char* out_text,
                      size_t text_cap, int64_t* out_off, size_t* out_needed, bpsw_alnreg_t* out_regs) {
  if (!c || !topt || !g || !out_off) return fail(BPSW_ERR_ARG, "sam_pe: null argument");
  SwScoring sc;
  int rc = make_sw_scoring(opt, &sc);
  if (rc != BPSW_OK) return rc;
  const int G = g->group_size;
  if (G < 0) return fail(BPSW_ERR_ARG, "sam_pe: negative group size");
  if (G == 0) { out_off[0] = 0; if (out_needed) *out_needed = 0; return BPSW_OK; }
  if (!g->read_len || !g->read_off || !g->read_pool || !g->name_off || !g->name_pool || !g->reg_cnt) return fail(BPSW_ERR_ARG, "sam_pe: null group arrays");
  size_t n_regs = 0;
  for (int r = 0; r < 2 * G; ++r) {
    if (g->reg_cnt[r] < 0) return fail(BPSW_ERR_ARG, "sam_pe: negative region count");
    if (g->read_len[r] < 1 || g->read_off[r] < 0 || (unsigned long long)(g->read_off[r] + g->read_len[r]) > g->read_pool_bytes)
      return fail(BPSW_ERR_ARG, "sam_pe: read outside its pool (or empty)");
    n_regs += (size_t)g->reg_cnt[r];
  }
  if (n_regs && !g->regs) return fail(BPSW_ERR_ARG, "sam_pe: null region array");
  std::lock_guard<std::mutex> lock(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  { const int prc_ = finish_pending(c); if (prc_ != BPSW_OK) return prc_; }
  BnsView bns;
  rc = snapshot_bns(c, &bns);
  if (rc != BPSW_OK) return rc;
  const bpsw_opt_t& o = *opt;
  const bpsw_tail_opt_t& t = *topt;

  // ---- plan -----------------------------------------------------------------------------------------------------------
  c->last_tail_ms = 0.f; c->last_tail_jobs = 0; c->last_tail_resubmitted = 0;
  const double t_plan = now_ms();
  struct PlanScratch {
    std::vector<std::vector<bpsw_alnreg_t> > regs;
    std::vector<PairPlan> plan;
    std::vector<int32_t> job_len;
    std::vector<int64_t> job_off;
    std::vector<bpsw_alnreg_t> job_reg;
  };
  static thread_local PlanScratch scratch;  // 2 G + G vectors, re-filled in place: their heap blocks are reused from call to call
  std::vector<std::vector<bpsw_alnreg_t> >& regs = scratch.regs;
  std::vector<PairPlan>& plan = scratch.plan;
  std::vector<int32_t>& job_len = scratch.job_len;
  std::vector<int64_t>& job_off = scratch.job_off;
  std::vector<bpsw_alnreg_t>& job_reg = scratch.job_reg;
  if (regs.size() < (size_t)(2 * G)) regs.resize((size_t)(2 * G));
  if (plan.size() < (size_t)G) plan.resize((size_t)G);
  job_len.clear(); job_off.clear(); job_reg.clear();
  auto add_job = [&](int read, const bpsw_alnreg_t& ar) -> int {
    if (ar.rb < 0 || ar.re < 0) return -1;  // the unmapped record, R2S:175-180
    job_len.push_back(g->read_len[read]); job_off.push_back(g->read_off[read]); job_reg.push_back(ar);
    return (int)job_reg.size() - 1;
  };
  size_t at = 0;
  for (int k = 0; k < G; ++k) {
    PairPlan& P = plan[(size_t)k];
    P.paired = false; P.z[0] = P.z[1] = 0; P.q_se[0] = P.q_se[1] = 0; P.extra_flag = 1;
    for (int i = 0; i < 2; ++i) { P.end[i].h_job = -1; P.end[i].se_jobs.clear(); }
    std::vector<bpsw_alnreg_t>* a = &regs[(size_t)(2 * k)];
    for (int i = 0; i < 2; ++i) {
      a[i].assign(g->regs + at, g->regs + at + (size_t)g->reg_cnt[2 * k + i]);
      at += (size_t)g->reg_cnt[2 * k + i];
      mark_primary(o, t, a[i], ((g->id0 + k) << 1) | i);  // PE:1418-1419
    }
    bool pairing = !(o.flag & BPSW_MEM_F_NOPAIRING) && !a[0].empty() && !a[1].empty();
    PairScore ps;
    if (pairing) { ps = mem_pair(o, t, bns.l_pac, g->pes, a, g->id0 + k); pairing = ps.score > 0; }
    if (pairing) {  // PE:1436-1452: an end with a second good primary hit is left to the single-end path
      for (int i = 0; i < 2 && pairing; ++i)
        for (size_t j = 1; j < a[i].size(); ++j)
          if (a[i][j].secondary < 0 && a[i][j].score >= o.T) { pairing = false; break; }
    }
    if (pairing) {  // PE:1454-1510
      const int score_un = a[0][0].score + a[1][0].score - o.pen_unpaired;
      const int subo = std::max(ps.sub, score_un);
      int q_pe = raw_mapq(ps.score - subo, o.a);
      if (ps.n_sub > 0) q_pe -= (int)(4.343 * log((double)(ps.n_sub + 1)) + .499);
      q_pe = std::min(60, std::max(0, q_pe));
      P.paired = true;
      if (ps.score > score_un) {  // the paired alignment is preferred
        P.z[0] = ps.z[0]; P.z[1] = ps.z[1];
        for (int i = 0; i < 2; ++i) {
          bpsw_alnreg_t& cr = a[i][(size_t)P.z[i]];
          if (cr.secondary >= 0) { cr.sub = a[i][(size_t)cr.secondary].score; cr.secondary = t.flavour == BPSW_TAIL_C ? -2 : -1; }
          int q = approx_mapq(o, t, cr);
          q = q > q_pe ? q : (q_pe < q + 40 ? q_pe : q + 40);
          P.q_se[i] = std::min(q, raw_mapq(cr.score - cr.csub, o.a));  // cap at the tandem repeat score
        }
        P.extra_flag |= 2;
      } else {
        P.z[0] = P.z[1] = 0;
        for (int i = 0; i < 2; ++i) P.q_se[i] = approx_mapq(o, t, a[i][0]);
      }
      for (int i = 0; i < 2; ++i) P.end[i].h_job = add_job(2 * k + i, a[i][(size_t)P.z[i]]);
    } else {  // no_pairing, PE:1553-1605 + memRegToSAMSe, R2S:67-118
      for (int i = 0; i < 2; ++i) {
        EndPlan& E = P.end[i];
        if (!a[i].empty() && a[i][0].score >= o.T) E.h_job = add_job(2 * k + i, a[i][0]);
        for (size_t j = 0; j < a[i].size(); ++j) {
          const bpsw_alnreg_t& p = a[i][j];
          if (p.score < o.T) continue;
          if (p.secondary >= 0 && !(o.flag & BPSW_MEM_F_ALL)) continue;
          if (p.secondary >= 0 && p.score < a[i][(size_t)p.secondary].score * .5) continue;
          E.se_jobs.push_back(std::make_pair((int)j, j == 0 && E.h_job >= 0 ? E.h_job : add_job(2 * k + i, p)));
        }
      }
    }
  }

  // ---- device -----------------------------------------------------------------------------------------------------------
  static thread_local JobResults results;  // (the scratch of a call is kept per calling thread: no allocation in the steady state)
  JobResults& R = results;
  std::vector<JobResult>& res = R.r;
  const double t_dev = now_ms();
  rc = run_jobs(c, sc, opt, t.flavour, bns, job_len, job_off, g->read_pool, g->read_pool_bytes, job_reg, &R);
  if (rc != BPSW_OK) return rc;
  for (size_t j = 0; j < res.size(); ++j)
    if (res[j].k.status == BPSW_ALN_XREF || res[j].k.status == BPSW_ALN_OVERFLOW)
      return fail(BPSW_ERR_LIMIT, res[j].k.status == BPSW_ALN_XREF ? "sam_pe: bwaFixXref2 could not repair a region (the reference aborts here)"
                                                                   : "sam_pe: an alignment has more CIGAR operations than the kernel stages");

  // ---- emit ---------------------------------------------------------------------------------------------------------------
  const double t_emit = now_ms();
  Text text(out_text, text_cap);
  for (int k = 0; k < G; ++k) {
    const PairPlan& P = plan[(size_t)k];
    const std::vector<bpsw_alnreg_t>* a = &regs[(size_t)(2 * k)];
    const char* name = g->name_pool + g->name_off[k];
    const size_t name_len = (size_t)(g->name_off[k + 1] - g->name_off[k]);
    const uint8_t* seq[2] = {g->read_pool + g->read_off[2 * k], g->read_pool + g->read_off[2 * k + 1]};
    const uint8_t* qual[2] = {g->qual_pool ? g->qual_pool + g->read_off[2 * k] : nullptr, g->qual_pool ? g->qual_pool + g->read_off[2 * k + 1] : nullptr};
    Aln h[2];
    if (P.paired) {
      for (int i = 0; i < 2; ++i) {
        const int jb = P.end[i].h_job;
        h[i] = make_aln(o, t, &a[i][(size_t)P.z[i]], jb >= 0 ? &res[(size_t)jb] : nullptr, R);
        h[i].a.mapq = P.q_se[i];
        h[i].a.flag |= (i ? 0x80 : 0x40) | P.extra_flag;
      }
      for (int i = 0; i < 2; ++i) {
        out_off[2 * k + i] = (int64_t)text.size();
        aln_to_sam(bns, t.flavour, text, name, name_len, g->read_len[2 * k + i], seq[i], qual[i], &h[i], 1, 0, &h[1 - i], t.rg_id);
      }
      continue;
    }
    int extra_flag = 1;
    for (int i = 0; i < 2; ++i) {
      const int jb = P.end[i].h_job;
      h[i] = make_aln(o, t, jb >= 0 ? &a[i][0] : nullptr, jb >= 0 ? &res[(size_t)jb] : nullptr, R);
    }
    if (!(o.flag & BPSW_MEM_F_NOPAIRING) && h[0].a.rid == h[1].a.rid && h[0].a.rid >= 0) {  // PE:1571-1594
      long long dist;
      const int d = infer_dir(bns.l_pac, a[0][0].rb, a[1][0].rb, &dist);
      if (!g->pes[d].failed && dist >= g->pes[d].low && dist <= g->pes[d].high) extra_flag |= 2;
    }
    for (int i = 0; i < 2; ++i) {
      out_off[2 * k + i] = (int64_t)text.size();
      const int xf = (i ? 0x81 : 0x41) | extra_flag;
      std::vector<Aln> aa;
      for (size_t x = 0; x < P.end[i].se_jobs.size(); ++x) {
        const int j = P.end[i].se_jobs[x].first, jb = P.end[i].se_jobs[x].second;
        const bpsw_alnreg_t& p = a[i][(size_t)j];
        Aln q = make_aln(o, t, &p, jb >= 0 ? &res[(size_t)jb] : nullptr, R);
        q.a.flag |= xf;
        if (p.secondary >= 0) q.a.sub = -1;  // don't output the sub-optimal score
        if (j && p.secondary < 0) q.a.flag |= (o.flag & BPSW_MEM_F_NO_MULTI) ? 0x10000 : 0x800;  // supplementary
        if (j && !aa.empty() && q.a.mapq > aa[0].a.mapq) q.a.mapq = aa[0].a.mapq;
        aa.push_back(q);
      }
      if (aa.empty()) {
        Aln u = make_aln(o, t, nullptr, nullptr, R);
        u.a.flag |= xf;
        aa.push_back(u);
        aln_to_sam(bns, t.flavour, text, name, name_len, g->read_len[2 * k + i], seq[i], qual[i], aa.data(), aa.size(), 0, &h[1 - i], t.rg_id);
      } else {
        for (size_t x = 0; x < aa.size(); ++x)
          aln_to_sam(bns, t.flavour, text, name, name_len, g->read_len[2 * k + i], seq[i], qual[i], aa.data(), aa.size(), (int)x, &h[1 - i], t.rg_id);
      }
    }
  }
  out_off[2 * G] = (int64_t)text.size();
  if (out_regs) {
    size_t w = 0;
    for (int r = 0; r < 2 * G; ++r) { if (!regs[(size_t)r].empty()) memcpy(out_regs + w, regs[(size_t)r].data(), sizeof(bpsw_alnreg_t) * regs[(size_t)r].size()); w += regs[(size_t)r].size(); }
  }
  c->tail_host_ms[0] = t_dev - t_plan; c->tail_host_ms[1] = t_emit - t_dev; c->tail_host_ms[2] = now_ms() - t_emit;
  if (out_needed) *out_needed = text.size();
  if (!out_text || text.size() > text_cap) return fail(BPSW_ERR_CAPACITY, "sam_pe: text buffer too small (see *out_needed)");
  return BPSW_OK;
}

// ---- host-only exports (no device) -------------------------------------------------------------------------------------------
int bpsw_mark_primary_se(const bpsw_opt_t* opt, const bpsw_tail_opt_t* topt, int32_t n, bpsw_alnreg_t* regs, int64_t id) {
  if (!opt || !topt || n < 0 || (n > 0 && !regs)) return fail(BPSW_ERR_ARG, "mark_primary_se: null argument");
  std::vector<bpsw_alnreg_t> v(regs, regs + n);
  mark_primary(*opt, *topt, v, id);
  if (n) memcpy(regs, v.data(), sizeof(bpsw_alnreg_t) * (size_t)n);
  return BPSW_OK;
}
int bpsw_approx_mapq_se(const bpsw_opt_t* opt, const bpsw_tail_opt_t* topt, const bpsw_alnreg_t* reg) {
  if (!opt || !topt || !reg) return fail(BPSW_ERR_ARG, "approx_mapq_se: null argument");
  return approx_mapq(*opt, *topt, *reg);
}
int bpsw_mem_pair(const bpsw_opt_t* opt, const bpsw_tail_opt_t* topt, int64_t l_pac, const bpsw_pestat_t pes[4], int32_t n0,
                  const bpsw_alnreg_t* regs0, int32_t n1, const bpsw_alnreg_t* regs1, int64_t id, int32_t out5[5]) {
  if (!opt || !topt || !pes || !out5 || n0 < 0 || n1 < 0 || (n0 > 0 && !regs0) || (n1 > 0 && !regs1))
    return fail(BPSW_ERR_ARG, "mem_pair: null argument");
  std::vector<bpsw_alnreg_t> a[2];
  a[0].assign(regs0, regs0 + n0);
  a[1].assign(regs1, regs1 + n1);
  const PairScore p = mem_pair(*opt, *topt, (long long)l_pac, pes, a, id);
  out5[0] = p.score; out5[1] = p.sub; out5[2] = p.n_sub; out5[3] = p.z[0]; out5[4] = p.z[1];
  return BPSW_OK;
}
int bpsw_sort_dedup(int32_t n, bpsw_alnreg_t* regs, float mask_level_redun, int mode) {
  if (n < 0 || (n > 0 && !regs) || (mode != BPSW_RESCUE_C && mode != BPSW_RESCUE_SCALA)) return fail(BPSW_ERR_ARG, "sort_dedup: bad argument");
  std::vector<bpsw_alnreg_t> v(regs, regs + n);
  const int m = sort_dedup_regs(v, mask_level_redun, mode);
  if (m > 0) memcpy(regs, v.data(), sizeof(bpsw_alnreg_t) * (size_t)m);
  return m;
}

int bpsw_pe_stat(const bpsw_opt_t* opt, const bpsw_tail_opt_t* topt, int64_t l_pac, int32_t n_pairs, const int32_t* reg_cnt,
                 const bpsw_alnreg_t* regs, bpsw_pestat_t pes[4]) {
  if (!opt || !topt || !pes || n_pairs < 0 || (n_pairs > 0 && !reg_cnt)) return fail(BPSW_ERR_ARG, "pe_stat: null argument");
  auto cal_sub = [&](const bpsw_alnreg_t* a, int n) {  // PE:77-101
    int j = 1;
    for (; j < n; ++j) {
      const int b_max = std::max(a[j].qb, a[0].qb), e_min = std::min(a[j].qe, a[0].qe);
      if (e_min > b_max) {
        const int min_l = std::min(a[j].qe - a[j].qb, a[0].qe - a[0].qb);
        if ((float)(e_min - b_max) >= (float)min_l * topt->mask_level) break;
      }
    }
    return j < n ? a[j].score : opt->min_seed_len * opt->a;
  };
  std::vector<int64_t> isize[4];
  size_t at = 0;
  for (int i = 0; i < n_pairs; ++i) {
    const int n0 = reg_cnt[2 * i], n1 = reg_cnt[2 * i + 1];
    if (n0 < 0 || n1 < 0 || ((n0 || n1) && !regs)) return fail(BPSW_ERR_ARG, "pe_stat: bad region counts");
    const bpsw_alnreg_t* r0 = regs + at;
    const bpsw_alnreg_t* r1 = r0 + n0;
    at += (size_t)n0 + (size_t)n1;
    if (n0 == 0 || n1 == 0) continue;
    if (cal_sub(r0, n0) > 0.8 * r0[0].score || cal_sub(r1, n1) > 0.8 * r1[0].score) continue;  // MIN_RATIO
    long long is;
    const int dir = infer_dir((long long)l_pac, r0[0].rb, r1[0].rb, &is);
    if (topt->flavour == BPSW_TAIL_SCALA) is = (long long)(int32_t)is;  // `var dist: Int`, PE:148
    if (is > 0 && is <= opt->max_ins) isize[dir].push_back(is);
  }
  memset(pes, 0, 4 * sizeof(bpsw_pestat_t));
  size_t most = 0;
  for (int d = 0; d < 4; ++d) {
    bpsw_pestat_t& r = pes[d];
    std::vector<int64_t>& q = isize[d];
    most = std::max(most, q.size());
    if (q.size() < 10) { r.failed = 1; continue; }  // MIN_DIR_CNT
    std::sort(q.begin(), q.end());
    const int p25 = (int)q[(size_t)(int)(.25 * q.size() + .499)], p75 = (int)q[(size_t)(int)(.75 * q.size() + .499)];
    r.low = std::max(1, (int)(p25 - 2.0 * (p75 - p25) + .499));  // OUTLIER_BOUND
    r.high = (int)(p75 + 2.0 * (p75 - p25) + .499);
    int x = 0;
    double avg = 0, var = 0;
    for (int64_t v : q) if (v >= r.low && v <= r.high) { avg += (double)v; ++x; }
    avg /= x;
    for (int64_t v : q) if (v >= r.low && v <= r.high) var += ((double)v - avg) * ((double)v - avg);
    r.avg = avg;
    r.std = sqrt(var / x);
    r.low = (int)(p25 - 3.0 * (p75 - p25) + .499);  // MAPPING_BOUND
    r.high = (int)(p75 + 3.0 * (p75 - p25) + .499);
    if (r.low > r.avg - 4.0 * r.std) r.low = (int)(r.avg - 4.0 * r.std + .499);  // MAX_STDDEV
    if (r.high < r.avg - 4.0 * r.std)  // PE:215 assigns avg - 4 sigma, native/bwamem_pair.c:99 avg + 4 sigma
      r.high = topt->flavour == BPSW_TAIL_SCALA ? (int)(r.avg - 4.0 * r.std + .499) : (int)(r.avg + 4.0 * r.std + .499);
    if (r.low < 1) r.low = 1;
  }
  for (int d = 0; d < 4; ++d)
    if (pes[d].failed == 0 && (double)isize[d].size() < most * 0.05) pes[d].failed = 1;  // MIN_DIR_RATIO
  return BPSW_OK;
}

// ---- worker2 in one call: the rescue (boundary 1) followed by the tail -------------------------------------------------------
// memSamPeGroupJNIPrepare (PE:1895-2000) with getAlnRegRefJNI (PE:1810-1878) in coordinate form: per end the regions within
// penUnpaired of the best one (at most maxMatesw) are anchors, each with the four orientation windows of the mate; then
// bpsw_matesw_group (windows read from the resident reference), then bpsw_sam_pe_batch on the rescued lists.
int bpsw_worker2_batch(bpsw_ctx_t* c, const bpsw_opt_t* opt, const bpsw_tail_opt_t* topt, const bpsw_pairs_t* g, int rescue_mode,
                       char* out_text, size_t text_cap, int64_t* out_off, size_t* out_needed, int32_t* out_reg_cnt,
                       bpsw_alnreg_t* out_regs, int64_t out_regs_cap, int64_t* out_regs_total) {
  if (!c || !opt || !topt || !g || !out_off) return fail(BPSW_ERR_ARG, "worker2: null argument");
  const int G = g->group_size;
  if (G < 0) return fail(BPSW_ERR_ARG, "worker2: negative group size");
  if (G > 0 && (!g->read_len || !g->read_off || !g->read_pool || !g->reg_cnt)) return fail(BPSW_ERR_ARG, "worker2: null group arrays");
  const long long l_pac = (long long)bpsw_ref_length(c);
  if (l_pac <= 0) return fail(BPSW_ERR_ARG, "worker2: no reference loaded on this device (bpsw_ref_load)");
  // ---- prepare: anchors and their windows ----------------------------------------------------------------------------------
  std::vector<int32_t> ref_cnt((size_t)(2 * G), 0);
  std::vector<int64_t> ref_rb, ref_re;
  size_t at = 0, n_in = 0;
  for (int e = 0; e < 2 * G; ++e) {
    const int n = g->reg_cnt[e];
    if (n < 0) return fail(BPSW_ERR_ARG, "worker2: negative region count");
    const bpsw_alnreg_t* a = g->regs + at;
    const int mate_len = g->read_len[e ^ 1];
    int cnt = 0;
    for (int j = 0; j < n && cnt < opt->max_matesw; ++j) {
      if (!(a[j].score >= a[0].score - opt->pen_unpaired)) continue;  // PE:1944-1947
      for (int r = 0; r < 4; ++r) {                                    // PE:1834-1868
        long long rb = -1, re = -1;
        if (!g->pes[r].failed) {
          const bool is_rev = (r >> 1) != (r & 1), is_larger = !(r >> 1);
          const long long lo = g->pes[r].low, hi = g->pes[r].high;
          if (!is_rev) {
            rb = is_larger ? a[j].rb + lo : a[j].rb - hi;
            re = (is_larger ? a[j].rb + hi : a[j].rb - lo) + mate_len;
          } else {
            rb = (is_larger ? a[j].rb + lo : a[j].rb - hi) - mate_len;
            re = is_larger ? a[j].rb + hi : a[j].rb - lo;
          }
          if (rb < 0) rb = 0;
          if (re > (l_pac << 1)) re = l_pac << 1;
        }
        ref_rb.push_back(rb); ref_re.push_back(re);
      }
      ++cnt;
    }
    ref_cnt[(size_t)e] = cnt;
    at += (size_t)n; n_in += (size_t)n;
  }
  // ---- rescue ----------------------------------------------------------------------------------------------------------------
  bpsw_rescue_group_t rg;
  memset(&rg, 0, sizeof rg);
  rg.group_size = G; rg.l_pac = l_pac;
  memcpy(rg.pes, g->pes, sizeof rg.pes);
  rg.seq_len = g->read_len; rg.seq_off = g->read_off; rg.seq_pool = g->read_pool; rg.seq_pool_bytes = g->read_pool_bytes;
  rg.reg_cnt = g->reg_cnt; rg.regs = g->regs; rg.ref_cnt = ref_cnt.data();
  static const int64_t kNone = 0;
  rg.ref_rb = ref_rb.empty() ? &kNone : ref_rb.data(); rg.ref_re = ref_re.empty() ? &kNone : ref_re.data();
  rg.ref_len = nullptr; rg.ref_off = nullptr; rg.ref_pool = nullptr; rg.ref_pool_bytes = 0;  // coordinate windows
  std::vector<int32_t> cnt2((size_t)(2 * G) + 1, 0);
  std::vector<bpsw_alnreg_t> regs2(n_in + 4 * ref_rb.size() / 4 * 4 + 16);
  int64_t total = 0;
  int rc = bpsw_matesw_group(c, opt, &rg, rescue_mode, cnt2.data(), regs2.data(), (int64_t)regs2.size(), &total);
  if (rc == BPSW_ERR_CAPACITY) {
    regs2.resize((size_t)total + 16);
    rc = bpsw_matesw_group(c, opt, &rg, rescue_mode, cnt2.data(), regs2.data(), (int64_t)regs2.size(), &total);
  }
  if (rc != BPSW_OK) return rc;
  // ---- tail ------------------------------------------------------------------------------------------------------------------
  bpsw_pairs_t t = *g;
  t.reg_cnt = cnt2.data();
  t.regs = regs2.data();
  if (out_regs_total) *out_regs_total = total;
  if (out_reg_cnt) memcpy(out_reg_cnt, cnt2.data(), sizeof(int32_t) * (size_t)(2 * G));
  bpsw_alnreg_t* tail_regs = (out_regs && out_regs_cap >= total) ? out_regs : nullptr;
  rc = bpsw_sam_pe_batch(c, opt, topt, &t, out_text, text_cap, out_off, out_needed, tail_regs);
  if (rc != BPSW_OK) return rc;
  if (out_regs && out_regs_cap < total) return fail(BPSW_ERR_CAPACITY, "worker2: out_regs too small (see *out_regs_total)");
  return BPSW_OK;
}

int bpsw_last_tail_times(bpsw_ctx_t* c, float* kernel_ms, int32_t* n_jobs, double host_ms[3]) {
  if (!c) return fail(BPSW_ERR_ARG, "null context");
  std::lock_guard<std::mutex> g(c->mu);
  if (kernel_ms) *kernel_ms = c->have_tail_ev ? c->last_tail_ms : 0.f;
  if (n_jobs) *n_jobs = c->have_tail_ev ? c->last_tail_jobs - c->last_tail_resubmitted : 0;  // distinct (read, region) jobs
  if (host_ms) for (int i = 0; i < 3; ++i) host_ms[i] = c->tail_host_ms[i];
  return BPSW_OK;
}
