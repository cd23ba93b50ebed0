/*
 * ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin adapter compiled INTO oracle/_ref/libbwaref.so together with the reference's own C
 * sources (taken where they lie under /root/reference/src/main/native, never copied).  It only
 * re-shapes the nested pointer arguments of mem_group_matesw (native/bwamem.h:108) into the flat
 * SoA form the tests use; every computation is the reference's.  ksw_extend2, ksw_align2,
 * ksw_global2 and mem_sort_and_dedup are called directly through ctypes and need no adapter.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "bwamem.h" /* from the reference include path, see oracle/Makefile */
#include "kvec.h"

typedef struct {
  int64_t rb, re;
  int32_t qb, qe, score, truesc, sub, csub, sub_n, w, seedcov, secondary;
  uint64_t hash;
} flat_alnreg_t;

typedef struct {
  int32_t low, high, failed, pad_;
  double avg, std;
} flat_pestat_t;

void ref_opt_default(int32_t ints[16], float *mask_level_redun, int8_t mat[25]) {
  mem_opt_t *o = mem_opt_init();
  ints[0] = o->a; ints[1] = o->b; ints[2] = o->o_del; ints[3] = o->e_del; ints[4] = o->o_ins; ints[5] = o->e_ins;
  ints[6] = o->pen_unpaired; ints[7] = o->pen_clip5; ints[8] = o->pen_clip3; ints[9] = o->w; ints[10] = o->zdrop;
  ints[11] = o->T; ints[12] = o->flag; ints[13] = o->min_seed_len; ints[14] = o->max_ins; ints[15] = o->max_matesw;
  *mask_level_redun = o->mask_level_redun;
  memcpy(mat, o->mat, 25);
  free(o);
}

int64_t ref_group_matesw_flat(const int32_t ints[16], float mask_level_redun, const int8_t mat[25], int64_t l_pac,
                              const flat_pestat_t pes_in[4], int group_size, const int32_t *seq_len,
                              const int64_t *seq_off, const uint8_t *seq_pool, const int32_t *reg_cnt,
                              const flat_alnreg_t *regs, const int32_t *ref_cnt, const int64_t *ref_rb,
                              const int64_t *ref_re, const int64_t *ref_len, const int64_t *ref_off,
                              const uint8_t *ref_pool, int32_t *out_cnt, flat_alnreg_t *out_regs, int64_t out_cap) {
  mem_opt_t *o = mem_opt_init();
  o->a = ints[0]; o->b = ints[1]; o->o_del = ints[2]; o->e_del = ints[3]; o->o_ins = ints[4]; o->e_ins = ints[5];
  o->pen_unpaired = ints[6]; o->pen_clip5 = ints[7]; o->pen_clip3 = ints[8]; o->w = ints[9]; o->zdrop = ints[10];
  o->T = ints[11]; o->flag = ints[12]; o->min_seed_len = ints[13]; o->max_ins = ints[14]; o->max_matesw = ints[15];
  o->mask_level_redun = mask_level_redun;
  memcpy(o->mat, mat, 25);
  mem_pestat_t pes[4];
  for (int r = 0; r < 4; ++r) {
    pes[r].low = pes_in[r].low; pes[r].high = pes_in[r].high; pes[r].failed = pes_in[r].failed;
    pes[r].avg = pes_in[r].avg; pes[r].std = pes_in[r].std;
  }
  int **seq_len_pairs = (int **)malloc(sizeof(int *) * group_size);
  uint8_t ***seqs = (uint8_t ***)malloc(sizeof(uint8_t **) * group_size);
  mem_alnreg_v **vec = (mem_alnreg_v **)malloc(sizeof(mem_alnreg_v *) * group_size);
  ref_t ****refs = (ref_t ****)malloc(sizeof(ref_t ***) * group_size);
  int64_t reg_base = 0, ref_base = 0;
  for (int k = 0; k < group_size; ++k) {
    seq_len_pairs[k] = (int *)malloc(sizeof(int) * 2);
    seqs[k] = (uint8_t **)malloc(sizeof(uint8_t *) * 2);
    vec[k] = (mem_alnreg_v *)malloc(sizeof(mem_alnreg_v) * 2);
    refs[k] = (ref_t ***)malloc(sizeof(ref_t **) * 2);
    for (int i = 0; i < 2; ++i) {
      seq_len_pairs[k][i] = seq_len[2 * k + i];
      seqs[k][i] = (uint8_t *)malloc((size_t)seq_len[2 * k + i] + 1);
      memcpy(seqs[k][i], seq_pool + seq_off[2 * k + i], (size_t)seq_len[2 * k + i]);
      kv_init(vec[k][i]);
      for (int j = 0; j < reg_cnt[2 * k + i]; ++j) {
        mem_alnreg_t a;
        memcpy(&a, &regs[reg_base + j], sizeof a); /* identical layout: native/bwamem.h:49-61 */
        kv_push(mem_alnreg_t, vec[k][i], a);
      }
      reg_base += reg_cnt[2 * k + i];
      const int nr = ref_cnt[2 * k + i];
      refs[k][i] = (ref_t **)malloc(sizeof(ref_t *) * (nr > 0 ? nr : 1));
      for (int j = 0; j < nr; ++j) {
        refs[k][i][j] = (ref_t *)malloc(sizeof(ref_t) * 4);
        for (int r = 0; r < 4; ++r) {
          const int64_t x = (ref_base + j) * 4 + r;
          refs[k][i][j][r].rBeg = ref_rb[x]; refs[k][i][j][r].rEnd = ref_re[x]; refs[k][i][j][r].len = ref_len[x];
          refs[k][i][j][r].ref = 0;
          if (ref_len[x] > 0) { /* ksw_align2 reverses the target in place: give it a private copy */
            refs[k][i][j][r].ref = (uint8_t *)malloc((size_t)ref_len[x]);
            memcpy(refs[k][i][j][r].ref, ref_pool + ref_off[x], (size_t)ref_len[x]);
          }
        }
      }
      ref_base += nr;
    }
  }
  mem_group_matesw(o, l_pac, pes, group_size, seq_len_pairs, seqs, refs, &vec);
  int64_t total = 0;
  int overflow = 0;
  for (int k = 0; k < group_size; ++k) {
    for (int i = 0; i < 2; ++i) {
      out_cnt[2 * k + i] = (int32_t)vec[k][i].n;
      for (size_t j = 0; j < vec[k][i].n; ++j) {
        if (total < out_cap) memcpy(&out_regs[total], &vec[k][i].a[j], sizeof(flat_alnreg_t)); else overflow = 1;
        ++total;
      }
      for (int j = 0; j < ref_cnt[2 * k + i]; ++j) {
        for (int r = 0; r < 4; ++r) free(refs[k][i][j][r].ref);
        free(refs[k][i][j]);
      }
      free(refs[k][i]); free(vec[k][i].a); free(seqs[k][i]);
    }
    free(refs[k]); free(vec[k]); free(seqs[k]); free(seq_len_pairs[k]);
  }
  free(refs); free(vec); free(seqs); free(seq_len_pairs); free(o);
  return overflow ? -total : total;
}

/* ---- batch drivers for the CPU baseline (bench.py): loops in C so the timing is the reference kernels', not ctypes ---- */
#include "ksw.h"

/* The extension() control of MemChainToAlignBatched.scala:789-883 around the reference's ksw_extend2
 * (native/ksw.c:379-476), over SoA tasks (bytes per base).  out: 7 ints per task (qBeg,qEnd,rBeg,rEnd,score,trueScore,width). */
void ref_extend_batch(int n, const int32_t *lq, const int32_t *lr, const int32_t *rq, const int32_t *rr, const int64_t *lq_off,
                      const int64_t *lr_off, const int64_t *rq_off, const int64_t *rr_off, const int32_t *reg_score,
                      const int32_t *q_beg, const int32_t *h0, const uint8_t *pool, const int8_t mat[25], int o_del, int e_del,
                      int o_ins, int e_ins, int w, int pen_clip5, int pen_clip3, int zdrop, int32_t *out) {
  for (int t = 0; t < n; ++t) {
    int aw0 = w, aw1 = w, score = reg_score[t], qle = -1, tle = -1, gtle = -1, gscore = -1, maxoff = -1, prev;
    int32_t *o = out + 7 * (size_t)t;
    o[0] = 0; o[1] = rq[t]; o[2] = 0; o[3] = 0; o[4] = -1; o[5] = reg_score[t]; o[6] = w;
    if (lq[t] > 0) {
      for (int i = 0; i < 2; ++i) {
        prev = score; aw0 = w << i;
        score = ksw_extend2(lq[t], pool + lq_off[t], lr[t], pool + lr_off[t], 5, mat, o_del, e_del, o_ins, e_ins, aw0, pen_clip5,
                            zdrop, h0[t], &qle, &tle, &gtle, &gscore, &maxoff);
        if (score == prev || maxoff < (aw0 >> 1) + (aw0 >> 2)) break;
      }
      o[4] = score;
      if (gscore <= 0 || gscore <= score - pen_clip5) { o[0] = q_beg[t] - qle; o[2] = -tle; o[5] = score; }
      else { o[0] = 0; o[2] = -gtle; o[5] = gscore; }
    }
    if (rq[t] > 0) {
      const int sc0 = score;
      for (int i = 0; i < 2; ++i) {
        prev = score; aw1 = w << i;
        score = ksw_extend2(rq[t], pool + rq_off[t], rr[t], pool + rr_off[t], 5, mat, o_del, e_del, o_ins, e_ins, aw1, pen_clip3,
                            zdrop, sc0, &qle, &tle, &gtle, &gscore, &maxoff);
        if (score == prev || maxoff < (aw1 >> 1) + (aw1 >> 2)) break;
      }
      o[4] = score;
      if (gscore <= 0 || gscore <= score - pen_clip3) { o[1] = qle; o[3] = tle; o[5] += score - sc0; }
      else { o[1] = rq[t]; o[3] = gtle; o[5] += gscore - sc0; }
    }
    o[6] = aw0 > aw1 ? aw0 : aw1;
  }
}

/* ksw_align2 (native/ksw.c:342-364; the SSE2 u8/i16 kernels jniNative.so runs) over SoA jobs.  out: 7 ints per job. */
void ref_align2_batch(int n, const int32_t *q_len, const int32_t *t_len, const int64_t *q_off, const int64_t *t_off,
                      const uint8_t *q_rev, const uint8_t *q_pool, const uint8_t *t_pool, const int8_t mat[25], int o_del,
                      int e_del, int o_ins, int e_ins, int xtra, int32_t *out) {
  uint8_t *q = (uint8_t *)malloc(65536), *t = (uint8_t *)malloc(1 << 20);
  for (int j = 0; j < n; ++j) {
    const int ql = q_len[j], tl = t_len[j];
    for (int i = 0; i < ql; ++i) {
      const uint8_t c = q_pool[q_off[j] + (q_rev[j] ? ql - 1 - i : i)];
      q[i] = q_rev[j] ? (c < 4 ? 3 - c : 4) : c;
    }
    memcpy(t, t_pool + t_off[j], (size_t)tl); /* ksw_align2 reverses its inputs in place */
    kswr_t r = ksw_align2(ql, q, tl, t, 5, mat, o_del, e_del, o_ins, e_ins, xtra, 0);
    int32_t *o = out + 7 * (size_t)j;
    o[0] = r.score; o[1] = r.te; o[2] = r.qe; o[3] = r.score2; o[4] = r.te2; o[5] = r.tb; o[6] = r.qb;
  }
  free(q); free(t);
}

/* bns_get_seq (native/bntseq.c:355-376) into a caller buffer: returns the length, or -length if it exceeds cap */
#include "bntseq.h"
int64_t ref_bns_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, uint8_t *out, int64_t cap) {
  int64_t len = 0;
  uint8_t *seq = bns_get_seq(l_pac, pac, beg, end, &len);
  if (len > cap) { free(seq); return -len; }
  if (seq && len > 0) memcpy(out, seq, (size_t)len);
  free(seq);
  return len;
}

/* mem_chain2aln (native/bwamem.c:552-672) over the flat batch layout of orc_chain2aln_batch.  mem_seed_t / mem_chain_t are
 * private to bwamem.c (lines 167-176); the declarations below restate their layout for the call. */
typedef struct { int64_t rbeg; int32_t qbeg, len; } shim_seed_t;
typedef struct { int n, m; int64_t pos; shim_seed_t *seeds; } shim_chain_t;
void mem_chain2aln(const mem_opt_t *opt, int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *query, const void *c,
                   mem_alnreg_v *av);
int64_t ref_chain2aln_batch(const int32_t ints[16], const int8_t mat[25], int64_t l_pac, const uint8_t *pac, int n_reads,
                            const int32_t *read_len, const int64_t *read_off, const uint8_t *read_pool,
                            const int32_t *chain_cnt, const int32_t *seed_cnt, const int64_t *seed_rbeg,
                            const int32_t *seed_qbeg, const int32_t *seed_len, int32_t *out_cnt, flat_alnreg_t *out_regs,
                            int64_t out_cap) {
  mem_opt_t *o = mem_opt_init();
  o->a = ints[0]; o->b = ints[1]; o->o_del = ints[2]; o->e_del = ints[3]; o->o_ins = ints[4]; o->e_ins = ints[5];
  o->pen_unpaired = ints[6]; o->pen_clip5 = ints[7]; o->pen_clip3 = ints[8]; o->w = ints[9]; o->zdrop = ints[10];
  o->T = ints[11]; o->flag = ints[12]; o->min_seed_len = ints[13]; o->max_ins = ints[14]; o->max_matesw = ints[15];
  memcpy(o->mat, mat, 25);
  int64_t total = 0, chain_at = 0, seed_at = 0;
  int overflow = 0;
  for (int r = 0; r < n_reads; ++r) {
    mem_alnreg_v av;
    kv_init(av);
    for (int c = 0; c < chain_cnt[r]; ++c) {
      shim_chain_t ch;
      ch.n = ch.m = seed_cnt[chain_at + c];
      ch.pos = ch.n > 0 ? seed_rbeg[seed_at] : 0;
      ch.seeds = (shim_seed_t *)malloc(sizeof(shim_seed_t) * (size_t)(ch.n > 0 ? ch.n : 1));
      for (int i = 0; i < ch.n; ++i) {
        ch.seeds[i].rbeg = seed_rbeg[seed_at + i]; ch.seeds[i].qbeg = seed_qbeg[seed_at + i]; ch.seeds[i].len = seed_len[seed_at + i];
      }
      mem_chain2aln(o, l_pac, pac, read_len[r], read_pool + read_off[r], &ch, &av);
      free(ch.seeds);
      seed_at += ch.n;
    }
    chain_at += chain_cnt[r];
    out_cnt[r] = (int32_t)av.n;
    for (size_t j = 0; j < av.n; ++j) {
      if (total < out_cap) memcpy(&out_regs[total], &av.a[j], sizeof(flat_alnreg_t)); else overflow = 1;
      ++total;
    }
    free(av.a);
  }
  free(o);
  return overflow ? -total : total;
}

/* ---- worker2's tail: the reference's own mem_mark_primary_se / mem_approx_mapq_se / mem_pair / mem_reg2aln / mem_sam_pe
 * over flat arrays.  ints[16] as above; tail = {mask_level, mapQ_coef_len, (float)mapQ_coef_fac}. ---- */
#include "bwa.h"
void mem_mark_primary_se(const mem_opt_t *opt, int n, mem_alnreg_t *a, int64_t id);
int mem_approx_mapq_se(const mem_opt_t *opt, const mem_alnreg_t *a);
int mem_pair(const mem_opt_t *opt, int64_t l_pac, const uint8_t *pac, const mem_pestat_t pes[4], bseq1_t s[2], mem_alnreg_v a[2],
             int id, int *sub, int *n_sub, int z[2]);
int mem_sam_pe(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2],
               mem_alnreg_v a[2]);

static mem_opt_t *shim_opt(const int32_t ints[16], float mask_level_redun, const int8_t mat[25], const float tail[3]) {
  mem_opt_t *o = mem_opt_init();
  o->a = ints[0]; o->b = ints[1]; o->o_del = ints[2]; o->e_del = ints[3]; o->o_ins = ints[4]; o->e_ins = ints[5];
  o->pen_unpaired = ints[6]; o->pen_clip5 = ints[7]; o->pen_clip3 = ints[8]; o->w = ints[9]; o->zdrop = ints[10];
  o->T = ints[11]; o->flag = ints[12]; o->min_seed_len = ints[13]; o->max_ins = ints[14]; o->max_matesw = ints[15];
  o->mask_level_redun = mask_level_redun;
  memcpy(o->mat, mat, 25);
  if (tail) { o->mask_level = tail[0]; o->mapQ_coef_len = tail[1]; o->mapQ_coef_fac = (int)tail[2]; }
  return o;
}
static void shim_pes(const flat_pestat_t in[4], mem_pestat_t pes[4]) {
  for (int r = 0; r < 4; ++r) {
    pes[r].low = in[r].low; pes[r].high = in[r].high; pes[r].failed = in[r].failed; pes[r].avg = in[r].avg; pes[r].std = in[r].std;
  }
}
static bntseq_t *shim_bns(int64_t l_pac, int n_seqs, const int64_t *ann_off, const int32_t *ann_len, const int64_t *name_off,
                          const char *name_pool) {
  bntseq_t *b = (bntseq_t *)calloc(1, sizeof(bntseq_t));
  b->l_pac = l_pac; b->n_seqs = n_seqs;
  b->anns = (bntann1_t *)calloc((size_t)n_seqs, sizeof(bntann1_t));
  for (int i = 0; i < n_seqs; ++i) {
    b->anns[i].offset = ann_off[i]; b->anns[i].len = ann_len[i];
    const size_t l = name_off ? (size_t)(name_off[i + 1] - name_off[i]) : 0;
    b->anns[i].name = (char *)calloc(l + 1, 1);
    if (l) memcpy(b->anns[i].name, name_pool + name_off[i], l);
    b->anns[i].anno = b->anns[i].name + l; /* empty */
  }
  return b;
}
static void shim_bns_free(bntseq_t *b) {
  for (int i = 0; i < b->n_seqs; ++i) free(b->anns[i].name);
  free(b->anns); free(b);
}

void ref_mark_primary_se(const int32_t ints[16], const int8_t mat[25], const float tail[3], int n, flat_alnreg_t *regs, int64_t id) {
  mem_opt_t *o = shim_opt(ints, 0.95f, mat, tail);
  mem_mark_primary_se(o, n, (mem_alnreg_t *)regs, id); /* identical layout: native/bwamem.h:49-61 */
  free(o);
}
int ref_approx_mapq_se(const int32_t ints[16], const int8_t mat[25], const float tail[3], const flat_alnreg_t *reg) {
  mem_opt_t *o = shim_opt(ints, 0.95f, mat, tail);
  const int q = mem_approx_mapq_se(o, (const mem_alnreg_t *)reg);
  free(o);
  return q;
}
/* out4 = {ret, sub, n_sub, z0, z1} */
void ref_mem_pair(const int32_t ints[16], const int8_t mat[25], int64_t l_pac, const flat_pestat_t pes_in[4], int n0,
                  const flat_alnreg_t *a0, int n1, const flat_alnreg_t *a1, int id, int32_t out5[5]) {
  mem_opt_t *o = shim_opt(ints, 0.95f, mat, 0);
  mem_pestat_t pes[4];
  shim_pes(pes_in, pes);
  mem_alnreg_v a[2];
  a[0].n = a[0].m = (size_t)n0; a[0].a = (mem_alnreg_t *)a0;
  a[1].n = a[1].m = (size_t)n1; a[1].a = (mem_alnreg_t *)a1;
  bseq1_t s[2];
  memset(s, 0, sizeof s);
  int sub = 0, n_sub = 0, z[2] = {-1, -1};
  out5[0] = mem_pair(o, l_pac, 0, pes, s, a, id, &sub, &n_sub, z);
  out5[1] = sub; out5[2] = n_sub; out5[3] = z[0]; out5[4] = z[1];
  free(o);
}

/* mem_reg2aln (native/bwamem.c:949-1021) for n (read, region) jobs.  out: 10 x int64 per job {pos, rid, flag, is_rev, mapq, NM,
 * n_cigar, score, sub, md_len}; cigar: cigar_cap words per job; md: md_cap bytes per job (text without the NUL). */
void ref_reg2aln_batch(const int32_t ints[16], const int8_t mat[25], const float tail[3], int64_t l_pac, const uint8_t *pac,
                       int n_seqs, const int64_t *ann_off, const int32_t *ann_len, int n, const int32_t *read_len,
                       const int64_t *read_off, const uint8_t *read_pool, const flat_alnreg_t *regs, int64_t *out, uint32_t *cigar,
                       int cigar_cap, char *md, int md_cap) {
  mem_opt_t *o = shim_opt(ints, 0.95f, mat, tail);
  bntseq_t *bns = shim_bns(l_pac, n_seqs, ann_off, ann_len, 0, 0);
  for (int j = 0; j < n; ++j) {
    mem_aln_t a = mem_reg2aln(o, bns, pac, read_len[j], (const char *)(read_pool + read_off[j]), (const mem_alnreg_t *)&regs[j]);
    int64_t *r = out + 10 * (size_t)j;
    r[0] = a.pos; r[1] = a.rid; r[2] = a.flag; r[3] = a.is_rev; r[4] = a.mapq; r[5] = a.NM; r[6] = a.n_cigar; r[7] = a.score; r[8] = a.sub;
    r[9] = 0;
    if (a.cigar) {
      for (int k = 0; k < a.n_cigar && k < cigar_cap; ++k) cigar[(size_t)j * cigar_cap + k] = a.cigar[k];
      const char *m = (const char *)(a.cigar + a.n_cigar);
      const size_t l = strlen(m);
      r[9] = (int64_t)l;
      memcpy(md + (size_t)j * md_cap, m, l < (size_t)md_cap ? l : (size_t)md_cap);
      free(a.cigar);
    }
  }
  shim_bns_free(bns); free(o);
}

/* the reference keeps the read-group ID in a global (native/bwa.c:16, filled by bwa_set_rg :340-373 from the -R line; mem_aln2sam
 * prints it, native/bwamem.c:815): set it ("" = none) before ref_sam_pe_batch */
extern char bwa_rg_id[256];
void ref_set_rg_id(const char *id) {
  memset(bwa_rg_id, 0, 256);
  if (id) strncpy(bwa_rg_id, id, 255);
}

/* mem_sam_pe (native/bwamem_pair.c:361-453) for a group of pairs; the caller sets MEM_F_NO_RESCUE (0x20) in ints[12] when the
 * region lists are already rescued.  Text of read 2k+i in out_text[out_off[2k+i] .. out_off[2k+i+1]); out_cnt/out_regs (optional)
 * receive the region lists as mem_sam_pe leaves them.  returns total text bytes or -(needed). */
int64_t ref_sam_pe_batch(const int32_t ints[16], const int8_t mat[25], const float tail[3], int64_t l_pac, const uint8_t *pac,
                         int n_seqs, const int64_t *ann_off, const int32_t *ann_len, const int64_t *ann_name_off,
                         const char *ann_name_pool, const flat_pestat_t pes_in[4], int group_size, int64_t id0,
                         const int32_t *read_len, const int64_t *read_off, const uint8_t *read_pool, const char *qual_pool,
                         const int64_t *name_off, const char *name_pool, const int32_t *reg_cnt, const flat_alnreg_t *regs,
                         char *out_text, int64_t cap, int64_t *out_off) {
  mem_opt_t *o = shim_opt(ints, 0.95f, mat, tail);
  bntseq_t *bns = shim_bns(l_pac, n_seqs, ann_off, ann_len, ann_name_off, ann_name_pool);
  mem_pestat_t pes[4];
  shim_pes(pes_in, pes);
  int64_t total = 0, reg_at = 0;
  int overflow = 0;
  for (int k = 0; k < group_size; ++k) {
    bseq1_t s[2];
    mem_alnreg_v a[2];
    memset(s, 0, sizeof s);
    const size_t nl = (size_t)(name_off[k + 1] - name_off[k]);
    char *name = (char *)calloc(nl + 1, 1);
    memcpy(name, name_pool + name_off[k], nl);
    for (int i = 0; i < 2; ++i) {
      const int l = read_len[2 * k + i];
      s[i].l_seq = l; s[i].name = name;
      s[i].seq = (char *)malloc((size_t)l + 1);
      memcpy(s[i].seq, read_pool + read_off[2 * k + i], (size_t)l);
      if (qual_pool) { s[i].qual = (char *)calloc((size_t)l + 1, 1); memcpy(s[i].qual, qual_pool + read_off[2 * k + i], (size_t)l); }
      kv_init(a[i]);
      for (int j = 0; j < reg_cnt[2 * k + i]; ++j) {
        mem_alnreg_t r;
        memcpy(&r, &regs[reg_at + j], sizeof r);
        kv_push(mem_alnreg_t, a[i], r);
      }
      reg_at += reg_cnt[2 * k + i];
    }
    mem_sam_pe(o, bns, pac, pes, (uint64_t)(id0 + k), s, a);
    for (int i = 0; i < 2; ++i) {
      const size_t l = s[i].sam ? strlen(s[i].sam) : 0;
      out_off[2 * k + i] = total;
      if (total + (int64_t)l <= cap) memcpy(out_text + total, s[i].sam, l); else overflow = 1;
      total += (int64_t)l;
      free(s[i].sam); free(s[i].seq); free(s[i].qual); free(a[i].a);
    }
    free(name);
  }
  out_off[2 * group_size] = total;
  shim_bns_free(bns); free(o);
  return overflow ? -total : total;
}

/* mem_pestat (native/bwamem_pair.c:50-112) over flat region lists; it reports on stderr (unconditional fprintf) */
void mem_pestat(const mem_opt_t *opt, int64_t l_pac, int n, const mem_alnreg_v *regs, mem_pestat_t pes[4]);
void ref_pestat(const int32_t ints[16], const int8_t mat[25], const float tail[3], int64_t l_pac, int n_pairs, const int32_t *reg_cnt,
                const flat_alnreg_t *regs, flat_pestat_t out[4]) {
  mem_opt_t *o = shim_opt(ints, 0.95f, mat, tail);
  mem_alnreg_v *v = (mem_alnreg_v *)calloc((size_t)(2 * n_pairs > 0 ? 2 * n_pairs : 1), sizeof(mem_alnreg_v));
  size_t at = 0;
  for (int e = 0; e < 2 * n_pairs; ++e) {
    v[e].n = v[e].m = (size_t)reg_cnt[e];
    v[e].a = (mem_alnreg_t *)(regs + at);
    at += (size_t)reg_cnt[e];
  }
  mem_pestat_t pes[4];
  mem_pestat(o, l_pac, 2 * n_pairs, v, pes);
  for (int r = 0; r < 4; ++r) {
    out[r].low = pes[r].low; out[r].high = pes[r].high; out[r].failed = pes[r].failed; out[r].pad_ = 0;
    out[r].avg = pes[r].avg; out[r].std = pes[r].std;
  }
  free(v); free(o);
}
