/*
 * bpsw_oracle_tail.c -- TEST INFRASTRUCTURE ONLY (see bpsw_oracle.h).
 *
 * Plain-C restatement of worker2's tail, i.e. everything after the mate rescue:
 *   memMarkPrimarySe      worker2/MemMarkPrimarySe.scala:37-122          (C: native/bwamem.c:444-477)
 *   memPair               worker2/MemSamPe.scala:462-572   (PE)          (C: native/bwamem_pair.c:298-357)
 *   memApproxMapqSe       worker2/MemRegToADAMSAM.scala:568-604 (R2S)    (C: native/bwamem.c:845-872)
 *   bwaGenCigar2          R2S:738-891                                    (C: native/bwa.c:89-171)
 *   bwaFixXref2           R2S:624-719                                    (C: native/bwa.c:179-222)
 *   memRegToAln           R2S:172-313                                    (C: native/bwamem.c:949-1021)
 *   memRegToSAMSe         R2S:67-118                                     (C: native/bwamem.c:868-905)
 *   memAlnToSAM           R2S:328-560                                    (C: native/bwamem.c:726-838)
 *   memSamPeGroupRest     PE:1390-1612                                   (C: native/bwamem_pair.c:361-453 minus the rescue)
 *
 * Parity status: PINNED in the C flavour (ORC_TAIL_C) against the reference's own mem_reg2aln / mem_mark_primary_se /
 * mem_pair / mem_approx_mapq_se / mem_sam_pe compiled in place (oracle/_ref; tests/golden/mem_reg2aln.npz,
 * tests/golden/mem_sam_pe.npz).  The Scala flavour (ORC_TAIL_SCALA) differs from it in exactly the places listed in
 * DESIGN.md 4.7 (T1..T4); each difference is a one-line switch on `flavour` below and cites the Scala line.
 * log() / erfc() are the C library's: the JVM's Math.log and commons-math3 Erf.erfc cannot be run here;
 * both results go through (int)(x + .499), so only a value within one ulp of a rounding boundary could differ.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bpsw_oracle.h"

#define T_MEM_F_NOPAIRING 0x4
#define T_MEM_F_ALL 0x8
#define T_MEM_F_NO_MULTI 0x10
#define T_MAPQ_COEF 30.0 /* MEM_MAPQ_COEF, native/bwamem.c:843 */

void orc_tail_opt_default(orc_tail_opt_t *t) { /* MemOptType.scala:47-52 */
  t->mask_level = 0.50f;
  t->mapq_coef_len = 50.f;
  t->mapq_coef_fac = (int)log(50.0);
  t->pad_ = 0;
  memset(t->rg_id, 0, sizeof t->rg_id);
}

/* MemMarkPrimarySe.scala:111-122 == hash_64, native/utils.h */
static uint64_t hash_64(uint64_t key) {
  key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
  key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
  return key;
}

/* ------------------------------------------------------------------ memMarkPrimarySe */
static int g_hash_signed;
static int cmp_score_hash(const void *x_, const void *y_) { /* alnreg_hlt, native/bwamem.c:391 ; sortBy(-score, hash) :62 */
  const orc_alnreg_t *x = (const orc_alnreg_t *)x_, *y = (const orc_alnreg_t *)y_;
  if (x->score != y->score) return x->score > y->score ? -1 : 1;
  if (x->hash == y->hash) return 0;
  if (g_hash_signed) return (int64_t)x->hash < (int64_t)y->hash ? -1 : 1; /* Scala Long ordering */
  return x->hash < y->hash ? -1 : 1;
}

void orc_mark_primary_se(const orc_opt_t *o, const orc_tail_opt_t *t, int n, orc_alnreg_t *a, int64_t id, int flavour) {
  int i, k, nz = 0;
  if (n == 0) return;
  int *z = (int *)calloc((size_t)n + 1, sizeof(int)); /* Scala: new Array[Int](n), zero-filled (:46) */
  for (i = 0; i < n; ++i) { a[i].sub = 0; a[i].secondary = -1; a[i].hash = hash_64((uint64_t)(id + i)); }
  /* hash_64 is a bijection, so (score, hash) is a total order: stable vs introsort cannot matter, signedness can */
  g_hash_signed = flavour == ORC_TAIL_SCALA;
  qsort(a, (size_t)n, sizeof(orc_alnreg_t), cmp_score_hash);
  int tmp = o->a + o->b;
  if (o->o_del + o->e_del > tmp) tmp = o->o_del + o->e_del;
  if (o->o_ins + o->e_ins > tmp) tmp = o->o_ins + o->e_ins;
  z[nz++] = 0;
  for (i = 1; i < n; ++i) {
    for (k = 0; k < nz; ++k) {
      const int j = z[k];
      const int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb;
      const int e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
      if (e_min > b_max) {
        const int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
        if (e_min - b_max >= min_l * t->mask_level) {
          if (a[j].sub == 0) a[j].sub = a[i].score;
          if (a[j].score - a[i].score <= tmp) ++a[j].sub_n;
          break;
        }
      }
    }
    if (k == nz) z[nz++] = i;
    else a[i].secondary = flavour == ORC_TAIL_C ? z[k] : z[k + 1]; /* T4: the Scala loop has already stepped k (:93-101) */
  }
  free(z);
}

/* ------------------------------------------------------------------ memApproxMapqSe, R2S:568-604 */
int orc_approx_mapq_se(const orc_opt_t *o, const orc_tail_opt_t *t, const orc_alnreg_t *a, int flavour) {
  int mapq, l, sub = a->sub > 0 ? a->sub : o->min_seed_len * o->a; /* Scala `> 0`, C `!= 0`: sub is never negative here */
  if (flavour == ORC_TAIL_C) sub = a->sub ? a->sub : o->min_seed_len * o->a;
  double identity;
  if (a->csub > sub) sub = a->csub;
  if (sub >= a->score) return 0;
  l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
  identity = 1. - (double)(l * o->a - a->score) / (o->a + o->b) / l;
  if (a->score == 0) {
    mapq = 0;
  } else if (t->mapq_coef_len > 0) {
    double tmp;
    if (flavour == ORC_TAIL_C) tmp = l < t->mapq_coef_len ? 1. : t->mapq_coef_fac / log(l); /* native/bwamem.c:857 */
    else tmp = l > t->mapq_coef_len ? t->mapq_coef_fac / log(l) : 1.;                       /* T2: R2S:586, differs at l == 50 */
    tmp *= identity * identity;
    mapq = (int)(6.02 * (a->score - sub) / o->a * tmp * tmp + .499);
  } else {
    mapq = (int)(T_MAPQ_COEF * (1. - (double)sub / a->score) * log(a->seedcov) + .499);
    if (identity < 0.95) mapq = (int)(mapq * identity * identity + .499);
  }
  if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
  if (mapq > 60) mapq = 60;
  if (mapq < 0) mapq = 0;
  return mapq;
}

/* ------------------------------------------------------------------ bwaGenCigar2, R2S:738-891 */
static int put_num(char *md, int cap, int at, int v) { /* kputw */
  char b[16];
  int n = snprintf(b, sizeof b, "%d", v);
  for (int i = 0; i < n; ++i) if (at + i < cap) md[at + i] = b[i];
  return at + n;
}
static int put_ch(char *md, int cap, int at, char c) {
  if (at < cap) md[at] = c;
  return at + 1;
}

/* returns 0 when the Scala returns (0,0,0,null) (R2S:748, :755), else 1.  md receives the MD text (no NUL), *md_len its
 * length (may exceed md_cap: then the text is truncated); cigar likewise against cigar_cap. */
int orc_gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t *pac,
                   int l_query, const uint8_t *query_in, int64_t rb, int64_t re, int flavour, int *score, int *n_cigar, int *NM,
                   uint32_t *cigar, int cigar_cap, char *md, int md_cap, int *md_len) {
  *n_cigar = 0; *NM = flavour == ORC_TAIL_C ? -1 : 0; *score = 0; *md_len = 0;
  if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
  uint8_t *rseq = (uint8_t *)malloc((size_t)(re - rb) + 8);
  uint8_t *query = (uint8_t *)malloc((size_t)l_query + 8);
  memcpy(query, query_in, (size_t)l_query);
  const int64_t rlen = orc_bns_get_seq(l_pac, pac, rb, re, rseq, re - rb);
  int ok = 0, i;
  if (re - rb != rlen) goto done; /* R2S:755 */
  ok = 1;
  if (rb >= l_pac) { /* reverse both so that indels are placed leftmost, R2S:760-779 */
    for (i = 0; i < l_query >> 1; ++i) { uint8_t t = query[i]; query[i] = query[l_query - 1 - i]; query[l_query - 1 - i] = t; }
    for (i = 0; i < rlen >> 1; ++i) { uint8_t t = rseq[i]; rseq[i] = rseq[rlen - 1 - i]; rseq[rlen - 1 - i] = t; }
  }
  if (l_query == re - rb && w_ == 0) { /* no gap, no DP: R2S:781-794 */
    if (cigar_cap > 0) cigar[0] = (uint32_t)l_query << 4;
    *n_cigar = 1;
    for (i = 0; i < l_query; ++i) *score += mat[rseq[i] * 5 + query[i]];
  } else {
    int max_ins = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_ins) / e_ins + 1.);
    int max_del = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_del) / e_del + 1.);
    int max_gap = max_ins > max_del ? max_ins : max_del, w, min_w;
    if (flavour == ORC_TAIL_C) { /* native/bwa.c:120-122 */
      max_gap = max_gap > 1 ? max_gap : 1;
      w = (max_gap + abs((int)(rlen - l_query)) + 1) >> 1;
    } else { /* T1: no clamp, and abs((rlen - queryLen) + 1), R2S:796-799 */
      w = (max_gap + abs((int)(rlen - l_query) + 1)) >> 1;
    }
    w = w < w_ ? w : w_;
    min_w = abs((int)(rlen - l_query)) + 3;
    w = w > min_w ? w : min_w;
    *score = orc_sw_global(l_query, query, (int)rlen, rseq, 5, mat, o_del, e_del, o_ins, e_ins, w, n_cigar, cigar, cigar_cap);
  }
  { /* NM and MD, R2S:808-869 */
    int k, x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0, l = 0;
    const char *int2base = rb < l_pac ? "ACGTN" : "TGCAN";
    const int nc = *n_cigar <= cigar_cap ? *n_cigar : 0; /* an overflowing CIGAR is resubmitted by the caller */
    for (k = 0; k < nc; ++k) {
      const int op = cigar[k] & 0xf, len = (int)(cigar[k] >> 4);
      if (op == 0) {
        for (i = 0; i < len; ++i) {
          if (query[x + i] != rseq[y + i]) {
            l = put_num(md, md_cap, l, u);
            l = put_ch(md, md_cap, l, int2base[rseq[y + i]]);
            ++n_mm; u = 0;
          } else ++u;
        }
        x += len; y += len;
      } else if (op == 2) {
        if (k > 0 && k < nc - 1) { /* not for a leading or trailing D */
          l = put_num(md, md_cap, l, u);
          l = put_ch(md, md_cap, l, '^');
          for (i = 0; i < len; ++i) l = put_ch(md, md_cap, l, int2base[rseq[y + i]]);
          u = 0; n_gap += len;
        }
        y += len;
      } else if (op == 1) { x += len; n_gap += len; }
    }
    l = put_num(md, md_cap, l, u);
    *md_len = l;
    *NM = n_mm + n_gap;
  }
done:
  free(rseq); free(query);
  return ok;
}

/* bnsPosToRid == bns_pos2rid, native/bntseq.c:316-331 */
int orc_bns_pos2rid(int n_seqs, const int64_t *ann_off, int64_t l_pac, int64_t pos_f) {
  int left = 0, mid = 0, right = n_seqs;
  if (pos_f >= l_pac) return -1;
  while (left < right) {
    mid = (left + right) >> 1;
    if (pos_f >= ann_off[mid]) {
      if (mid == n_seqs - 1) break;
      if (pos_f < ann_off[mid + 1]) break;
      left = mid + 1;
    } else right = mid;
  }
  return mid;
}
static int64_t bns_depos(int64_t l_pac, int64_t pos, int *is_rev) { /* native/bntseq.h:83-86 */
  return (*is_rev = (pos >= l_pac)) ? (l_pac << 1) - 1 - pos : pos;
}

#define TAIL_CIG_CAP 1024
#define TAIL_MD_CAP 4096

/* bwaFixXref2, R2S:624-719.  returns iden (0, -1 unable, -2 empty) */
static int fix_xref2(const orc_opt_t *o, int n_seqs, const int64_t *ann_off, const int32_t *ann_len, int64_t l_pac,
                     const uint8_t *pac, const uint8_t *query, int *qb, int *qe, int64_t *rb, int64_t *re, int flavour) {
  int is_rev;
  if (*rb < l_pac && *re > l_pac) { *qb = *qe = -1; *rb = *re = -1; return -1; }
  const int64_t fm = bns_depos(l_pac, (*rb + *re) >> 1, &is_rev);
  const int rid = orc_bns_pos2rid(n_seqs, ann_off, l_pac, fm);
  int64_t cb = is_rev ? (l_pac << 1) - (ann_off[rid] + ann_len[rid]) : ann_off[rid];
  int64_t ce = cb + ann_len[rid];
  if (cb > *rb || ce < *re) {
    int i, score, n_cigar, NM, md_len, y;
    int64_t x;
    uint32_t *cigar = (uint32_t *)malloc(sizeof(uint32_t) * TAIL_CIG_CAP);
    char *md = (char *)malloc(TAIL_MD_CAP);
    cb = cb > *rb ? cb : *rb;
    ce = ce < *re ? ce : *re;
    orc_gen_cigar2(o->mat, o->o_del, o->e_del, o->o_ins, o->e_ins, o->w, l_pac, pac, *qe - *qb, query + *qb, *rb, *re, flavour,
                   &score, &n_cigar, &NM, cigar, TAIL_CIG_CAP, md, TAIL_MD_CAP, &md_len);
    for (i = 0, x = *rb, y = *qb; i < n_cigar; ++i) {
      const int op = cigar[i] & 0xf, len = (int)(cigar[i] >> 4);
      if (op == 0) {
        if (x <= cb && cb < x + len) { *qb = (int)(y + (cb - x)); *rb = cb; }
        if (x < ce && ce <= x + len) { *qe = (int)(y + (ce - x)); *re = ce; break; }
        else { x += len; y += len; }
      } else if (op == 1) {
        y += len;
      } else if (op == 2) {
        if (x <= cb && cb < x + len) { *qb = y; *rb = x + len; }
        if (x < ce && ce <= x + len) { *qe = y; *re = x; break; }
        else x += len;
      }
    }
    free(cigar); free(md);
  }
  return (*qb == *qe || *rb == *re) ? -2 : 0;
}

static int infer_bw(int l1, int l2, int score, int a, int q, int r) { /* R2S:127-140 == native/bwamem.c:706-713 */
  int w;
  if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
  w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
  if (w < abs(l1 - l2)) w = abs(l1 - l2);
  return w;
}

/* memRegToAln, R2S:172-313.  reg == NULL (or rb/re < 0): the unmapped record.  cigar/md as in orc_gen_cigar2 (final CIGAR,
 * after the squeeze and the clipping).  a->status: 0 ok, 1 bwaFixXref2 failed (the Scala asserts), 2 no CIGAR (null) */
void orc_reg2aln(const orc_opt_t *o, const orc_tail_opt_t *t, int n_seqs, const int64_t *ann_off, const int32_t *ann_len,
                 int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *query, const orc_alnreg_t *ar, int flavour,
                 orc_aln_t *a, uint32_t *cigar, int cigar_cap, char *md, int md_cap) {
  memset(a, 0, sizeof *a);
  if (ar == NULL || ar->rb < 0 || ar->re < 0) { a->rid = -1; a->pos = -1; a->flag |= 0x4; return; }
  int qb = ar->qb, qe = ar->qe, i, w2, tmp, score = 0, last_sc = -(1 << 30), NM = 0, n_cigar = 0, md_len = 0, is_rev;
  int64_t rb = ar->rb, re = ar->re, pos;
  a->mapq = ar->secondary < 0 ? orc_approx_mapq_se(o, t, ar, flavour) : 0;
  if (ar->secondary >= 0) a->flag |= 0x100;
  if (fix_xref2(o, n_seqs, ann_off, ann_len, l_pac, pac, query, &qb, &qe, &rb, &re, flavour) < 0) { a->status = 1; a->rid = -1; a->pos = -1; return; }
  tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, o->a, o->o_del, o->e_del);
  w2 = infer_bw(qe - qb, (int)(re - rb), ar->truesc, o->a, o->o_ins, o->e_ins);
  w2 = w2 > tmp ? w2 : tmp;
  if (w2 > o->w) w2 = w2 < ar->w ? w2 : ar->w;
  uint32_t *cg = (uint32_t *)malloc(sizeof(uint32_t) * (TAIL_CIG_CAP + 2));
  i = 0;
  int have = 0;
  do {
    have = orc_gen_cigar2(o->mat, o->o_del, o->e_del, o->o_ins, o->e_ins, w2, l_pac, pac, qe - qb, query + qb, rb, re, flavour, &score,
                          &n_cigar, &NM, cg, TAIL_CIG_CAP, md, md_cap, &md_len);
    if (score == last_sc) break;
    last_sc = score;
    w2 <<= 1;
  } while (++i < 3 && score < ar->truesc - o->a);
  if (!have) a->status = 2;
  a->NM = NM;
  a->md_len = md_len;
  pos = bns_depos(l_pac, rb < l_pac ? rb : re - 1, &is_rev);
  a->is_rev = is_rev;
  int off = 0; /* first live entry of cg */
  if (n_cigar > 0) { /* squeeze out a leading or trailing deletion, R2S:257-268 */
    if ((cg[0] & 0xf) == 2) { pos += cg[0] >> 4; --n_cigar; off = 1; }
    else if ((cg[n_cigar - 1] & 0xf) == 2) --n_cigar;
  }
  int n_out = 0;
  int clip5 = 0, clip3 = 0;
  if (qb != 0 || qe != l_query) { /* R2S:271-297 */
    clip5 = is_rev ? l_query - qe : qb;
    clip3 = is_rev ? qb : l_query - qe;
  }
  if (clip5) { if (n_out < cigar_cap) cigar[n_out] = (uint32_t)clip5 << 4 | 3; ++n_out; }
  for (i = 0; i < n_cigar; ++i) { if (n_out < cigar_cap) cigar[n_out] = cg[off + i]; ++n_out; }
  if (clip3) { if (n_out < cigar_cap) cigar[n_out] = (uint32_t)clip3 << 4 | 3; ++n_out; }
  a->n_cigar = n_out;
  a->rid = orc_bns_pos2rid(n_seqs, ann_off, l_pac, pos);
  a->pos = pos - ann_off[a->rid];
  a->score = ar->score;
  a->sub = ar->sub > ar->csub ? ar->sub : ar->csub;
  free(cg);
}

/* ------------------------------------------------------------------ memPeStat, PE:77-260 (C: native/bwamem_pair.c:36-112) */
static int cal_sub(const orc_opt_t *o, const orc_tail_opt_t *t, int n, const orc_alnreg_t *a) { /* PE:77-101 */
  int j;
  for (j = 1; j < n; ++j) {
    const int b_max = a[j].qb > a[0].qb ? a[j].qb : a[0].qb;
    const int e_min = a[j].qe < a[0].qe ? a[j].qe : a[0].qe;
    if (e_min > b_max) {
      const int min_l = a[j].qe - a[j].qb < a[0].qe - a[0].qb ? a[j].qe - a[j].qb : a[0].qe - a[0].qb;
      if (e_min - b_max >= min_l * t->mask_level) break;
    }
  }
  return j < n ? a[j].score : o->min_seed_len * o->a;
}
static int cmp_i64(const void *a, const void *b) {
  const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}
/* regs: the region lists of the 2*n_pairs reads in (pair, end, j) order (after memSortAndDedup, before the rescue) */
void orc_pe_stat(const orc_opt_t *o, const orc_tail_opt_t *t, int64_t l_pac, int n_pairs, const int32_t *reg_cnt,
                 const orc_alnreg_t *regs, int flavour, orc_pestat_t pes[4]) {
  int64_t *isize[4];
  size_t ni[4] = {0, 0, 0, 0};
  int d;
  memset(pes, 0, 4 * sizeof(orc_pestat_t));
  for (d = 0; d < 4; ++d) isize[d] = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_pairs > 0 ? n_pairs : 1));
  size_t at = 0;
  for (int i = 0; i < n_pairs; ++i) {
    const int n0 = reg_cnt[2 * i], n1 = reg_cnt[2 * i + 1];
    const orc_alnreg_t *r0 = regs + at, *r1 = regs + at + n0;
    at += (size_t)n0 + (size_t)n1;
    if (n0 == 0 || n1 == 0) continue;
    if (cal_sub(o, t, n0, r0) > 0.8 * r0[0].score) continue; /* MIN_RATIO */
    if (cal_sub(o, t, n1, r1) > 0.8 * r1[0].score) continue;
    int64_t is;
    const int dir = orc_infer_dir(l_pac, r0[0].rb, r1[0].rb, &is);
    if (flavour == ORC_TAIL_SCALA) is = (int64_t)(int32_t)is; /* `var dist: Int`, PE:148 */
    if (is > 0 && is <= o->max_ins) isize[dir][ni[dir]++] = is;
  }
  for (d = 0; d < 4; ++d) {
    orc_pestat_t *r = &pes[d];
    int64_t *q = isize[d];
    const size_t n = ni[d];
    if (n < 10) { r->failed = 1; continue; } /* MIN_DIR_CNT */
    qsort(q, n, sizeof(int64_t), cmp_i64);
    const int p25 = (int)q[(int)(.25 * n + .499)], p75 = (int)q[(int)(.75 * n + .499)];
    r->low = (int)(p25 - 2.0 * (p75 - p25) + .499); /* OUTLIER_BOUND */
    if (r->low < 1) r->low = 1;
    r->high = (int)(p75 + 2.0 * (p75 - p25) + .499);
    size_t i;
    int x = 0;
    r->avg = 0;
    for (i = 0; i < n; ++i)
      if (q[i] >= r->low && q[i] <= r->high) { r->avg += (double)q[i]; ++x; }
    r->avg /= x;
    r->std = 0;
    for (i = 0; i < n; ++i)
      if (q[i] >= r->low && q[i] <= r->high) r->std += ((double)q[i] - r->avg) * ((double)q[i] - r->avg);
    r->std = sqrt(r->std / x);
    r->low = (int)(p25 - 3.0 * (p75 - p25) + .499); /* MAPPING_BOUND */
    r->high = (int)(p75 + 3.0 * (p75 - p25) + .499);
    if (r->low > r->avg - 4.0 * r->std) r->low = (int)(r->avg - 4.0 * r->std + .499); /* MAX_STDDEV */
    if (r->high < r->avg - 4.0 * r->std) /* B6: the Scala assigns avg - 4 sigma (PE:215), the C avg + 4 sigma (native/bwamem_pair.c:99) */
      r->high = flavour == ORC_TAIL_SCALA ? (int)(r->avg - 4.0 * r->std + .499) : (int)(r->avg + 4.0 * r->std + .499);
    if (r->low < 1) r->low = 1;
  }
  size_t max = 0;
  for (d = 0; d < 4; ++d) max = max > ni[d] ? max : ni[d];
  for (d = 0; d < 4; ++d)
    if (pes[d].failed == 0 && ni[d] < max * 0.05) pes[d].failed = 1; /* MIN_DIR_RATIO */
  for (d = 0; d < 4; ++d) free(isize[d]);
}

/* ------------------------------------------------------------------ memPair, PE:462-572 */
typedef struct { uint64_t x, y; } pair64_t;
static int cmp_pair64(const void *a_, const void *b_) { /* pair64_lt, native/utils.h: (x, y) ascending; keys are unique */
  const pair64_t *a = (const pair64_t *)a_, *b = (const pair64_t *)b_;
  if (a->x != b->x) return a->x < b->x ? -1 : 1;
  if (a->y != b->y) return a->y < b->y ? -1 : 1;
  return 0;
}

int orc_mem_pair(const orc_opt_t *o, int64_t l_pac, const orc_pestat_t pes[4], int n0, const orc_alnreg_t *a0, int n1,
                 const orc_alnreg_t *a1, int64_t id, int flavour, int *sub, int *n_sub, int z[2]) {
  const int nv = n0 + n1;
  pair64_t *v = (pair64_t *)malloc(sizeof(pair64_t) * (size_t)(nv > 0 ? nv : 1));
  size_t nu = 0, mu = 16;
  pair64_t *u = (pair64_t *)malloc(sizeof(pair64_t) * mu);
  int r, i, k, y[4], ret, n = 0;
  if (flavour == ORC_TAIL_SCALA) z[0] = z[1] = -1; /* PE:470-471; the C leaves z untouched when nothing pairs */
  for (r = 0; r < 2; ++r) {
    const orc_alnreg_t *a = r ? a1 : a0;
    const int na = r ? n1 : n0;
    for (i = 0; i < na; ++i) {
      v[n].x = (uint64_t)(a[i].rb < l_pac ? a[i].rb : (l_pac << 1) - 1 - a[i].rb);
      v[n].y = (uint64_t)a[i].score << 32 | (uint64_t)(i << 2) | (uint64_t)((a[i].rb >= l_pac) << 1) | (uint64_t)r;
      ++n;
    }
  }
  qsort(v, (size_t)n, sizeof(pair64_t), cmp_pair64);
  y[0] = y[1] = y[2] = y[3] = -1;
  for (i = 0; i < n; ++i) {
    for (r = 0; r < 2; ++r) {
      const int dir = r << 1 | (int)(v[i].y >> 1 & 1);
      if (pes[dir].failed) continue;
      const int which = r << 1 | (int)((v[i].y & 1) ^ 1);
      if (y[which] < 0) continue;
      for (k = y[which]; k >= 0; --k) {
        if ((int)(v[k].y & 3) != which) continue;
        const int64_t dist = (int64_t)v[i].x - (int64_t)v[k].x;
        if (dist > pes[dir].high) break;
        if (dist < pes[dir].low) continue;
        const double ns = (dist - pes[dir].avg) / pes[dir].std;
        int q = (int)((v[i].y >> 32) + (v[k].y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * o->a + .499);
        if (q < 0) q = 0;
        if (nu == mu) { mu <<= 1; u = (pair64_t *)realloc(u, sizeof(pair64_t) * mu); }
        u[nu].y = (uint64_t)k << 32 | (uint64_t)i;
        /* hash_64(p->y ^ id<<8): `id` is an int in the C (native/bwamem_pair.c:298), a Long in the Scala (PE:462) */
        const uint64_t idsh = flavour == ORC_TAIL_C ? (uint64_t)(int64_t)(int32_t)((uint32_t)id << 8) : (uint64_t)id << 8;
        u[nu].x = (uint64_t)q << 32 | (hash_64(u[nu].y ^ idsh) & 0xffffffffU);
        ++nu;
      }
    }
    y[v[i].y & 3] = i;
  }
  if (nu) {
    int tmp = o->a + o->b;
    tmp = tmp > o->o_del + o->e_del ? tmp : o->o_del + o->e_del;
    tmp = tmp > o->o_ins + o->e_ins ? tmp : o->o_ins + o->e_ins;
    qsort(u, nu, sizeof(pair64_t), cmp_pair64);
    i = (int)(u[nu - 1].y >> 32); k = (int)(u[nu - 1].y << 32 >> 32);
    z[v[i].y & 1] = (int)(v[i].y << 32 >> 34);
    z[v[k].y & 1] = (int)(v[k].y << 32 >> 34);
    ret = (int)(u[nu - 1].x >> 32);
    *sub = nu > 1 ? (int)(u[nu - 2].x >> 32) : 0;
    *n_sub = 0;
    for (long ii = (long)nu - 2; ii >= 0; --ii)
      if (*sub - (int)(u[ii].x >> 32) <= tmp) ++*n_sub;
  } else { ret = 0; *sub = 0; *n_sub = 0; }
  free(u); free(v);
  return ret;
}

/* ------------------------------------------------------------------ SAM text */
typedef struct { char *s; size_t l, m; } sbuf_t;
static void sb_need(sbuf_t *b, size_t extra) {
  if (b->l + extra + 1 > b->m) { b->m = (b->l + extra + 1) * 2 + 256; b->s = (char *)realloc(b->s, b->m); }
}
static void sb_putc(sbuf_t *b, char c) { sb_need(b, 1); b->s[b->l++] = c; }
static void sb_putsn(sbuf_t *b, const char *s, size_t n) { sb_need(b, n); memcpy(b->s + b->l, s, n); b->l += n; }
static void sb_puts(sbuf_t *b, const char *s) { sb_putsn(b, s, strlen(s)); }
static void sb_putl(sbuf_t *b, long long v) { char t[32]; int n = snprintf(t, sizeof t, "%lld", v); sb_putsn(b, t, (size_t)n); }

typedef struct {
  orc_aln_t a;
  uint32_t cigar[TAIL_CIG_CAP];
  char md[TAIL_MD_CAP];
} full_aln_t;

typedef struct {
  int n_seqs;
  const int64_t *ann_off;
  const int32_t *ann_len;
  const int64_t *name_off; /* n_seqs + 1 offsets into name_pool */
  const char *name_pool;
  int64_t l_pac;
  const uint8_t *pac;
} bns_view_t;

static void put_ann_name(sbuf_t *b, const bns_view_t *bns, int rid) {
  sb_putsn(b, bns->name_pool + bns->name_off[rid], (size_t)(bns->name_off[rid + 1] - bns->name_off[rid]));
}
static int get_rlen(const full_aln_t *p) { /* R2S:146-160 */
  int k, l = 0;
  for (k = 0; k < p->a.n_cigar; ++k) { const int op = p->cigar[k] & 0xf; if (op == 0 || op == 2) l += (int)(p->cigar[k] >> 4); }
  return l;
}

/* memAlnToSAM, R2S:328-560 (== mem_aln2sam, native/bwamem.c:726-838, without the comment field the Scala drops :546-556) */
static void aln2sam(const bns_view_t *bns, int flavour, sbuf_t *str, const char *name, size_t name_len, int l_seq, const uint8_t *seq,
                    const char *qual, int n, const full_aln_t *list, int which, const full_aln_t *m_, const char *rg_id) {
  int i;
  full_aln_t *p = (full_aln_t *)malloc(sizeof(full_aln_t)), *m = NULL;
  *p = list[which];
  if (m_) { m = (full_aln_t *)malloc(sizeof(full_aln_t)); *m = *m_; }
  p->a.flag |= m ? 0x1 : 0;
  p->a.flag |= p->a.rid < 0 ? 0x4 : 0;
  p->a.flag |= m && m->a.rid < 0 ? 0x8 : 0;
  if (p->a.rid < 0 && m && m->a.rid >= 0) { p->a.rid = m->a.rid; p->a.pos = m->a.pos; p->a.is_rev = m->a.is_rev; p->a.n_cigar = 0; }
  if (m && m->a.rid < 0 && p->a.rid >= 0) { m->a.rid = p->a.rid; m->a.pos = p->a.pos; m->a.is_rev = p->a.is_rev; m->a.n_cigar = 0; }
  p->a.flag |= p->a.is_rev ? 0x10 : 0;
  p->a.flag |= m && m->a.is_rev ? 0x20 : 0;
  sb_putsn(str, name, name_len); sb_putc(str, '\t');
  /* T5: the Scala rewrites alnTmp.flag here (R2S:362-363), so its later `flag & 0x100` tests see the folded value; the C
   * only prints it (native/bwamem.c:746).  Differs only under MEM_F_NO_MULTI. */
  const int folded = (p->a.flag & 0xffff) | (p->a.flag & 0x10000 ? 0x100 : 0);
  if (flavour == ORC_TAIL_SCALA) p->a.flag = folded;
  sb_putl(str, folded); sb_putc(str, '\t');
  if (p->a.rid >= 0) {
    put_ann_name(str, bns, p->a.rid); sb_putc(str, '\t');
    sb_putl(str, p->a.pos + 1); sb_putc(str, '\t');
    sb_putl(str, p->a.mapq); sb_putc(str, '\t');
    if (p->a.n_cigar) {
      for (i = 0; i < p->a.n_cigar; ++i) {
        int c = p->cigar[i] & 0xf;
        if (c == 3 || c == 4) c = which ? 4 : 3;
        sb_putl(str, p->cigar[i] >> 4); sb_putc(str, "MIDSH"[c]);
      }
    } else sb_putc(str, '*');
  } else sb_putsn(str, "*\t0\t0\t*", 7);
  sb_putc(str, '\t');
  if (m && m->a.rid >= 0) {
    if (p->a.rid == m->a.rid) sb_putc(str, '='); else put_ann_name(str, bns, m->a.rid);
    sb_putc(str, '\t');
    sb_putl(str, m->a.pos + 1); sb_putc(str, '\t');
    if (p->a.rid == m->a.rid) {
      const int64_t p0 = p->a.pos + (p->a.is_rev ? get_rlen(p) - 1 : 0);
      const int64_t p1 = m->a.pos + (m->a.is_rev ? get_rlen(m) - 1 : 0);
      if (m->a.n_cigar == 0 || p->a.n_cigar == 0) sb_putc(str, '0');
      else sb_putl(str, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
    } else sb_putc(str, '0');
  } else sb_putsn(str, "*\t0\t0", 5);
  sb_putc(str, '\t');
  if (p->a.flag & 0x100) {
    sb_putsn(str, "*\t*", 3);
  } else {
    int qb = 0, qe = l_seq;
    const int nc = p->a.n_cigar;
    const int first_clip = nc && ((p->cigar[0] & 0xf) == 4 || (p->cigar[0] & 0xf) == 3);
    const int last_clip = nc && ((p->cigar[nc - 1] & 0xf) == 4 || (p->cigar[nc - 1] & 0xf) == 3);
    if (!p->a.is_rev) {
      if (which && first_clip) qb += (int)(p->cigar[0] >> 4);
      if (which && last_clip) qe -= (int)(p->cigar[nc - 1] >> 4);
      for (i = qb; i < qe; ++i) sb_putc(str, "ACGTN"[seq[i]]);
      sb_putc(str, '\t');
      if (qual) for (i = qb; i < qe; ++i) sb_putc(str, qual[i]); else sb_putc(str, '*');
    } else {
      if (which && first_clip) qe -= (int)(p->cigar[0] >> 4);
      if (which && last_clip) qb += (int)(p->cigar[nc - 1] >> 4);
      for (i = qe - 1; i >= qb; --i) sb_putc(str, "TGCAN"[seq[i]]);
      sb_putc(str, '\t');
      if (qual) for (i = qe - 1; i >= qb; --i) sb_putc(str, qual[i]); else sb_putc(str, '*');
    }
  }
  if (p->a.n_cigar) {
    sb_puts(str, "\tNM:i:"); sb_putl(str, p->a.NM);
    sb_puts(str, "\tMD:Z:"); sb_putsn(str, p->md, (size_t)p->a.md_len);
  }
  if (p->a.score >= 0) { sb_puts(str, "\tAS:i:"); sb_putl(str, p->a.score); }
  if (p->a.sub >= 0) { sb_puts(str, "\tXS:i:"); sb_putl(str, p->a.sub); }
  if (rg_id && rg_id[0]) { sb_puts(str, "\tRG:Z:"); sb_puts(str, rg_id); } /* R2S:496-500 (samHeader.bwaReadGroupID), native/bwamem.c:815 */
  if (!(p->a.flag & 0x100)) {
    for (i = 0; i < n; ++i) if (i != which && !(list[i].a.flag & 0x100)) break;
    if (i < n) {
      sb_puts(str, "\tSA:Z:");
      for (i = 0; i < n; ++i) {
        const full_aln_t *r = &list[i];
        int k;
        if (i == which || (r->a.flag & 0x100)) continue;
        put_ann_name(str, bns, r->a.rid); sb_putc(str, ',');
        sb_putl(str, r->a.pos + 1); sb_putc(str, ',');
        sb_putc(str, "+-"[r->a.is_rev]); sb_putc(str, ',');
        for (k = 0; k < r->a.n_cigar; ++k) { sb_putl(str, r->cigar[k] >> 4); sb_putc(str, "MIDSH"[r->cigar[k] & 0xf]); }
        sb_putc(str, ','); sb_putl(str, r->a.mapq);
        sb_putc(str, ','); sb_putl(str, r->a.NM);
        sb_putc(str, ';');
      }
    }
  }
  sb_putc(str, '\n');
  free(p); free(m);
}

typedef struct {
  const orc_opt_t *o;
  const orc_tail_opt_t *t;
  const bns_view_t *bns;
  int flavour;
  int64_t n_reg2aln; /* diagnostics */
} tail_env_t;

static void reg2aln_full(tail_env_t *E, int l_seq, const uint8_t *seq, const orc_alnreg_t *ar, full_aln_t *out) {
  orc_reg2aln(E->o, E->t, E->bns->n_seqs, E->bns->ann_off, E->bns->ann_len, E->bns->l_pac, E->bns->pac, l_seq, seq, ar, E->flavour,
              &out->a, out->cigar, TAIL_CIG_CAP, out->md, TAIL_MD_CAP);
  if (ar) ++E->n_reg2aln;
}

/* memRegToSAMSe, R2S:67-118 */
static void reg2sam_se(tail_env_t *E, sbuf_t *str, const char *name, size_t name_len, int l_seq, const uint8_t *seq, const char *qual,
                       int n, const orc_alnreg_t *a, int extra_flag, const full_aln_t *m) {
  full_aln_t *aa = (full_aln_t *)malloc(sizeof(full_aln_t) * (size_t)(n > 0 ? n : 1));
  int k, na = 0;
  for (k = 0; k < n; ++k) {
    const orc_alnreg_t *p = &a[k];
    if (p->score < E->o->T) continue;
    if (p->secondary >= 0 && !(E->o->flag & T_MEM_F_ALL)) continue;
    if (p->secondary >= 0 && p->score < a[p->secondary].score * .5) continue;
    full_aln_t *q = &aa[na++];
    reg2aln_full(E, l_seq, seq, p, q);
    q->a.flag |= extra_flag;
    if (p->secondary >= 0) q->a.sub = -1;
    if (k && p->secondary < 0) q->a.flag |= (E->o->flag & T_MEM_F_NO_MULTI) ? 0x10000 : 0x800;
    if (k && q->a.mapq > aa[0].a.mapq) q->a.mapq = aa[0].a.mapq;
  }
  if (na == 0) {
    full_aln_t *t = (full_aln_t *)malloc(sizeof(full_aln_t));
    reg2aln_full(E, l_seq, seq, NULL, t);
    t->a.flag |= extra_flag;
    aln2sam(E->bns, E->flavour, str, name, name_len, l_seq, seq, qual, 1, t, 0, m, E->t->rg_id);
    free(t);
  } else {
    for (k = 0; k < na; ++k) aln2sam(E->bns, E->flavour, str, name, name_len, l_seq, seq, qual, na, aa, k, m, E->t->rg_id);
  }
  free(aa);
}

#define RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))

/* one pair of memSamPeGroupRest, PE:1399-1608 (== mem_sam_pe after the rescue, native/bwamem_pair.c:385-452).
 * a[i] (n[i] regions) are modified in place the way the reference modifies them (sort, sub, secondary, hash). */
static void sam_pe_one(tail_env_t *E, const orc_pestat_t pes[4], int64_t id, const int l_seq[2], const uint8_t *const seq[2],
                       const char *const qual[2], const char *name, size_t name_len, int n[2], orc_alnreg_t *a[2], sbuf_t out[2]) {
  const orc_opt_t *o = E->o;
  int i, j, z[2] = {0, 0}, subo = 0, n_sub = 0, extra_flag = 1, ret;
  full_aln_t *h = (full_aln_t *)malloc(sizeof(full_aln_t) * 2);
  orc_mark_primary_se(o, E->t, n[0], a[0], id << 1 | 0, E->flavour);
  orc_mark_primary_se(o, E->t, n[1], a[1], id << 1 | 1, E->flavour);
  if (o->flag & T_MEM_F_NOPAIRING) goto no_pairing;
  if (n[0] && n[1] && (ret = orc_mem_pair(o, E->bns->l_pac, pes, n[0], a[0], n[1], a[1], id, E->flavour, &subo, &n_sub, z)) > 0) {
    int is_multi[2], q_pe, score_un, q_se[2];
    for (i = 0; i < 2; ++i) {
      for (j = 1; j < n[i]; ++j)
        if (a[i][j].secondary < 0 && a[i][j].score >= o->T) break;
      is_multi[i] = j < n[i] ? 1 : 0;
    }
    if (is_multi[0] || is_multi[1]) goto no_pairing;
    score_un = a[0][0].score + a[1][0].score - o->pen_unpaired;
    subo = subo > score_un ? subo : score_un;
    q_pe = RAW_MAPQ(ret - subo, o->a);
    if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
    if (q_pe < 0) q_pe = 0;
    if (q_pe > 60) q_pe = 60;
    if (ret > score_un) {
      orc_alnreg_t *c[2] = {&a[0][z[0]], &a[1][z[1]]};
      for (i = 0; i < 2; ++i) {
        if (c[i]->secondary >= 0) { c[i]->sub = a[i][c[i]->secondary].score; c[i]->secondary = E->flavour == ORC_TAIL_C ? -2 : -1; }
        q_se[i] = orc_approx_mapq_se(o, E->t, c[i], E->flavour);
      }
      q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
      q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
      extra_flag |= 2;
      q_se[0] = q_se[0] < RAW_MAPQ(c[0]->score - c[0]->csub, o->a) ? q_se[0] : RAW_MAPQ(c[0]->score - c[0]->csub, o->a);
      q_se[1] = q_se[1] < RAW_MAPQ(c[1]->score - c[1]->csub, o->a) ? q_se[1] : RAW_MAPQ(c[1]->score - c[1]->csub, o->a);
    } else {
      z[0] = z[1] = 0;
      q_se[0] = orc_approx_mapq_se(o, E->t, &a[0][0], E->flavour);
      q_se[1] = orc_approx_mapq_se(o, E->t, &a[1][0], E->flavour);
    }
    reg2aln_full(E, l_seq[0], seq[0], &a[0][z[0]], &h[0]); h[0].a.mapq = q_se[0]; h[0].a.flag |= 0x40 | extra_flag;
    reg2aln_full(E, l_seq[1], seq[1], &a[1][z[1]], &h[1]); h[1].a.mapq = q_se[1]; h[1].a.flag |= 0x80 | extra_flag;
    aln2sam(E->bns, E->flavour, &out[0], name, name_len, l_seq[0], seq[0], qual[0], 1, &h[0], 0, &h[1], E->t->rg_id);
    aln2sam(E->bns, E->flavour, &out[1], name, name_len, l_seq[1], seq[1], qual[1], 1, &h[1], 0, &h[0], E->t->rg_id);
    free(h);
    return;
  }
no_pairing:
  for (i = 0; i < 2; ++i) {
    if (n[i] && a[i][0].score >= o->T) reg2aln_full(E, l_seq[i], seq[i], &a[i][0], &h[i]);
    else reg2aln_full(E, l_seq[i], seq[i], NULL, &h[i]);
  }
  if (!(o->flag & T_MEM_F_NOPAIRING) && h[0].a.rid == h[1].a.rid && h[0].a.rid >= 0) {
    int64_t dist;
    const int d = orc_infer_dir(E->bns->l_pac, a[0][0].rb, a[1][0].rb, &dist);
    if (!pes[d].failed && dist >= pes[d].low && dist <= pes[d].high) extra_flag |= 2;
  }
  reg2sam_se(E, &out[0], name, name_len, l_seq[0], seq[0], qual[0], n[0], a[0], 0x41 | extra_flag, &h[1]);
  reg2sam_se(E, &out[1], name, name_len, l_seq[1], seq[1], qual[1], n[1], a[1], 0x81 | extra_flag, &h[0]);
  free(h);
}

/*
 * The tail over a group of pairs, flat SoA form.
 *   read_len/read_off[2G] in read_pool (codes 0..4); qual_pool: same offsets, or NULL; name_off[G+1] into name_pool
 *   reg_cnt[2G], regs[]: regions after the rescue in (k, i, j) order; modified in place like the reference does
 *   out_text/out_off[2G+1]: the SAM text of read 2k+i is out_text[out_off[2k+i] .. out_off[2k+i+1])
 * returns the total text bytes, or -(needed) when cap is too small (nothing usable is written then).
 */
int64_t orc_sam_pe_batch(const orc_opt_t *o, const orc_tail_opt_t *t, int n_seqs, const int64_t *ann_off, const int32_t *ann_len,
                         const int64_t *ann_name_off, const char *ann_name_pool, int64_t l_pac, const uint8_t *pac,
                         const orc_pestat_t pes[4], int group_size, int64_t id0, const int32_t *read_len, const int64_t *read_off,
                         const uint8_t *read_pool, const char *qual_pool, const int64_t *name_off, const char *name_pool,
                         const int32_t *reg_cnt, orc_alnreg_t *regs, int flavour, char *out_text, int64_t cap, int64_t *out_off,
                         int64_t *n_reg2aln) {
  bns_view_t bns = {n_seqs, ann_off, ann_len, ann_name_off, ann_name_pool, l_pac, pac};
  tail_env_t E = {o, t, &bns, flavour, 0};
  int64_t total = 0, reg_at = 0;
  int overflow = 0;
  for (int k = 0; k < group_size; ++k) {
    int l_seq[2], n[2];
    const uint8_t *seq[2];
    const char *qual[2];
    orc_alnreg_t *a[2];
    sbuf_t out[2] = {{NULL, 0, 0}, {NULL, 0, 0}};
    for (int i = 0; i < 2; ++i) {
      l_seq[i] = read_len[2 * k + i];
      seq[i] = read_pool + read_off[2 * k + i];
      qual[i] = qual_pool ? qual_pool + read_off[2 * k + i] : NULL;
      n[i] = reg_cnt[2 * k + i];
      a[i] = regs + reg_at;
      reg_at += n[i];
    }
    sam_pe_one(&E, pes, id0 + k, l_seq, seq, qual, name_pool + name_off[k], (size_t)(name_off[k + 1] - name_off[k]), n, a, out);
    for (int i = 0; i < 2; ++i) {
      out_off[2 * k + i] = total;
      if (total + (int64_t)out[i].l <= cap) memcpy(out_text + total, out[i].s, out[i].l); else overflow = 1;
      total += (int64_t)out[i].l;
      free(out[i].s);
    }
  }
  out_off[2 * group_size] = total;
  if (n_reg2aln) *n_reg2aln = E.n_reg2aln;
  return overflow ? -total : total;
}
