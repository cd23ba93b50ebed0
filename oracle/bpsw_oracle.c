/*
 * bpsw_oracle.c -- TEST INFRASTRUCTURE ONLY (see bpsw_oracle.h for the rules).
 *
 * Scalar restatement of the CS-BWAMEM Smith-Waterman path, following the Scala
 * text statement by statement; each function cites the lines it follows.
 * Parity status: pinned against oracle/_ref (the reference C compiled in place)
 * by tests/test_oracle_vs_ref.py and the fixtures in tests/golden/.
 */
#include "bpsw_oracle.h"

#include <stdlib.h>
#include <string.h>

#define ORC_MINUS_INF (-0x40000000) /* SW:28 */

/* diagnostic: DP rows swept by orc_sw_extend since the last reset (used to size per-row costs in DESIGN.md) */
static int64_t g_ext_rows = 0, g_ext_calls = 0;
int64_t orc_diag_ext_rows(int reset) { int64_t r = g_ext_rows; if (reset) g_ext_rows = 0; return r; }
int64_t orc_diag_ext_calls(int reset) { int64_t r = g_ext_calls; if (reset) g_ext_calls = 0; return r; }
/* diagnostics for kernel design: histogram of the band width end - beg + 1 of every SWExtend row (bucket = width / 8, the
 * last bucket takes everything from 256 columns up), and the same weighted by nothing else -- read with orc_diag_ext_widths */
static int64_t g_ext_width_hist[33];
void orc_diag_ext_widths(int64_t out[33], int reset) {
  for (int k = 0; k < 33; ++k) { out[k] = g_ext_width_hist[k]; if (reset) g_ext_width_hist[k] = 0; }
}

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iabs(int a) { return a < 0 ? -a : a; }

/* Scala's Double.toInt / C's (int) cast: truncate toward zero, saturating. */
static int dtoi(double x) {
  if (x >= 2147483647.0) return 2147483647;
  if (x <= -2147483648.0) return (int)(-2147483647 - 1);
  return (int)x;
}

/* ------------------------------------------------------------------ SWExtend */
/* SW:61-230 */
void orc_sw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                   int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                   int zdrop_mode, int32_t out[6], int64_t *cells) {
  int32_t *eh_h = (int32_t *)calloc((size_t)qlen + 2, sizeof(int32_t)); /* SW:66,75-78 */
  int32_t *eh_e = (int32_t *)calloc((size_t)qlen + 2, sizeof(int32_t));
  int8_t *qp = (int8_t *)malloc((size_t)(qlen > 0 ? qlen : 1) * m);
  const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins; /* SW:68-69 */
  int i, j, k;
  int64_t ncell = 0;

  for (k = 0, i = 0; k < m; ++k) /* query profile, SW:81-94 */
    for (j = 0; j < qlen; ++j) qp[i++] = mat[k * m + query[j]];

  eh_h[0] = h0; /* first row, SW:97-104 */
  if (qlen >= 1) eh_h[1] = h0 > oe_ins ? h0 - oe_ins : 0;
  for (j = 2; j <= qlen && eh_h[j - 1] > e_ins; ++j) eh_h[j] = eh_h[j - 1] - e_ins;

  int max = mat[0]; /* mat.max, SW:109 */
  for (k = 1; k < m * m; ++k) max = imax(max, mat[k]);
  int max_ins = dtoi((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.0); /* SW:110-115 */
  if (max_ins < 1) max_ins = 1;
  if (w > max_ins) w = max_ins;
  int max_del = dtoi((double)(qlen * max + end_bonus - o_del) / e_del + 1.0);
  if (max_del < 1) max_del = 1;
  if (w > max_del) w = max_del;

  max = h0; /* SW:118-125 */
  int max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
  int beg = 0, end = qlen;
  int stop = 0;

  __atomic_fetch_add(&g_ext_calls, 1, __ATOMIC_RELAXED); /* (diagnostic counters: relaxed atomics, the oracle is called from many threads in tests/host_san) */
  for (i = 0; i < tlen && !stop; ++i) { /* SW:129-220 */
    __atomic_fetch_add(&g_ext_rows, 1, __ATOMIC_RELAXED);
    int t, f = 0, h1, mm = 0, mj = -1;
    const int8_t *q = qp + (size_t)target[i] * qlen;
    h1 = h0 - (o_del + e_del * (i + 1)); /* SW:137-138 */
    if (h1 < 0) h1 = 0;
    if (beg < i - w) beg = i - w; /* SW:140-142 */
    if (end > i + w + 1) end = i + w + 1;
    if (end > qlen) end = qlen;
    for (j = beg; j < end; ++j) { /* SW:145-172 */
      int h = eh_h[j], e = eh_e[j];
      eh_h[j] = h1;
      h += q[j];
      if (h < e) h = e;
      if (h < f) h = f;
      h1 = h;
      if (mm <= h) { mj = j; mm = h; } /* last arg-max, SW:158-161 */
      t = h - oe_del;
      if (t < 0) t = 0;
      e -= e_del;
      if (e < t) e = t;
      eh_e[j] = e;
      t = h - oe_ins;
      if (t < 0) t = 0;
      f -= e_ins;
      if (f < t) f = t;
      ++ncell;
    }
    eh_h[end] = h1; /* SW:174-175 */
    eh_e[end] = 0;
    { const int wd = end - beg + 1; __atomic_fetch_add(&g_ext_width_hist[wd < 0 ? 0 : (wd >= 256 ? 32 : wd >> 3)], 1, __ATOMIC_RELAXED); }
    if (j == qlen) { /* SW:177-182; j == max(beg,end) after the loop */
      if (gscore <= h1) { max_ie = i; gscore = h1; }
    }
    if (mm == 0) { /* SW:184-185 */
      stop = 1;
    } else {
      if (mm > max) { /* SW:187-193 */
        max = mm; max_i = i; max_j = mj;
        if (max_off < iabs(mj - i)) max_off = iabs(mj - i);
      } else if (zdrop > 0) { /* SW:194-199 vs native/ksw.c:455-461 */
        const int A = (i - max_i) > (mj - max_j);
        const int B = max - mm - ((i - max_i) - (mj - max_j)) * e_del > zdrop;
        const int C = max - mm - ((mj - max_j) - (i - max_i)) * e_ins > zdrop;
        if (zdrop_mode == ORC_ZDROP_SCALA) {
          if (A) { if (B) stop = 1; else if (C) stop = 1; }
        } else {
          if (A) { if (B) stop = 1; } else { if (C) stop = 1; }
        }
      }
      if (!stop) { /* SW:202-214 */
        for (j = mj; j >= beg && eh_h[j] > 0; --j) {}
        beg = j + 1;
        for (j = mj + 2; j <= end && eh_h[j] > 0; ++j) {}
        end = j;
      }
    }
  }
  out[0] = max; out[1] = max_j + 1; out[2] = max_i + 1; /* SW:222-227 */
  out[3] = max_ie + 1; out[4] = gscore; out[5] = max_off;
  if (cells) *cells += ncell;
  free(eh_h); free(eh_e); free(qp);
}

/* ------------------------------------------------------------ extension task */
/* C2AB:789-883 */
static void orc_extension_sides(const orc_ext_param_t *p, int zdrop_mode, orc_ext_ret_t *ret, int64_t *cells, int64_t *side_cells);
void orc_extension(const orc_ext_param_t *p, int zdrop_mode, orc_ext_ret_t *ret, int64_t *cells) {
  orc_extension_sides(p, zdrop_mode, ret, cells, NULL);
}
/* side_cells (optional): DP cells of the left / right side of this task, every band try included (bench.py's GCUPS split) */
static void orc_extension_sides(const orc_ext_param_t *p, int zdrop_mode, orc_ext_ret_t *ret, int64_t *cells, int64_t *side_cells) {
  const int MAX_BAND_TRY = 2; /* C2AB:50 */
  int64_t local_cells = 0;
  if (!cells) cells = &local_cells;
  const int64_t cells_at_start = *cells;
  int64_t cells_after_left = *cells;
  int aw0 = p->w, aw1 = p->w;
  int qle = -1, tle = -1, gtle = -1, gscore = -1, maxoff = -1;
  int i, brk, prev = -1, reg_score = p->reg_score;
  int32_t r[6];

  ret->q_beg = 0; ret->r_beg = 0; ret->q_end = p->right_qlen; ret->r_end = 0; /* C2AB:802-807 */
  ret->true_score = p->reg_score;
  ret->score = -1; ret->width = -1; ret->idx = -1; /* ExtRet defaults, ExtensionParameters.scala:79-87 */

  if (p->left_qlen > 0) { /* C2AB:809-842 */
    for (i = 0, brk = 0; i < MAX_BAND_TRY && !brk; ++i) {
      prev = reg_score;
      aw0 = p->w << i;
      orc_sw_extend(p->left_qlen, p->left_qs, p->left_rlen, p->left_rs, 5, p->mat, p->o_del, p->e_del,
                    p->o_ins, p->e_ins, aw0, p->pen_clip5, p->zdrop, p->h0, zdrop_mode, r, cells);
      reg_score = r[0]; qle = r[1]; tle = r[2]; gtle = r[3]; gscore = r[4]; maxoff = r[5];
      if (reg_score == prev || maxoff < (aw0 >> 1) + (aw0 >> 2)) brk = 1;
    }
    ret->score = reg_score;
    if (gscore <= 0 || gscore <= reg_score - p->pen_clip5) {
      ret->q_beg = p->q_beg - qle; ret->r_beg = -tle; ret->true_score = reg_score;
    } else {
      ret->q_beg = 0; ret->r_beg = -gtle; ret->true_score = gscore;
    }
  }
  cells_after_left = *cells;
  if (p->right_qlen > 0) { /* C2AB:844-876 */
    const int sc0 = reg_score;
    for (i = 0, brk = 0; i < MAX_BAND_TRY && !brk; ++i) {
      prev = reg_score;
      aw1 = p->w << i;
      orc_sw_extend(p->right_qlen, p->right_qs, p->right_rlen, p->right_rs, 5, p->mat, p->o_del, p->e_del,
                    p->o_ins, p->e_ins, aw1, p->pen_clip3, p->zdrop, sc0, zdrop_mode, r, cells);
      reg_score = r[0]; qle = r[1]; tle = r[2]; gtle = r[3]; gscore = r[4]; maxoff = r[5];
      if (reg_score == prev || maxoff < (aw1 >> 1) + (aw1 >> 2)) brk = 1;
    }
    ret->score = reg_score;
    if (gscore <= 0 || gscore <= reg_score - p->pen_clip3) {
      ret->q_end = qle; ret->r_end = tle; ret->true_score += reg_score - sc0;
    } else {
      ret->q_end = p->right_qlen; ret->r_end = gtle; ret->true_score += gscore - sc0;
    }
  }
  ret->width = aw0 > aw1 ? aw0 : aw1; /* C2AB:877-879 */
  ret->idx = p->idx;
  if (side_cells) { side_cells[0] = cells_after_left - cells_at_start; side_cells[1] = *cells - cells_after_left; }
}

/* ------------------------------------------------------- boundary-2 wire format */
static void put16(uint8_t *b, size_t at, int v) { b[at] = (uint8_t)(v & 0xff); b[at + 1] = (uint8_t)((v >> 8) & 0xff); }
static void put32(uint8_t *b, size_t at, int v) {
  b[at] = (uint8_t)(v & 0xff); b[at + 1] = (uint8_t)((v >> 8) & 0xff);
  b[at + 2] = (uint8_t)((v >> 16) & 0xff); b[at + 3] = (uint8_t)((v >> 24) & 0xff);
}
static int get16(const uint8_t *b, size_t at) { return (int16_t)(b[at] | (b[at + 1] << 8)); }
static int get32(const uint8_t *b, size_t at) {
  return (int32_t)((uint32_t)b[at] | ((uint32_t)b[at + 1] << 8) | ((uint32_t)b[at + 2] << 16) | ((uint32_t)b[at + 3] << 24));
}
static int task_words(const orc_ext_param_t *t) { /* C2AB:101 */
  return (((t->left_qlen + t->left_rlen + t->right_qlen + t->right_rlen) + 1) / 2 + 3) / 4;
}

size_t orc_wire_size(int n, const orc_ext_param_t *tasks) {
  size_t words = (size_t)(32 + 32 * n) >> 2;
  for (int i = 0; i < n; ++i) words += (size_t)task_words(&tasks[i]);
  return words << 2;
}

/* C2AB:76-172 */
size_t orc_wire_pack(int n, const orc_ext_param_t *tasks, uint8_t *buf, size_t cap) {
  const size_t total = orc_wire_size(n, tasks);
  if (total > cap || n < 1) return 0;
  memset(buf, 0, total);
  const size_t buf1_len = 32 + 32 * (size_t)n;
  buf[0] = (uint8_t)(int8_t)tasks[0].o_del; buf[1] = (uint8_t)(int8_t)tasks[0].e_del; /* C2AB:78-85 */
  buf[2] = (uint8_t)(int8_t)tasks[0].o_ins; buf[3] = (uint8_t)(int8_t)tasks[0].e_ins;
  buf[4] = (uint8_t)(int8_t)tasks[0].pen_clip5; buf[5] = (uint8_t)(int8_t)tasks[0].pen_clip3;
  buf[6] = (uint8_t)(int8_t)tasks[0].w;
  put32(buf, 8, n);
  int task_pos = (int)(buf1_len >> 2);
  size_t at = 32;
  for (int i = 0; i < n; ++i) { /* C2AB:95-117 */
    const orc_ext_param_t *t = &tasks[i];
    int mx = t->mat[0];
    for (int k = 1; k < 25; ++k) mx = imax(mx, t->mat[k]);
    put16(buf, at, t->left_qlen); put16(buf, at + 2, t->left_rlen);
    put16(buf, at + 4, t->right_qlen); put16(buf, at + 6, t->right_rlen);
    put32(buf, at + 8, task_pos);
    task_pos += task_words(t);
    put16(buf, at + 12, t->reg_score); put16(buf, at + 14, t->q_beg);
    put16(buf, at + 16, t->h0); put16(buf, at + 18, t->idx);
    put16(buf, at + 20, dtoi((double)(t->left_qlen * mx + t->pen_clip5 - t->o_ins) / t->e_ins + 1));
    put16(buf, at + 22, dtoi((double)(t->left_qlen * mx + t->pen_clip5 - t->o_del) / t->e_del + 1));
    put16(buf, at + 24, dtoi((double)(t->right_qlen * mx + t->pen_clip3 - t->o_ins) / t->e_ins + 1));
    put16(buf, at + 26, dtoi((double)(t->right_qlen * mx + t->pen_clip3 - t->o_del) / t->e_del + 1));
    put32(buf, at + 28, t->idx);
    at += 32;
  }
  /* nibble stream, C2AB:119-170: leftQs, rightQs, leftRs, rightRs; 8 nibbles per int32,
   * first base in the most significant nibble; each task zero-padded to a word. */
  size_t o = buf1_len;
  uint32_t acc = 0;
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    const orc_ext_param_t *t = &tasks[i];
    const uint8_t *seg[4] = {t->left_qs, t->right_qs, t->left_rs, t->right_rs};
    const int len[4] = {t->left_qlen, t->right_qlen, t->left_rlen, t->right_rlen};
    for (int s = 0; s < 4; ++s)
      for (int j = 0; j < len[s]; ++j) {
        acc = (acc << 4) | (uint32_t)(seg[s][j] & 0x0F);
        if (++cnt % 8 == 0) { put32(buf, o, (int)acc); o += 4; }
      }
    if (cnt % 8 != 0) {
      while (cnt % 8 != 0) { acc <<= 4; ++cnt; }
      put32(buf, o, (int)acc); o += 4;
    }
  }
  return total;
}

static int orc_wire_extend_impl(const uint8_t *wire, size_t bytes, const int8_t mat[25], int zdrop, int zdrop_mode,
                                int16_t *out, int64_t *cells, int64_t *side_cells);
int orc_wire_extend(const uint8_t *wire, size_t bytes, const int8_t mat[25], int zdrop, int zdrop_mode,
                    int16_t *out, int64_t *cells) {
  return orc_wire_extend_impl(wire, bytes, mat, zdrop, zdrop_mode, out, cells, NULL);
}
/* the same, also reporting the DP cells of every side: side_cells[2 t] left, [2 t + 1] right */
int orc_wire_extend_sides(const uint8_t *wire, size_t bytes, const int8_t mat[25], int zdrop, int zdrop_mode,
                          int16_t *out, int64_t *cells, int64_t *side_cells) {
  return orc_wire_extend_impl(wire, bytes, mat, zdrop, zdrop_mode, out, cells, side_cells);
}
static int orc_wire_extend_impl(const uint8_t *wire, size_t bytes, const int8_t mat[25], int zdrop, int zdrop_mode,
                                int16_t *out, int64_t *cells, int64_t *side_cells) {
  if (bytes < 32) return -1;
  const int n = get32(wire, 8);
  if (n < 0 || 32 + 32 * (size_t)n > bytes) return -1;
  uint8_t *tmp = (uint8_t *)malloc(4 * 65536);
  for (int i = 0; i < n; ++i) {
    const size_t at = 32 + 32 * (size_t)i;
    orc_ext_param_t p;
    memset(&p, 0, sizeof p);
    p.o_del = (int8_t)wire[0]; p.e_del = (int8_t)wire[1]; p.o_ins = (int8_t)wire[2]; p.e_ins = (int8_t)wire[3];
    p.pen_clip5 = (int8_t)wire[4]; p.pen_clip3 = (int8_t)wire[5]; p.w = (int8_t)wire[6];
    p.left_qlen = get16(wire, at); p.left_rlen = get16(wire, at + 2);
    p.right_qlen = get16(wire, at + 4); p.right_rlen = get16(wire, at + 6);
    const size_t pos = (size_t)get32(wire, at + 8) * 4;
    p.reg_score = get16(wire, at + 12); p.q_beg = get16(wire, at + 14);
    p.h0 = get16(wire, at + 16); p.idx = get32(wire, at + 28);
    p.zdrop = zdrop; p.mat = mat;
    const int total = p.left_qlen + p.right_qlen + p.left_rlen + p.right_rlen;
    if (total > 4 * 65536 || pos + 4 * (size_t)((total + 7) / 8) > bytes) { free(tmp); return -1; }
    for (int j = 0; j < total; ++j) { /* undo the nibble packing */
      const uint32_t wd = (uint32_t)get32(wire, pos + 4 * (size_t)(j >> 3));
      tmp[j] = (uint8_t)((wd >> (28 - 4 * (j & 7))) & 0xF);
    }
    p.left_qs = tmp; p.right_qs = tmp + p.left_qlen;
    p.left_rs = p.right_qs + p.right_qlen; p.right_rs = p.left_rs + p.left_rlen;
    orc_ext_ret_t r;
    orc_extension_sides(&p, zdrop_mode, &r, cells, side_cells ? side_cells + 2 * (size_t)i : NULL);
    int16_t *o = out + 10 * (size_t)i; /* C2AB:181-188 */
    o[0] = (int16_t)(r.idx & 0xffff); o[1] = (int16_t)((r.idx >> 16) & 0xffff);
    o[2] = (int16_t)r.q_beg; o[3] = (int16_t)r.q_end; o[4] = (int16_t)r.r_beg; o[5] = (int16_t)r.r_end;
    o[6] = (int16_t)r.score; o[7] = (int16_t)r.true_score; o[8] = (int16_t)r.width; o[9] = 0;
  }
  free(tmp);
  return n;
}

/* ------------------------------------------------------------------- SWAlign */
/* SW:417-570 */
void orc_sw_align(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                  int a, int b, int o_del, int e_del, int o_ins, int e_ins, int xtra, int32_t out[7],
                  int64_t *cells) {
  const int max_score = 255 - iabs(b), q_max = a; /* SW:423-424 */
  const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
  int32_t *eh_h = (int32_t *)calloc((size_t)qlen + 1, sizeof(int32_t)); /* SW:428,440-444 */
  int32_t *eh_e = (int32_t *)calloc((size_t)qlen + 1, sizeof(int32_t));
  int8_t *qp = (int8_t *)malloc((size_t)(qlen > 0 ? qlen : 1) * m);
  int32_t *best = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tlen > 0 ? tlen : 1)); /* SW:430-432 */
  int32_t *tend = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tlen > 0 ? tlen : 1));
  int nb = 0;
  int min_score = 0x10000, end_score = 0x10000; /* SW:434-437 */
  if (xtra & ORC_KSW_XSUBO) min_score = xtra & 0xffff;
  if (xtra & ORC_KSW_XSTOP) end_score = xtra & 0xffff;
  int i, j, k;
  int64_t ncell = 0;
  for (k = 0, i = 0; k < m; ++k) /* SW:447-461 */
    for (j = 0; j < qlen; ++j) qp[i++] = mat[k * m + query[j]];

  int max = ORC_MINUS_INF, max_i = -1, max_j = -1; /* SW:463-465 */
  int stop = 0;
  for (i = 0; i < tlen && !stop; ++i) { /* SW:469-542 */
    int t, f = 0, h1 = 0, mm = 0, mj = -1;
    const int8_t *q = qp + (size_t)target[i] * qlen;
    for (j = 0; j < qlen; ++j) { /* SW:478-511 */
      int h = eh_h[j], e = eh_e[j];
      eh_h[j] = h1;
      h += q[j];
      if (h < e) h = e;
      if (h < f) h = f;
      h1 = h;
      if (mm < h) { mj = j; mm = h; } /* first arg-max, SW:493-496 */
      t = h - oe_del;
      if (t < 0) t = 0;
      e -= e_del;
      if (e < t) e = t;
      eh_e[j] = e;
      t = h - oe_ins;
      if (t < 0) t = 0;
      f -= e_ins;
      if (f < t) f = t;
    }
    ncell += qlen;
    if (mm >= min_score) { /* SW:517-529 */
      if (nb == 0 || tend[nb - 1] + 1 != i) { best[nb] = mm; tend[nb] = i; ++nb; }
      else if (best[nb - 1] < mm) { best[nb - 1] = mm; tend[nb - 1] = i; }
    }
    if (mm > max) { /* SW:532-538 */
      max = mm; max_i = i; max_j = mj;
      if (max >= end_score || max >= max_score) stop = 1;
    }
  }
  if (max >= max_score) max = 255; /* SW:544 */
  /* SWAlnType defaults, datatype/SWAlnType.scala */
  out[0] = max; out[1] = max_i; out[2] = -1; out[3] = -1; out[4] = -1; out[5] = -1; out[6] = -1;
  if (out[0] != 255) { /* SW:549-567 */
    out[2] = max_j;
    if (nb > 0) {
      const int tmp = (out[0] + q_max - 1) / q_max;
      const int low = out[1] - tmp, high = out[1] + tmp;
      for (i = 0; i < nb; ++i)
        if ((tend[i] < low || tend[i] > high) && best[i] > out[3]) { out[3] = best[i]; out[4] = tend[i]; }
    }
  }
  if (cells) *cells += ncell;
  free(eh_h); free(eh_e); free(qp); free(best); free(tend);
}

static void rev_seq(int len, uint8_t *s) { /* SW:572-581 */
  for (int i = 0; i < (len >> 1); ++i) { uint8_t t = s[i]; s[i] = s[len - 1 - i]; s[len - 1 - i] = t; }
}

/* SW:583-601 */
void orc_sw_align2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                   int a, int b, int o_del, int e_del, int o_ins, int e_ins, int xtra, int32_t out[7],
                   int64_t *cells) {
  orc_sw_align(qlen, query, tlen, target, m, mat, a, b, o_del, e_del, o_ins, e_ins, xtra, out, cells);
  if ((xtra & ORC_KSW_XSTART) == 0 || ((xtra & ORC_KSW_XSUBO) && out[0] < (xtra & 0xffff))) return;
  /* the Scala text would index with qEnd = -1 here when score == 255; SWAlign2 is never reached that
   * way on reads <= 250 bp.  Leave tBeg/qBeg unset. */
  if (out[2] < 0 || out[1] < 0) return;
  uint8_t *q = (uint8_t *)malloc((size_t)qlen + 1), *t = (uint8_t *)malloc((size_t)tlen + 1);
  memcpy(q, query, (size_t)qlen); memcpy(t, target, (size_t)tlen);
  rev_seq(out[2] + 1, q); rev_seq(out[1] + 1, t);
  int32_t rv[7];
  orc_sw_align(out[2] + 1, q, tlen, t, m, mat, a, b, o_del, e_del, o_ins, e_ins, ORC_KSW_XSTOP | out[0], rv, cells);
  if (out[0] == rv[0]) { out[5] = out[1] - rv[1]; out[6] = out[2] - rv[2]; }
  free(q); free(t);
}

/* ------------------------------------------------------------------ SWGlobal */
static int push_cigar(int n, uint32_t *cigar, int cap, int op, int len) { /* SW:401-414 */
  if (n == 0 || (int)(cigar[n - 1] & 0xf) != op) {
    if (n < cap) cigar[n] = ((uint32_t)len << 4) | (uint32_t)op;
    return n + 1;
  }
  if (n <= cap) cigar[n - 1] += (uint32_t)len << 4;
  return n;
}

/* SW:233-397 */
int orc_sw_global(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                  int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar, uint32_t *cigar,
                  int cigar_cap) {
  const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
  const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1; /* SW:248-249 */
  int32_t *eh_h = (int32_t *)calloc((size_t)qlen + 2, sizeof(int32_t));
  int32_t *eh_e = (int32_t *)calloc((size_t)qlen + 2, sizeof(int32_t));
  int8_t *qp = (int8_t *)malloc((size_t)(qlen > 0 ? qlen : 1) * m);
  uint8_t *z = (uint8_t *)calloc((size_t)(n_col > 0 ? n_col : 1) * (size_t)(tlen > 0 ? tlen : 1), 1);
  int i, j, k;
  for (k = 0, i = 0; k < m; ++k)
    for (j = 0; j < qlen; ++j) qp[i++] = mat[k * m + query[j]];
  eh_h[0] = 0; eh_e[0] = ORC_MINUS_INF; /* SW:274-288 */
  for (j = 1; j <= qlen && j <= w; ++j) { eh_h[j] = -(o_ins + e_ins * j); eh_e[j] = ORC_MINUS_INF; }
  for (; j <= qlen; ++j) { eh_h[j] = ORC_MINUS_INF; eh_e[j] = ORC_MINUS_INF; }
  for (i = 0; i < tlen; ++i) { /* SW:292-349 */
    int f = ORC_MINUS_INF, beg = 0, end = qlen, h1 = ORC_MINUS_INF;
    const int8_t *q = qp + (size_t)target[i] * qlen;
    uint8_t *zi = z + (size_t)i * n_col;
    if (i > w) beg = i - w;
    if (i + w + 1 < qlen) end = i + w + 1;
    if (beg == 0) h1 = -(o_del + e_del * (i + 1));
    for (j = beg; j < end; ++j) {
      int mm = eh_h[j], e = eh_e[j], d, h, t;
      eh_h[j] = h1;
      mm += q[j];
      d = mm >= e ? 0 : 1;
      h = mm >= e ? mm : e;
      if (h < f) d = 2;
      if (h < f) h = f;
      h1 = h;
      t = mm - oe_del;
      e -= e_del;
      if (e > t) d |= 1 << 2;
      if (e < t) e = t;
      eh_e[j] = e;
      t = mm - oe_ins;
      f -= e_ins;
      if (f > t) d |= 2 << 4;
      if (f < t) f = t;
      zi[j - beg] = (uint8_t)d;
    }
    eh_h[end] = h1; eh_e[end] = ORC_MINUS_INF;
  }
  const int score = eh_h[qlen]; /* SW:351 */
  int n = 0, which = 0; /* backtrack, SW:355-382 */
  i = tlen - 1;
  k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
  while (i >= 0 && k >= 0) {
    if (i > w) which = (z[(size_t)i * n_col + (k - (i - w))] >> (which << 1)) & 3;
    else which = (z[(size_t)i * n_col + k] >> (which << 1)) & 3;
    if (which == 0) { n = push_cigar(n, cigar, cigar_cap, 0, 1); --i; --k; }
    else if (which == 1) { n = push_cigar(n, cigar, cigar_cap, 2, 1); --i; }
    else { n = push_cigar(n, cigar, cigar_cap, 1, 1); --k; }
  }
  if (i >= 0) n = push_cigar(n, cigar, cigar_cap, 2, i + 1);
  if (k >= 0) n = push_cigar(n, cigar, cigar_cap, 1, k + 1);
  if (n <= cigar_cap)
    for (i = 0; i < (n >> 1); ++i) { uint32_t t = cigar[i]; cigar[i] = cigar[n - 1 - i]; cigar[n - 1 - i] = t; } /* SW:384-394 */
  *n_cigar = n;
  free(eh_h); free(eh_e); free(qp); free(z);
  return score;
}

/* ------------------------------------------------------------ sort and dedup */
typedef int (*reg_lt_fn)(const orc_alnreg_t *, const orc_alnreg_t *);
static int lt_re(const orc_alnreg_t *x, const orc_alnreg_t *y) { return x->re < y->re; } /* native/bwamem.c:385 */
static int lt_score(const orc_alnreg_t *x, const orc_alnreg_t *y) { /* native/bwamem.c:388 */
  return x->score > y->score || (x->score == y->score && (x->rb < y->rb || (x->rb == y->rb && x->qb < y->qb)));
}
static int lt_re_rb(const orc_alnreg_t *x, const orc_alnreg_t *y) { /* DEDUP:40 */
  return x->re < y->re || (x->re == y->re && x->rb < y->rb);
}

static void insertion_sort(orc_alnreg_t *s, orc_alnreg_t *t, reg_lt_fn lt) { /* [s,t) ; stable */
  for (orc_alnreg_t *i = s + 1; i < t; ++i)
    for (orc_alnreg_t *j = i; j > s && lt(j, j - 1); --j) { orc_alnreg_t x = *j; *j = *(j - 1); *(j - 1) = x; }
}

static void comb_sort(size_t n, orc_alnreg_t *a, reg_lt_fn lt) { /* native/ksort.h:154-175 */
  const double shrink = 1.2473309501039786540366528676643;
  int swapped;
  size_t gap = n;
  do {
    if (gap > 2) {
      gap = (size_t)(gap / shrink);
      if (gap == 9 || gap == 10) gap = 11;
    }
    swapped = 0;
    for (orc_alnreg_t *i = a; i < a + n - gap; ++i) {
      orc_alnreg_t *j = i + gap;
      if (lt(j, i)) { orc_alnreg_t x = *i; *i = *j; *j = x; swapped = 1; }
    }
  } while (swapped || gap > 2);
  if (gap != 1) insertion_sort(a, a + n, lt);
}

/* Behavioural twin of klib's ks_introsort (native/ksort.h:176-227): the C library's tie order
 * depends on its exact pivot/partition sequence, so the oracle reproduces that sequence. */
static void klib_introsort(size_t n, orc_alnreg_t *a, reg_lt_fn lt) {
  typedef struct { orc_alnreg_t *l, *r; int d; } frame_t;
  if (n < 1) return;
  if (n == 2) {
    if (lt(&a[1], &a[0])) { orc_alnreg_t x = a[0]; a[0] = a[1]; a[1] = x; }
    return;
  }
  int d;
  for (d = 2; (1ul << d) < n; ++d) {}
  frame_t *stack = (frame_t *)malloc(sizeof(frame_t) * (sizeof(size_t) * (size_t)d + 2));
  frame_t *top = stack;
  orc_alnreg_t *s = a, *t = a + (n - 1);
  d <<= 1;
  for (;;) {
    if (s < t) {
      if (--d == 0) { comb_sort((size_t)(t - s) + 1, s, lt); t = s; continue; }
      orc_alnreg_t *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
      if (lt(k, i)) { if (lt(k, j)) k = j; }
      else k = lt(j, i) ? i : j;
      orc_alnreg_t rp = *k;
      if (k != t) { orc_alnreg_t x = *k; *k = *t; *t = x; }
      for (;;) {
        do ++i; while (lt(i, &rp));
        do --j; while (i <= j && lt(&rp, j));
        if (j <= i) break;
        orc_alnreg_t x = *i; *i = *j; *j = x;
      }
      { orc_alnreg_t x = *i; *i = *t; *t = x; }
      if (i - s > t - i) {
        if (i - s > 16) { top->l = s; top->r = i - 1; top->d = d; ++top; }
        s = t - i > 16 ? i + 1 : t;
      } else {
        if (t - i > 16) { top->l = i + 1; top->r = t; top->d = d; ++top; }
        t = i - s > 16 ? i - 1 : s;
      }
    } else {
      if (top == stack) { free(stack); insertion_sort(a, a + n, lt); return; }
      --top; s = top->l; t = top->r; d = top->d;
    }
  }
}

static void stable_sort(int n, orc_alnreg_t *a, reg_lt_fn lt) { insertion_sort(a, a + n, lt); }

/* native/bwamem.c:394-435 (mode C)  /  DEDUP:33-141 (mode Scala) */
int orc_sort_dedup(int n, orc_alnreg_t *a, float mask_level_redun, int mode) {
  int m, i, j;
  if (n <= 1) return n;
  if (mode == ORC_RESCUE_C) klib_introsort((size_t)n, a, lt_re);
  else stable_sort(n, a, lt_re_rb);
  for (i = 1; i < n; ++i) {
    orc_alnreg_t *p = &a[i];
    if (p->rb >= a[i - 1].re) continue;
    for (j = i - 1; j >= 0 && p->rb < a[j].re; --j) {
      orc_alnreg_t *q = &a[j];
      int64_t orr, oq, mr, mq;
      if (q->qe == q->qb) continue; /* a[j] has been excluded */
      orr = q->re - p->rb;
      oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
      mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
      mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
      /* Scala: Long > Float*Long (float arithmetic); C: int64 > float*int64 (float arithmetic) */
      if ((float)orr > mask_level_redun * (float)mr && (float)oq > mask_level_redun * (float)mq) {
        if (p->score < q->score) { p->qe = p->qb; break; }
        else q->qe = q->qb;
      }
    }
  }
  for (i = 0, m = 0; i < n; ++i)
    if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
  n = m;
  if (mode == ORC_RESCUE_C) klib_introsort((size_t)n, a, lt_score);
  else stable_sort(n, a, lt_score);
  for (i = 1; i < n; ++i)
    if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
  if (mode == ORC_RESCUE_C) { /* native/bwamem.c:430-434: a[0] is kept unconditionally */
    for (i = 1, m = 1; i < n; ++i)
      if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
    return n < 1 ? n : m;
  }
  for (i = 0, m = 0; i < n; ++i) /* DEDUP:123 filter */
    if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
  return m;
}

/* ------------------------------------------------------------------ defaults */
void orc_opt_default(orc_opt_t *o) { /* datatype/MemOptType.scala:28-73 */
  memset(o, 0, sizeof *o);
  o->a = 1; o->b = 4; o->o_del = 6; o->e_del = 1; o->o_ins = 6; o->e_ins = 1;
  o->pen_unpaired = 17; o->pen_clip5 = 5; o->pen_clip3 = 5; o->w = 100; o->zdrop = 100;
  o->T = 30; o->flag = 0; o->min_seed_len = 19; o->max_ins = 10000; o->max_matesw = 100;
  o->mask_level_redun = 0.95f;
  int k = 0;
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 4; ++j) o->mat[k++] = (int8_t)(i == j ? o->a : -o->b);
    o->mat[k++] = -1;
  }
  for (int j = 0; j < 5; ++j) o->mat[k++] = -1;
}

int orc_infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist) { /* native/bwamem_pair.c:27-34 */
  const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
  const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
  *dist = p2 > b1 ? p2 - b1 : b1 - p2;
  return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

/* ---------------------------------------------------------------- rescue core */
typedef struct { orc_alnreg_t *a; int n, cap; } regvec_t;
static void rv_push(regvec_t *v, const orc_alnreg_t *r) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 8; v->a = (orc_alnreg_t *)realloc(v->a, sizeof(orc_alnreg_t) * (size_t)v->cap); }
  v->a[v->n++] = *r;
}

/* native/bwamem_pair.c:159-228 (mode C);  PE:1111-1238 (mode Scala) */
static int matesw_precompute(const orc_opt_t *opt, int64_t l_pac, const orc_pestat_t pes[4],
                             const orc_alnreg_t *a, int l_ms, const uint8_t *ms, regvec_t *ma,
                             const int64_t *rrb, const int64_t *rre, const int64_t *rlen,
                             const int64_t *roff, const uint8_t *ref_pool, int mode, int64_t *n_sw,
                             int64_t *cells) {
  int skip[4], r, i, n = 0;
  for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
  for (i = 0; i < ma->n; ++i) {
    int64_t dist;
    r = orc_infer_dir(l_pac, a->rb, ma->a[i].rb, &dist);
    if (mode == ORC_RESCUE_SCALA) dist = (int64_t)(int32_t)dist; /* PE:1137-1138 narrows to Int */
    if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
  }
  if (mode == ORC_RESCUE_C && skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
  /* PE:1152: the Scala early return is a discarded expression; the loop below then does nothing */

  regvec_t upd = {0, 0, 0}; /* Scala: mateRegsUpdated, PE:1155-1161 (never replaced by the dedup result, B3) */
  if (mode == ORC_RESCUE_SCALA)
    for (i = 0; i < ma->n; ++i) rv_push(&upd, &ma->a[i]);
  regvec_t last = {0, 0, 0}; /* Scala: regArray after the most recent dedup */

  uint8_t *rev = (uint8_t *)malloc((size_t)l_ms + 1);
  for (r = 0; r < 4; ++r) {
    if (skip[r]) continue;
    const int is_rev = (r >> 1) != (r & 1);
    const uint8_t *seq = ms;
    if (is_rev) {
      for (i = 0; i < l_ms; ++i) rev[l_ms - 1 - i] = ms[i] < 4 ? (uint8_t)(3 - ms[i]) : 4;
      seq = rev;
    }
    if (rlen[r] == rre[r] - rrb[r]) { /* "no funny things happening" */
      int32_t aln[7];
      const int xtra = ORC_KSW_XSUBO | ORC_KSW_XSTART | (l_ms * opt->a < 250 ? ORC_KSW_XBYTE : 0) | (opt->min_seed_len * opt->a);
      orc_sw_align2(l_ms, seq, (int)rlen[r], ref_pool + roff[r], 5, opt->mat, opt->a, opt->b, opt->o_del,
                    opt->e_del, opt->o_ins, opt->e_ins, xtra, aln, cells);
      if (n_sw) ++*n_sw;
      if (aln[0] >= opt->min_seed_len && aln[6] >= 0) {
        orc_alnreg_t b;
        memset(&b, 0, sizeof b);
        if (is_rev) {
          b.qb = l_ms - (aln[2] + 1); b.qe = l_ms - aln[6];
          b.rb = (l_pac << 1) - (rrb[r] + aln[1] + 1); b.re = (l_pac << 1) - (rrb[r] + aln[5]);
        } else if (mode == ORC_RESCUE_C) {
          b.qb = aln[6]; b.qe = aln[2] + 1;
          b.rb = rrb[r] + aln[5]; b.re = rrb[r] + aln[1] + 1; /* native/bwamem_pair.c:206-207 */
        } else {
          b.qb = aln[6]; b.qe = aln[2] + 1;
          b.rb = rrb[r] + aln[1] + 1; b.re = rrb[r] + aln[1] + 1; /* PE:1203-1204 (B2) */
        }
        b.score = aln[0]; b.csub = aln[3]; b.secondary = -1;
        b.seedcov = (int32_t)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : (int64_t)(b.qe - b.qb)) >> 1);
        if (mode == ORC_RESCUE_C) { /* sorted insert, native/bwamem_pair.c:213-219 */
          rv_push(ma, &b);
          for (i = 0; i < ma->n - 1; ++i)
            if (ma->a[i].score < b.score) break;
          const int tmp = i;
          for (i = ma->n - 1; i > tmp; --i) ma->a[i] = ma->a[i - 1];
          ma->a[i] = b;
        } else {
          rv_push(&upd, &b); /* PE:1215 */
        }
      }
      ++n;
    }
    if (n) {
      if (mode == ORC_RESCUE_C) {
        ma->n = orc_sort_dedup(ma->n, ma->a, opt->mask_level_redun, mode);
      } else { /* PE:1221-1229: stable sortBy(score) ascending, then dedup a copy */
        for (i = 1; i < upd.n; ++i)
          for (int j = i; j > 0 && upd.a[j].score < upd.a[j - 1].score; --j) { orc_alnreg_t x = upd.a[j]; upd.a[j] = upd.a[j - 1]; upd.a[j - 1] = x; }
        last.n = 0;
        for (i = 0; i < upd.n; ++i) rv_push(&last, &upd.a[i]);
        /* DEDUP mutates the shared objects (qEnd = qBeg) that mateRegsUpdated still references:
         * mirror that by running the dedup on `last` and copying the kill marks back by identity. */
        {
          /* tag each element with its index in upd through the (otherwise unused here) hash field */
          uint64_t *save = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(upd.n ? upd.n : 1));
          for (i = 0; i < upd.n; ++i) { save[i] = last.a[i].hash; last.a[i].hash = (uint64_t)i; }
          /* run the marking phase on a scratch copy so we can observe which objects were killed */
          regvec_t scratch = {0, 0, 0};
          for (i = 0; i < last.n; ++i) rv_push(&scratch, &last.a[i]);
          /* marking phase == orc_sort_dedup internals; replicate by calling it and diffing */
          int kept = orc_sort_dedup(scratch.n, scratch.a, opt->mask_level_redun, mode);
          char *alive = (char *)calloc((size_t)(upd.n ? upd.n : 1), 1);
          for (i = 0; i < kept; ++i) alive[scratch.a[i].hash] = 1;
          for (i = 0; i < upd.n; ++i)
            if (!alive[i]) upd.a[i].qe = upd.a[i].qb; /* killed objects stay killed in mateRegsUpdated */
          last.n = 0;
          for (i = 0; i < kept; ++i) { scratch.a[i].hash = save[scratch.a[i].hash]; rv_push(&last, &scratch.a[i]); }
          free(scratch.a); free(alive); free(save);
        }
      }
    }
  }
  free(rev);
  if (mode == ORC_RESCUE_SCALA) {
    if (n > 0) { /* PE:1236 */
      ma->n = 0;
      for (i = 0; i < last.n; ++i) rv_push(ma, &last.a[i]);
    }
    free(upd.a); free(last.a);
  }
  return n;
}

/* native/bwamem_pair.c:115-156 ; PE:1256-1317 + PE:1335-1369 */
int64_t orc_matesw_group(const orc_opt_t *opt, int64_t l_pac, const orc_pestat_t pes[4], int group_size,
                         const int32_t *seq_len, const int64_t *seq_off, const uint8_t *seq_pool,
                         const int32_t *reg_cnt, const orc_alnreg_t *regs, const int32_t *ref_cnt,
                         const int64_t *ref_rb, const int64_t *ref_re, const int64_t *ref_len,
                         const int64_t *ref_off, const uint8_t *ref_pool, int mode, int32_t *out_cnt,
                         orc_alnreg_t *out_regs, int64_t out_cap, int64_t *n_sw, int64_t *cells) {
  int64_t reg_base = 0, ref_base = 0, total = 0;
  int overflow = 0;
  for (int k = 0; k < group_size; ++k) {
    regvec_t v[2] = {{0, 0, 0}, {0, 0, 0}}, tmp[2] = {{0, 0, 0}, {0, 0, 0}};
    int64_t rbase[2];
    for (int i = 0; i < 2; ++i) {
      const int cnt = reg_cnt[2 * k + i];
      for (int j = 0; j < cnt; ++j) rv_push(&v[i], &regs[reg_base + j]);
      reg_base += cnt;
      rbase[i] = ref_base;
      ref_base += ref_cnt[2 * k + i];
      for (int j = 0; j < v[i].n; ++j) /* filtered copy taken before any rescue */
        if (v[i].a[j].score >= v[i].a[0].score - opt->pen_unpaired) rv_push(&tmp[i], &v[i].a[j]);
    }
    for (int i = 0; i < 2 && (opt->flag & 0x20) == 0; ++i) /* MEM_F_NO_RESCUE, native/bwamem.h:18 */
      for (int j = 0; j < tmp[i].n && j < opt->max_matesw; ++j) {
        const int64_t rj = (rbase[i] + j) * 4;
        matesw_precompute(opt, l_pac, pes, &tmp[i].a[j], seq_len[2 * k + !i], seq_pool + seq_off[2 * k + !i],
                          &v[!i], ref_rb + rj, ref_re + rj, ref_len + rj, ref_off + rj, ref_pool, mode,
                          n_sw, cells);
      }
    for (int i = 0; i < 2; ++i) {
      out_cnt[2 * k + i] = v[i].n;
      for (int j = 0; j < v[i].n; ++j) {
        if (total < out_cap) out_regs[total] = v[i].a[j]; else overflow = 1;
        ++total;
      }
      free(v[i].a); free(tmp[i].a);
    }
  }
  return overflow ? -total : total;
}

/* ------------------------------------------------------------------ bnsGetSeq */
/* util/BNTSeqUtil.scala:37-79 */
int64_t orc_bns_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, uint8_t *out, int64_t cap) {
  int64_t b = beg, e = end; /* BNTSeqUtil.scala:38-47: swap if end < beg */
  if (end < beg) { e = beg; b = end; }
  if (e > (l_pac << 1)) e = l_pac << 1; /* :48-49 */
  if (b < 0) b = 0;
  int64_t rlen = e - b; /* :50 */
  if (rlen < 0) rlen = 0; /* both ends past 2*l_pac: the Scala would throw on a negative array size; never reached */
  if (!(b >= l_pac || e <= l_pac)) return 0; /* :75-76 bridging the forward-reverse boundary */
  if (rlen > cap) return -rlen;
  int64_t l = 0, k;
  if (b >= l_pac) { /* reverse strand, :56-65 */
    const int64_t beg_f = (l_pac << 1) - 1 - e, end_f = (l_pac << 1) - 1 - b;
    for (k = end_f; k >= beg_f + 1; --k) out[l++] = (uint8_t)((3 - (pac[k >> 2] >> ((~k & 3) << 1))) & 3);
  } else { /* :66-73 */
    for (k = b; k < e; ++k) out[l++] = (uint8_t)((pac[k >> 2] >> ((~k & 3) << 1)) & 3);
  }
  return rlen;
}

/* ------------------------------------------------------------------ memChainToAlnBatched */
static int cal_max_gap(const orc_opt_t *o, int qlen) { /* C2AB:625-643 */
  const int len_del = dtoi((double)(qlen * o->a - o->o_del) / (double)o->e_del + 1.0);
  const int len_ins = dtoi((double)(qlen * o->a - o->o_ins) / (double)o->e_ins + 1.0);
  int len = len_del > len_ins ? len_del : len_ins;
  if (len <= 1) len = 1;
  const int tmp = o->w << 1;
  return len < tmp ? len : tmp;
}

typedef struct { int32_t len, index; } srt_t;
static int srt_cmp(const void *a, const void *b) { /* sortBy(s => (s.len, s.index)), C2AB:372 */
  const srt_t *x = (const srt_t *)a, *y = (const srt_t *)b;
  if (x->len != y->len) return x->len < y->len ? -1 : 1;
  return x->index < y->index ? -1 : (x->index > y->index);
}
#define SRT_MARKED (-2) /* C2AB:51 */

/* one chain of one read; regs[0..*n_regs) are the regions the read already has (all its earlier chains included) */
static void chain2aln(const orc_opt_t *o, int zdrop_mode, int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *query,
                      int n_seeds, const int64_t *s_rbeg, const int32_t *s_qbeg, const int32_t *s_len, orc_alnreg_t *regs,
                      int *n_regs, int64_t *n_ext, int64_t *cells) {
  if (n_seeds == 0) return;
  int i, k;
  /* getMaxSpan, C2AB:648-676 */
  int64_t rmax0 = l_pac << 1, rmax1 = 0;
  for (i = 0; i < n_seeds; ++i) {
    const int64_t b = s_rbeg[i] - (s_qbeg[i] + cal_max_gap(o, s_qbeg[i]));
    const int64_t e = s_rbeg[i] + s_len[i] + (l_query - s_qbeg[i] - s_len[i]) + cal_max_gap(o, l_query - s_qbeg[i] - s_len[i]);
    if (rmax0 > b) rmax0 = b;
    if (rmax1 < e) rmax1 = e;
  }
  if (rmax0 <= 0) rmax0 = 0;
  if (rmax1 >= (l_pac << 1)) rmax1 = l_pac << 1;
  if (rmax0 < l_pac && l_pac < rmax1) { /* crossing the forward-reverse boundary: choose the side of seed 0 */
    if (s_rbeg[0] < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
  }
  /* calPreResultsOfSW, C2AB:344-375 */
  const int64_t span = rmax1 - rmax0;
  uint8_t *rseq = (uint8_t *)malloc((size_t)(span > 0 ? span : 1));
  const int64_t rlen = orc_bns_get_seq(l_pac, pac, rmax0, rmax1, rseq, span > 0 ? span : 0);
  (void)rlen; /* the Scala asserts rlen == rmax(1) - rmax(0) */
  srt_t *srt = (srt_t *)malloc(sizeof(srt_t) * (size_t)n_seeds);
  for (i = 0; i < n_seeds; ++i) { srt[i].len = s_len[i]; srt[i].index = i; }
  qsort(srt, (size_t)n_seeds, sizeof(srt_t), srt_cmp);
  uint8_t *lq = (uint8_t *)malloc((size_t)l_query + 1), *lr = (uint8_t *)malloc((size_t)(span > 0 ? span : 1));

  for (k = n_seeds - 1; k >= 0; --k) { /* one round per seed, longest first (C2AB:405-409, 420-455) */
    const int si = srt[k].index;
    const int64_t srb = s_rbeg[si];
    const int sqb = s_qbeg[si], sl = s_len[si];
    /* testExtension, C2AB:680-741 */
    int ext = *n_regs;
    for (i = 0; i < *n_regs; ++i) {
      const orc_alnreg_t *p = &regs[i];
      if (srb >= p->rb && srb + sl <= p->re && sqb >= p->qb && sqb + sl <= p->qe) {
        int qd = sqb - p->qb;
        int64_t rd = srb - p->rb;
        int mind = qd < rd ? qd : (int)rd;
        int mg = cal_max_gap(o, mind);
        int w = mg < o->w ? mg : o->w;
        if (qd - rd < w && rd - qd < w) { ext = i; break; }
        qd = p->qe - (sqb + sl);
        rd = p->re - (srb + sl);
        mind = qd < rd ? qd : (int)rd;
        mg = cal_max_gap(o, mind);
        w = mg < o->w ? mg : o->w;
        if (qd - rd < w && rd - qd < w) { ext = i; break; }
      }
    }
    if (ext < *n_regs) { /* checkOverlapping, C2AB:753-787 */
      int ovl = n_seeds;
      for (i = k + 1; i < n_seeds; ++i) {
        if (srt[i].index == SRT_MARKED) continue;
        const int ti = srt[i].index;
        if ((double)s_len[ti] >= (double)sl * 0.95) {
          if (sqb <= s_qbeg[ti] && sqb + sl - s_qbeg[ti] >= (sl >> 2) && (int64_t)(s_qbeg[ti] - sqb) != s_rbeg[ti] - srb) { ovl = i; break; }
          if (s_qbeg[ti] <= sqb && s_qbeg[ti] + s_len[ti] - sqb >= (sl >> 2) && (int64_t)(sqb - s_qbeg[ti]) != srb - s_rbeg[ti]) { ovl = i; break; }
        }
      }
      if (ovl == n_seeds) { srt[k].index = SRT_MARKED; continue; } /* C2AB:482-484 */
    }
    orc_alnreg_t reg; /* C2AB:486-499 */
    memset(&reg, 0, sizeof reg);
    reg.w = o->w;
    reg.score = sl * o->a; reg.truesc = sl * o->a;
    reg.qb = 0; reg.rb = srb; reg.qe = l_query; reg.re = srb + sl;
    if (sqb > 0 || sqb + sl != l_query) { /* C2AB:500-562 */
      orc_ext_param_t p;
      memset(&p, 0, sizeof p);
      p.left_qlen = sqb;
      if (p.left_qlen > 0) {
        for (i = 0; i < p.left_qlen; ++i) lq[i] = query[p.left_qlen - 1 - i];
        p.left_rlen = (int32_t)(srb - rmax0);
        for (i = 0; i < p.left_rlen; ++i) lr[i] = rseq[p.left_rlen - 1 - i];
        p.left_qs = lq; p.left_rs = lr;
      }
      const int qe = sqb + sl;
      p.right_qlen = l_query - qe;
      if (p.right_qlen > 0) {
        const int64_t re = srb + sl - rmax0;
        p.right_rlen = (int32_t)(rmax1 - rmax0 - re);
        p.right_qs = query + qe; p.right_rs = rseq + re;
      }
      p.w = o->w; p.mat = o->mat; p.o_del = o->o_del; p.o_ins = o->o_ins; p.e_del = o->e_del; p.e_ins = o->e_ins;
      p.pen_clip5 = o->pen_clip5; p.pen_clip3 = o->pen_clip3; p.zdrop = o->zdrop;
      p.h0 = sl * o->a; p.reg_score = reg.score; p.q_beg = sqb; p.idx = 0;
      orc_ext_ret_t r;
      orc_extension(&p, zdrop_mode, &r, cells);
      if (n_ext) ++*n_ext;
      reg.qb = r.q_beg; /* C2AB:590-599 */
      reg.rb = r.r_beg + srb;
      reg.qe = r.q_end + sqb + sl;
      reg.re = r.r_end + srb + sl;
      reg.score = r.score; reg.truesc = r.true_score; reg.w = r.width;
    }
    int cov = 0; /* computeSeedCoverage, C2AB:891-907 */
    for (i = 0; i < n_seeds; ++i)
      if (s_qbeg[i] >= reg.qb && s_qbeg[i] + s_len[i] <= reg.qe && s_rbeg[i] >= reg.rb && s_rbeg[i] + s_len[i] <= reg.re) cov += s_len[i];
    reg.seedcov = cov;
    regs[(*n_regs)++] = reg;
  }
  free(lq); free(lr); free(srt); free(rseq);
}

int64_t orc_chain2aln_batch(const orc_opt_t *opt, int zdrop_mode, int64_t l_pac, const uint8_t *pac, int n_reads,
                            const int32_t *read_len, const int64_t *read_off, const uint8_t *read_pool,
                            const int32_t *chain_cnt, const int32_t *seed_cnt, const int64_t *seed_rbeg,
                            const int32_t *seed_qbeg, const int32_t *seed_len, int32_t *out_cnt, orc_alnreg_t *out_regs,
                            int64_t out_cap, int64_t *n_ext, int64_t *cells) {
  int64_t total = 0, chain_at = 0, seed_at = 0;
  int overflow = 0;
  for (int r = 0; r < n_reads; ++r) {
    int64_t nseeds = 0;
    for (int c = 0; c < chain_cnt[r]; ++c) nseeds += seed_cnt[chain_at + c];
    orc_alnreg_t *regs = (orc_alnreg_t *)malloc(sizeof(orc_alnreg_t) * (size_t)(nseeds > 0 ? nseeds : 1)); /* maxLength, W1B:117 */
    int n_regs = 0;
    for (int c = 0; c < chain_cnt[r]; ++c) {
      const int ns = seed_cnt[chain_at + c];
      chain2aln(opt, zdrop_mode, l_pac, pac, read_len[r], read_pool + read_off[r], ns, seed_rbeg + seed_at, seed_qbeg + seed_at,
                seed_len + seed_at, regs, &n_regs, n_ext, cells);
      seed_at += ns;
    }
    chain_at += chain_cnt[r];
    out_cnt[r] = n_regs;
    for (int j = 0; j < n_regs; ++j) {
      if (total < out_cap) out_regs[total] = regs[j]; else overflow = 1;
      ++total;
    }
    free(regs);
  }
  return overflow ? -total : total;
}
