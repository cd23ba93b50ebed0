// bpsw_pack.cpp -- host-side mirror of the Scala batch packer for boundary 2.
//
// runOnFPGAJNI (MemChainToAlignBatched.scala:59-172) turns an Array[ExtParam] into the byte[] that
// crosses the JNI.  Non-JVM callers (the C++/ctypes harness, bench.py) build the same bytes here, so
// the device sees exactly what the Scala driver would send.
#include <string.h>

#include "bpsw_internal.h"

namespace {

inline void put16(uint8_t* b, size_t at, int v) {
  b[at] = (uint8_t)(v & 0xff);
  b[at + 1] = (uint8_t)((v >> 8) & 0xff);
}
inline void put32(uint8_t* b, size_t at, uint32_t v) {
  b[at] = (uint8_t)(v & 0xff);
  b[at + 1] = (uint8_t)((v >> 8) & 0xff);
  b[at + 2] = (uint8_t)((v >> 16) & 0xff);
  b[at + 3] = (uint8_t)((v >> 24) & 0xff);
}
// words of nibble data one task occupies, MemChainToAlignBatched.scala:101
inline size_t seq_words(const bpsw_ext_tasks_t* t, int i) {
  const int bases = t->left_qlen[i] + t->left_rlen[i] + t->right_qlen[i] + t->right_rlen[i];
  return (size_t)((((bases + 1) / 2) + 3) / 4);
}
// ((qlen*max + clip - o).toDouble / e + 1).toInt, MemChainToAlignBatched.scala:106-109
inline int gap_bound(int qlen, int mx, int clip, int o, int e) {
  const double v = (double)(qlen * mx + clip - o) / (double)e + 1.0;
  if (v >= 2147483647.0) return 2147483647;
  if (v <= -2147483648.0) return (int)0x80000000;
  return (int)v;
}

}  // namespace

extern "C" size_t bpsw_wire_size(const bpsw_ext_tasks_t* t) {
  if (!t || t->n < 0) return 0;
  size_t words = (size_t)(32 + 32 * (size_t)t->n) >> 2;
  for (int i = 0; i < t->n; ++i) words += seq_words(t, i);
  return words << 2;
}

extern "C" int bpsw_wire_pack(const bpsw_ext_tasks_t* t, uint8_t* buf, size_t cap, size_t* bytes) {
  if (!t || !buf || t->n < 1) return bpsw::fail(BPSW_ERR_ARG, "bpsw_wire_pack: empty task list");
  const size_t total = bpsw_wire_size(t);
  if (bytes) *bytes = total;
  if (total > cap) return bpsw::fail(BPSW_ERR_CAPACITY, "bpsw_wire_pack: buffer too small");
  if ((total >> 2) > 0x7fffffffull) return bpsw::fail(BPSW_ERR_LIMIT, "bpsw_wire_pack: batch exceeds int32 word offsets");
  memset(buf, 0, total);
  const int n = t->n;
  // header: seven signed bytes then taskNum at byte 8 (MemChainToAlignBatched.scala:78-85)
  buf[0] = (uint8_t)(int8_t)t->o_del; buf[1] = (uint8_t)(int8_t)t->e_del;
  buf[2] = (uint8_t)(int8_t)t->o_ins; buf[3] = (uint8_t)(int8_t)t->e_ins;
  buf[4] = (uint8_t)(int8_t)t->pen_clip5; buf[5] = (uint8_t)(int8_t)t->pen_clip3;
  buf[6] = (uint8_t)(int8_t)t->w;
  put32(buf, 8, (uint32_t)n);

  size_t rec = 32, data = 32 + 32 * (size_t)n;
  for (int i = 0; i < n; ++i, rec += 32) {
    const int lq = t->left_qlen[i], lr = t->left_rlen[i], rq = t->right_qlen[i], rr = t->right_rlen[i];
    put16(buf, rec + 0, lq); put16(buf, rec + 2, lr); put16(buf, rec + 4, rq); put16(buf, rec + 6, rr);
    put32(buf, rec + 8, (uint32_t)(data >> 2));
    put16(buf, rec + 12, t->reg_score[i]); put16(buf, rec + 14, t->q_beg[i]);
    put16(buf, rec + 16, t->h0[i]); put16(buf, rec + 18, t->idx[i]);
    put16(buf, rec + 20, gap_bound(lq, t->mat_max, t->pen_clip5, t->o_ins, t->e_ins));
    put16(buf, rec + 22, gap_bound(lq, t->mat_max, t->pen_clip5, t->o_del, t->e_del));
    put16(buf, rec + 24, gap_bound(rq, t->mat_max, t->pen_clip3, t->o_ins, t->e_ins));
    put16(buf, rec + 26, gap_bound(rq, t->mat_max, t->pen_clip3, t->o_del, t->e_del));
    put32(buf, rec + 28, (uint32_t)t->idx[i]);

    // sequences in the order leftQs, rightQs, leftRs, rightRs (MemChainToAlignBatched.scala:125-161);
    // 8 bases per little-endian int32, first base in the most significant nibble, zero padded per task.
    const uint8_t* seg[4] = {t->pool + (lq ? t->left_q_off[i] : 0), t->pool + (rq ? t->right_q_off[i] : 0),
                             t->pool + (lr ? t->left_r_off[i] : 0), t->pool + (rr ? t->right_r_off[i] : 0)};
    const int len[4] = {lq, rq, lr, rr};
    uint32_t acc = 0;
    int filled = 0;
    for (int s = 0; s < 4; ++s)
      for (int j = 0; j < len[s]; ++j) {
        acc = (acc << 4) | (uint32_t)(seg[s][j] & 0x0F);
        if (++filled == 8) { put32(buf, data, acc); data += 4; acc = 0; filled = 0; }
      }
    if (filled) { acc <<= 4 * (8 - filled); put32(buf, data, acc); data += 4; }
  }
  return BPSW_OK;
}

// ---- coordinate batches (wire format 2, include/bpsw.h): query flanks only, target flanks named by the seed's coordinates ----
namespace {
inline size_t seq_words2(const bpsw_ext_coord_tasks_t* t, int i) {
  const int bases = t->left_qlen[i] + t->right_qlen[i];
  return (size_t)((((bases + 1) / 2) + 3) / 4);
}
}  // namespace

extern "C" size_t bpsw_wire_coords_size(const bpsw_ext_coord_tasks_t* t) {
  if (!t || t->n < 0) return 0;
  size_t words = (size_t)(32 + 40 * (size_t)t->n) >> 2;
  for (int i = 0; i < t->n; ++i) words += seq_words2(t, i);
  return words << 2;
}

extern "C" int bpsw_wire_coords_pack(const bpsw_ext_coord_tasks_t* t, uint8_t* buf, size_t cap, size_t* bytes) {
  if (!t || !buf || t->n < 1) return bpsw::fail(BPSW_ERR_ARG, "bpsw_wire_coords_pack: empty task list");
  const size_t total = bpsw_wire_coords_size(t);
  if (bytes) *bytes = total;
  if (total > cap) return bpsw::fail(BPSW_ERR_CAPACITY, "bpsw_wire_coords_pack: buffer too small");
  if ((total >> 2) > 0x7fffffffull) return bpsw::fail(BPSW_ERR_LIMIT, "bpsw_wire_coords_pack: batch exceeds int32 word offsets");
  memset(buf, 0, total);
  const int n = t->n;
  buf[0] = (uint8_t)(int8_t)t->o_del; buf[1] = (uint8_t)(int8_t)t->e_del;
  buf[2] = (uint8_t)(int8_t)t->o_ins; buf[3] = (uint8_t)(int8_t)t->e_ins;
  buf[4] = (uint8_t)(int8_t)t->pen_clip5; buf[5] = (uint8_t)(int8_t)t->pen_clip3;
  buf[6] = (uint8_t)(int8_t)t->w;
  buf[7] = BPSW_WIRE_COORDS;
  put32(buf, 8, (uint32_t)n);
  size_t rec = 32, data = 32 + 40 * (size_t)n;
  for (int i = 0; i < n; ++i, rec += 40) {
    const int lq = t->left_qlen[i], lr = t->left_rlen[i], rq = t->right_qlen[i], rr = t->right_rlen[i];
    put16(buf, rec + 0, lq); put16(buf, rec + 2, lr); put16(buf, rec + 4, rq); put16(buf, rec + 6, rr);
    put32(buf, rec + 8, (uint32_t)(data >> 2));
    put16(buf, rec + 12, t->reg_score[i]); put16(buf, rec + 14, t->q_beg[i]);
    put16(buf, rec + 16, t->h0[i]); put16(buf, rec + 18, t->seed_len[i]);
    put16(buf, rec + 20, gap_bound(lq, t->mat_max, t->pen_clip5, t->o_ins, t->e_ins));
    put16(buf, rec + 22, gap_bound(lq, t->mat_max, t->pen_clip5, t->o_del, t->e_del));
    put16(buf, rec + 24, gap_bound(rq, t->mat_max, t->pen_clip3, t->o_ins, t->e_ins));
    put16(buf, rec + 26, gap_bound(rq, t->mat_max, t->pen_clip3, t->o_del, t->e_del));
    put32(buf, rec + 28, (uint32_t)t->idx[i]);
    const int64_t rb = t->seed_rbeg[i];
    memcpy(buf + rec + 32, &rb, 8);
    const uint8_t* seg[2] = {t->pool + (lq ? t->left_q_off[i] : 0), t->pool + (rq ? t->right_q_off[i] : 0)};
    const int len[2] = {lq, rq};
    uint32_t acc = 0;
    int filled = 0;
    for (int s = 0; s < 2; ++s)
      for (int j = 0; j < len[s]; ++j) {
        acc = (acc << 4) | (uint32_t)(seg[s][j] & 0x0F);
        if (++filled == 8) { put32(buf, data, acc); data += 4; acc = 0; filled = 0; }
      }
    if (filled) { acc <<= 4 * (8 - filled); put32(buf, data, acc); data += 4; }
  }
  return BPSW_OK;
}
