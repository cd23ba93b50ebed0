// sift_host.cpp -- the sift kernel's arithmetic (csrc/bpsw_extend_sift_core.h) compiled for the HOST, one task after the other, so
// that tests/test_sift_host.py can hold it against the oracle's full DP without a GPU.  Test infrastructure: mirrors what a lane of
// ext_sift_kernel<false> does with a task of a format-1 wire batch (bpsw_extend_sift.hip) -- stage the stream, judge both sides
// (closed form, every shift of its certificate, start-gap form), chain them -- minus the wavefront (no LDS, no lane pairs).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "bpsw_extend_sift_core.h"

using namespace bpsw::sift;

static int lo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
static int hi16(uint32_t v) { return (int)(int16_t)(v >> 16); }

// flag[t]: 0 not examined, 1 record written (out + 10 t), 2 verdicts only; kinds[2 t + side]: SIFT_UNSEEN / _FAIL / _FORM
extern "C" int sift_host_batch(const uint32_t* wire, size_t wire_words, int n, int zdrop, int certify, int exact_a, int dm, int qmax,
                               int16_t* out, uint8_t* flag, uint8_t* kinds) {
  const uint32_t hdr0 = wire[0], hdr1 = wire[1];
  SiftParams P;
  P.oDel = (int8_t)(hdr0 & 0xff); P.eDel = (int8_t)((hdr0 >> 8) & 0xff);
  P.oIns = (int8_t)((hdr0 >> 16) & 0xff); P.eIns = (int8_t)((hdr0 >> 24) & 0xff);
  const int penClip5 = (int8_t)(hdr1 & 0xff), penClip3 = (int8_t)((hdr1 >> 8) & 0xff);
  P.wBand = (int8_t)((hdr1 >> 16) & 0xff);
  P.zdrop = zdrop; P.certify = certify; P.dm = dm;
  const int oe_min = sift_min(P.oIns + P.eIns, P.oDel + P.eDel);
  P.a = (oe_min > 0 && P.wBand >= 2) ? exact_a : 0;
  std::vector<uint32_t> raw;
  for (int t = 0; t < n; ++t) {
    const uint32_t* rec = wire + 8 + 8 * (size_t)t;
    const int lq = lo16(rec[0]), lr = hi16(rec[0]), rq = lo16(rec[1]), rr = hi16(rec[1]);
    const int pos = (int)rec[2];
    const int nwords = (lq + lr + rq + rr + 7) >> 3;
    flag[t] = 0; kinds[2 * t] = kinds[2 * t + 1] = SIFT_UNSEEN;
    if (P.a <= 0 || dm <= 0 || lq > qmax || rq > qmax) continue;
    if ((size_t)pos + (size_t)nwords > wire_words) return -1;
    raw.assign(wire + pos, wire + pos + nwords);
    raw.resize((size_t)nwords + 4, 0u);
    SideRec sr[2] = {{SIFT_UNSEEN, 0, 0, 0, 0, 0, 0, 0}, {SIFT_UNSEEN, 0, 0, 0, 0, 0, 0, 0}};
    for (int side = 0; side < 2; ++side) {
      const int qLen = side ? rq : lq, rLen = side ? rr : lr;
      if (qLen <= 0) continue;
      const SiftSeq s = {raw.data(), side ? lq : 0, raw.data(), side ? lq + rq + lr : lq + rq};
      uint32_t n_seen = 0u;
      for (int j = 0; j < qLen; j += 8) n_seen |= s.q8(j) & top_nibbles(qLen - j) & 0xCCCCCCCCu;
      for (int j = 0; j < rLen; j += 8) n_seen |= s.t8(j) & top_nibbles(rLen - j) & 0xCCCCCCCCu;
      if (n_seen) continue;
      int k = 0, p[3] = {0, 0, 0}, dI = 0, dD = 0;
      int st = sift_closed_form(s, qLen, rLen, P, &sr[side], &k, p, &dI, &dD);
      if (st == CF_IF_CERTIFIED) {
        bool ok = true;
        for (int d = 1; d <= dI && ok; ++d) ok = sift_certificate_shift(s, qLen, rLen, P, k, p[0], p[1], p[2], true, d);
        for (int d = 1; d <= dD && ok; ++d) ok = sift_certificate_shift(s, qLen, rLen, P, k, p[0], p[1], p[2], false, d);
        st = ok ? CF_HOLDS : CF_FAILS;
      }
      if (st == CF_FAILS) {
        sr[side].kind = SIFT_FAIL;
        (void)sift_start_gap_form(s, qLen, rLen, P, &sr[side]);
      } else if (st == CF_UNSEEN) {
        sr[side].kind = SIFT_UNSEEN;
      }
      kinds[2 * t + side] = (uint8_t)sr[side].kind;
    }
    const SiftTask T = {lq, rq, lo16(rec[3]), hi16(rec[3]), lo16(rec[4]), (int)rec[7], penClip5, penClip3, P.wBand};
    uint32_t o[5];
    if (sift_chain(T, sr[0], sr[1], o)) {
      memcpy(out + 10 * (size_t)t, o, 20);
      flag[t] = 1;
    } else {
      flag[t] = 2;
    }
  }
  return 0;
}
