// bpsw_reg2aln.hip -- memRegToAln on the device, gfx950: one (read, region) job per wavefront.
//
// Follows worker2/MemRegToADAMSAM.scala (R2S): memRegToAln :172-313, inferBw :127-140, bwaFixXref2 :624-719,
// bwaGenCigar2 :738-891 (C twins: native/bwamem.c:949-1021, :706-713, native/bwa.c:179-222, :89-171).  The job's query
// segment and its reference window are staged in LDS straight from the read bytes and the 2-bit reference resident in
// HBM (bpsw_ref_load; nothing is materialised on the host), reversed on the fly for reverse-strand hits (R2S:760-779);
// the DP is the banded global alignment of bpsw_global_core.h.  What the kernel leaves to the host (bpsw_tail.cpp) is
// the part of mem_aln_t that needs no sequence: flag, mapq (double log), score, sub.
//
// Everything outside the lane-parallel loops is wave-uniform control on the scalar unit: the band retry (up to three
// global alignments, R2S:222-245), the CIGAR walk of bwaFixXref2, NM/MD, the deletion squeeze and the clipping.
#include "bpsw_global_core.h"

namespace bpsw {
namespace {

constexpr int WAVES_PER_BLOCK = 4;

__device__ __forceinline__ int pac_base(const uint8_t* __restrict__ pac, const long long l_pac, const long long pos) {
  const bool rev = pos >= l_pac;  // bnsGetSeq, util/BNTSeqUtil.scala:56-73
  const long long k = rev ? (l_pac << 1) - 1 - pos : pos;
  const int b = (pac[k >> 2] >> ((~k & 3) << 1)) & 3;
  return rev ? 3 - b : b;
}

// bnsPosToRid (== bns_pos2rid, native/bntseq.c:316-331); every lane walks the same probes
__device__ __forceinline__ int pos2rid(const int n_seqs, const long long* __restrict__ ann_off, const long long l_pac,
                                       const long long pos_f) {
  if (pos_f >= l_pac) return -1;
  int left = 0, mid = 0, right = n_seqs;
  while (left < right) {
    mid = (left + right) >> 1;
    if (pos_f >= ann_off[mid]) {
      if (mid == n_seqs - 1) break;
      if (pos_f < ann_off[mid + 1]) break;
      left = mid + 1;
    } else {
      right = mid;
    }
  }
  return mid;
}

__device__ __forceinline__ int infer_bw(int l1, int l2, int score, int a, int q, int r) {  // R2S:127-140
  if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
  int w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / (double)r + 2.);
  const int d = l1 < l2 ? l2 - l1 : l1 - l2;
  return w < d ? d : w;
}

__device__ __forceinline__ int wave_sum(int v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct WaveLds {
  uint8_t* qs;    // the query segment as aligned (reversed for reverse-strand hits)
  uint8_t* rs;    // the reference window, likewise
  int2* eh;
  int8_t* qp;
  uint32_t* cig;  // CIG_LDS operations, last one first
  uint8_t* md;    // md_cap bytes of MD text
};

struct GenRes {
  int have;   // 0: bwaGenCigar2 returned null (R2S:748, :755)
  int score, n, rlen;
};

// bwaGenCigar2 up to (not including) NM/MD, R2S:738-806.  The operations end up in L.cig, last one first.
__device__ GenRes gen_cigar(const int lane, const Reg2AlnDev& J, const SwScoring& sc, const WaveLds& L, uint8_t* __restrict__ z,
                            const uint8_t* __restrict__ read, const int qb, const int qe, const long long rb, const long long re,
                            const int w_) {
  GenRes R;
  R.have = 0; R.score = 0; R.n = 0; R.rlen = 0;
  const int lq = qe - qb;
  if (lq <= 0 || rb >= re || (rb < J.l_pac && re > J.l_pac)) return R;  // R2S:748
  if (re > (J.l_pac << 1)) return R;                                     // bnsGetSeq would clamp: rlen != re - rb, R2S:755
  const int rlen = (int)(re - rb);
  R.have = 1; R.rlen = rlen;
  const bool rev = rb >= J.l_pac;  // reverse both so that indels are placed leftmost, R2S:760-779
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < lq; j += 64) {
    int c = read[qb + (rev ? lq - 1 - j : j)];
    L.qs[j] = (uint8_t)(c > 4 ? 4 : c);
  }
  for (int i = lane; i < rlen; i += 64) L.rs[i] = (uint8_t)pac_base(J.pac, J.l_pac, rb + (rev ? rlen - 1 - i : i));
  __builtin_amdgcn_wave_barrier();
  if (lq == rlen && w_ == 0) {  // no gap, no DP: R2S:781-794
    int s = 0;
    for (int i = lane; i < lq; i += 64) {
      const int r = L.rs[i], q = L.qs[i];
      s += (int)(int8_t)((sc.mat.row[r] >> (8 * q)) & 0xff);
    }
    R.score = wave_sum(s);
    R.n = 1;
    if (lane == 0) L.cig[0] = (uint32_t)lq << 4;
    __builtin_amdgcn_wave_barrier();
    return R;
  }
  const int mat0 = (int)(int8_t)(sc.mat.row[0] & 0xff);
  const int max_ins = (int)((double)(((lq + 1) >> 1) * mat0 - sc.o_ins) / (double)sc.e_ins + 1.);
  const int max_del = (int)((double)(((lq + 1) >> 1) * mat0 - sc.o_del) / (double)sc.e_del + 1.);
  int max_gap = max_ins > max_del ? max_ins : max_del, w;
  const int d = rlen - lq;
  if (J.flavour == BPSW_TAIL_C) {  // native/bwa.c:120-122
    max_gap = max_gap > 1 ? max_gap : 1;
    w = (max_gap + (d < 0 ? -d : d) + 1) >> 1;
  } else {                         // R2S:796-799: no clamp, abs((rlen - queryLen) + 1)
    w = (max_gap + (d + 1 < 0 ? -(d + 1) : d + 1)) >> 1;
  }
  w = w < w_ ? w : w_;
  const int min_w = (d < 0 ? -d : d) + 3;
  w = w > min_w ? w : min_w;
  const int nCol = lq < 2 * w + 1 ? lq : 2 * w + 1;  // SWUtil.scala:248-249
  global_init(lane, lq, w, sc, L.qs, L.eh, L.qp);
  R.score = global_rows(lane, lq, rlen, w, sc, L.rs, L.eh, L.qp, z, nCol);
  R.n = global_backtrack(lane, lq, rlen, w, z, nCol, L.cig);
  return R;
}

__device__ __forceinline__ int md_put_num(const int lane, uint8_t* __restrict__ md, const int cap, int at, int v) {  // kputw
  int digits = 1;
  for (int t = v; t >= 10; t /= 10) ++digits;
  if (lane == 0) {
    int t = v;
    for (int k = digits - 1; k >= 0; --k) {
      if (at + k < cap) md[at + k] = (uint8_t)('0' + t % 10);
      t /= 10;
    }
  }
  return at + digits;
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void reg2aln_kernel(const Reg2AlnDev J, const SwScoring sc,
                                                                         Reg2AlnOut* __restrict__ out,
                                                                         uint32_t* __restrict__ out_cigar,
                                                                         uint8_t* __restrict__ out_md,
                                                                         uint8_t* __restrict__ zscratch,
                                                                         const unsigned long long z_per_wave, const int qcap,
                                                                         const int rcap, const int md_cap,
                                                                         const int lds_per_wave) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const int slot = uni((int)blockIdx.x * WAVES_PER_BLOCK + wave);
  unsigned char* base = smem + (size_t)wave * lds_per_wave;
  WaveLds L;
  L.eh = reinterpret_cast<int2*>(base);                                    // qcap + 2 entries
  L.cig = reinterpret_cast<uint32_t*>(base + 8 * (size_t)(qcap + 2));      // CIG_LDS entries
  L.qp = reinterpret_cast<int8_t*>(L.cig + CIG_LDS);                        // 5 x qcap
  L.qs = reinterpret_cast<uint8_t*>(L.qp + 5 * (size_t)qcap);               // qcap
  L.rs = L.qs + qcap;                                                       // rcap
  L.md = L.rs + rcap;                                                       // md_cap
  uint8_t* z = zscratch + (size_t)slot * z_per_wave;
  const int stride = gridDim.x * WAVES_PER_BLOCK;
  const long long l_pac = J.l_pac;

  for (int job = slot; job < J.n; job += stride) {
    const bpsw_alnreg_t* __restrict__ ar = J.regs + job;
    const int l_query = uni(J.read_len[job]);
    const uint8_t* __restrict__ read = J.read_pool + J.read_off[job];
    int qb = uni(ar->qb), qe = uni(ar->qe);
    long long rb = ar->rb, re = ar->re;
    rb = ((long long)uni((int)(rb >> 32)) << 32) | (unsigned)uni((int)rb);
    re = ((long long)uni((int)(re >> 32)) << 32) | (unsigned)uni((int)re);
    const int truesc = uni(ar->truesc), width = uni(ar->w);
    Reg2AlnOut o;
    o.pos = -1; o.rid = -1; o.is_rev = 0; o.NM = 0; o.n_cigar = 0; o.md_len = 0; o.status = 0; o.gscore = 0;

    // ---- bwaFixXref2, R2S:624-719 -------------------------------------------------------------------------------
    bool ok = !(rb < l_pac && re > l_pac);
    if (ok) {
      const long long mid = (rb + re) >> 1;
      const bool mrev = mid >= l_pac;
      const long long fm = mrev ? (l_pac << 1) - 1 - mid : mid;
      const int rid = pos2rid(J.n_seqs, J.ann_off, l_pac, fm);
      const long long a_off = J.ann_off[rid], a_len = J.ann_len[rid];
      long long cb = mrev ? (l_pac << 1) - (a_off + a_len) : a_off;
      long long ce = cb + a_len;
      if (cb > rb || ce < re) {  // the hit runs over the end of its chromosome: cut it there
        cb = cb > rb ? cb : rb;
        ce = ce < re ? ce : re;
        const GenRes g = gen_cigar(lane, J, sc, L, z, read, qb, qe, rb, re, J.opt_w);
        const int n = g.n <= CIG_LDS ? g.n : 0;
        long long x = rb;
        int y = qb;
        int nqb = qb, nqe = qe;
        long long nrb = rb, nre = re;
        for (int i = 0; i < n; ++i) {
          const uint32_t c = L.cig[n - 1 - i];
          const int op = (int)(c & 0xf), len = (int)(c >> 4);
          if (op == 0) {
            if (x <= cb && cb < x + len) { nqb = (int)(y + (cb - x)); nrb = cb; }
            if (x < ce && ce <= x + len) { nqe = (int)(y + (ce - x)); nre = ce; break; }
            x += len; y += len;
          } else if (op == 1) {
            y += len;
          } else {
            if (x <= cb && cb < x + len) { nqb = y; nrb = x + len; }
            if (x < ce && ce <= x + len) { nqe = y; nre = x; break; }
            x += len;
          }
        }
        qb = nqb; qe = nqe; rb = nrb; re = nre;
        __builtin_amdgcn_wave_barrier();
      }
      ok = !(qb == qe || rb == re);
    }
    if (!ok) {  // the Scala asserts here (R2S:196-199), the C exits: reported, never silently patched
      o.status = BPSW_ALN_XREF;
      if (lane == 0) out[job] = o;
      continue;
    }

    // ---- band inference and the retry loop, R2S:201-245 ------------------------------------------------------------
    int tmp = infer_bw(qe - qb, (int)(re - rb), truesc, J.a, sc.o_del, sc.e_del);
    int w2 = infer_bw(qe - qb, (int)(re - rb), truesc, J.a, sc.o_ins, sc.e_ins);
    w2 = w2 > tmp ? w2 : tmp;
    if (w2 > J.opt_w) w2 = w2 < width ? w2 : width;
    GenRes g;
    int last_sc = -(1 << 30), it = 0;
    do {
      g = gen_cigar(lane, J, sc, L, z, read, qb, qe, rb, re, w2);
      if (g.score == last_sc) break;
      last_sc = g.score;
      w2 <<= 1;
    } while (++it < 3 && g.score < truesc - J.a);
    o.gscore = g.score;
    const int n = g.n;
    if (!g.have) o.status = BPSW_ALN_NOCIGAR;
    if (n > CIG_LDS) o.status = BPSW_ALN_OVERFLOW;
    const int lq = qe - qb;

    // ---- NM and MD, R2S:808-869 (on the alignment as computed: both sequences possibly reversed) -----------------
    int mdl = 0, n_mm = 0, n_gap = 0;
    if (g.have && n <= CIG_LDS) {
      const bool fwd = rb < l_pac;  // int2base, R2S:817-818
      int x = 0, y = 0, u = 0;
      for (int k = 0; k < n; ++k) {
        const uint32_t c = L.cig[n - 1 - k];
        const int op = (int)(c & 0xf), len = (int)(c >> 4);
        if (op == 0) {
          for (int i0 = 0; i0 < len; i0 += 64) {
            const int i = i0 + lane;
            const bool mis = i < len && L.qs[x + i] != L.rs[y + i];
            unsigned long long m = __builtin_amdgcn_ballot_w64(mis);
            int last = i0;
            while (m) {
              const int b = __builtin_ctzll(m);
              m &= m - 1;
              u += i0 + b - last;
              mdl = md_put_num(lane, L.md, md_cap, mdl, u);
              const int rc = L.rs[y + i0 + b];
              if (lane == 0 && mdl < md_cap) L.md[mdl] = (uint8_t)(fwd ? "ACGTN"[rc] : "TGCAN"[rc]);
              ++mdl; ++n_mm; u = 0;
              last = i0 + b + 1;
            }
            u += min(len, i0 + 64) - last;
          }
          x += len; y += len;
        } else if (op == 2) {
          if (k > 0 && k < n - 1) {  // not for a leading or trailing deletion
            mdl = md_put_num(lane, L.md, md_cap, mdl, u);
            if (lane == 0 && mdl < md_cap) L.md[mdl] = (uint8_t)'^';
            ++mdl;
            for (int i = lane; i < len; i += 64) {
              const int rc = L.rs[y + i];
              if (mdl + i < md_cap) L.md[mdl + i] = (uint8_t)(fwd ? "ACGTN"[rc] : "TGCAN"[rc]);
            }
            mdl += len;
            u = 0; n_gap += len;
          }
          y += len;
        } else {
          x += len; n_gap += len;
        }
      }
      mdl = md_put_num(lane, L.md, md_cap, mdl, u);
      __builtin_amdgcn_wave_barrier();
    }
    o.NM = n_mm + n_gap;
    o.md_len = mdl;
    if (mdl > md_cap && o.status == 0) o.status = BPSW_ALN_OVERFLOW;

    // ---- position, deletion squeeze, clipping, R2S:248-303 ---------------------------------------------------------
    const long long p0 = rb < l_pac ? rb : re - 1;
    const bool is_rev = p0 >= l_pac;
    long long pos = is_rev ? (l_pac << 1) - 1 - p0 : p0;  // bnsDepos
    int first = 0, cnt = n <= CIG_LDS ? n : 0;            // operations [first, first + cnt) in forward order
    if (cnt > 0) {
      const uint32_t c0 = L.cig[n - 1], c1 = L.cig[0];
      if ((c0 & 0xf) == 2) { pos += (long long)(c0 >> 4); first = 1; --cnt; }
      else if ((c1 & 0xf) == 2) --cnt;
    }
    int clip5 = 0, clip3 = 0;
    if (qb != 0 || qe != l_query) {
      clip5 = is_rev ? l_query - qe : qb;
      clip3 = is_rev ? qb : l_query - qe;
    }
    const int n_out = (clip5 > 0) + cnt + (clip3 > 0);
    uint32_t* oc = out_cigar + (size_t)job * J.max_cigar;
    if (n_out <= J.max_cigar) {
      const int sh = clip5 > 0 ? 1 : 0;
      if (lane == 0 && clip5 > 0) oc[0] = (uint32_t)clip5 << 4 | 3u;
      for (int f = lane; f < cnt; f += 64) oc[sh + f] = L.cig[n - 1 - (first + f)];
      if (lane == 0 && clip3 > 0) oc[sh + cnt] = (uint32_t)clip3 << 4 | 3u;
    }
    uint8_t* om = out_md + (size_t)job * J.max_md;
    for (int i = lane; i < mdl && i < J.max_md && i < md_cap; i += 64) om[i] = L.md[i];
    o.n_cigar = n_out;
    o.is_rev = is_rev ? 1 : 0;
    o.rid = pos2rid(J.n_seqs, J.ann_off, l_pac, pos);
    o.pos = o.rid >= 0 ? pos - J.ann_off[o.rid] : pos;
    (void)lq;
    if (lane == 0) out[job] = o;
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

size_t reg2aln_lds_per_wave(int qcap, int rcap, int md_cap) {
  size_t b = 8 * (size_t)(qcap + 2) + 4 * (size_t)CIG_LDS + 5 * (size_t)qcap + (size_t)qcap + (size_t)rcap + (size_t)md_cap;
  return (b + 15) & ~(size_t)15;
}
int reg2aln_resident_waves(int num_cu, int qcap, int rcap, int md_cap) {
  const size_t lds = reg2aln_lds_per_wave(qcap, rcap, md_cap) * WAVES_PER_BLOCK;
  int per_cu = (int)((160 * 1024) / lds);
  per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
  return num_cu * per_cu * WAVES_PER_BLOCK;
}

hipError_t launch_reg2aln_kernel(const Reg2AlnDev& J, const SwScoring& sc, int qcap, int rcap, int md_cap, size_t z_per_wave,
                                 Reg2AlnOut* d_out, uint32_t* d_cigar, uint8_t* d_md, uint8_t* d_z, int num_cu, hipStream_t s) {
  if (J.n <= 0) return hipSuccess;
  const size_t per_wave = reg2aln_lds_per_wave(qcap, rcap, md_cap);
  const size_t lds = per_wave * WAVES_PER_BLOCK;
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  int blocks = (J.n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  const int max_blocks = reg2aln_resident_waves(num_cu, qcap, rcap, md_cap) / WAVES_PER_BLOCK;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(reg2aln_kernel, dim3(blocks), dim3(64 * WAVES_PER_BLOCK), lds, s, J, sc, d_out, d_cigar, d_md, d_z,
                     (unsigned long long)z_per_wave, qcap, rcap, md_cap, (int)per_wave);
  return hipGetLastError();
}

}  // namespace bpsw
