/*
 * bpsw_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded restatement of the CS-BWAMEM Smith-Waterman hot path
 * (the Scala text is the specification).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (libbPSW_hip.so) never links, loads or calls anything in oracle/.
 *
 * Parity status: PINNED.  The reference ships no golden vectors,
 * so the restatement is pinned against outputs of the reference's own C sources
 * compiled in place (oracle/_ref, see oracle/Makefile) -- ksw_extend2,
 * ksw_align2, ksw_global2, mem_sort_and_dedup, mem_group_matesw -- on seeded
 * inputs, and by committed fixtures under tests/golden/ generated from them.
 * Where the Scala text and the C differ (B1 z-drop parse,
 * B2/B3 rescue bookkeeping, B8 second-best) both forms are implemented and
 * selected by a mode argument; fixtures record which form produced them.
 *
 * Reference citations use the SURVEY.md shorthand:
 *   SW   = src/main/scala/cs/ucla/edu/bwaspark/util/SWUtil.scala
 *   C2AB = src/main/scala/cs/ucla/edu/bwaspark/worker1/MemChainToAlignBatched.scala
 *   PE   = src/main/scala/cs/ucla/edu/bwaspark/worker2/MemSamPe.scala
 *   DEDUP= src/main/scala/cs/ucla/edu/bwaspark/worker1/MemSortAndDedup.scala
 *   native/ = src/main/native/
 */
#ifndef BPSW_ORACLE_H
#define BPSW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* z-drop parse: SW:194-199 (Scala: dangling else binds to the inner if) vs
 * native/ksw.c:455-461 (BWA C). */
#define ORC_ZDROP_SCALA 0
#define ORC_ZDROP_BWA 1

/* rescue bookkeeping flavour: native/bwamem_pair.c:159-228 + native/bwamem.c:394-435
 * (what jniNative.so does) vs PE:1111-1238 + DEDUP:33-141 (pure-Scala path). */
#define ORC_RESCUE_C 0
#define ORC_RESCUE_SCALA 1

#define ORC_KSW_XBYTE 0x10000
#define ORC_KSW_XSTOP 0x20000
#define ORC_KSW_XSUBO 0x40000
#define ORC_KSW_XSTART 0x80000

/* ---- SWExtend, SW:61-230.  out = {max, qle, tle, gtle, gscore, max_off} ---- */
void orc_sw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                   int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                   int zdrop_mode, int32_t out[6], int64_t *cells);

/* ---- one extension task, C2AB:789-883 (ExtParam / ExtRet: datatype/ExtensionParameters.scala:21-87) ---- */
typedef struct {
  int32_t left_qlen, left_rlen, right_qlen, right_rlen;
  const uint8_t *left_qs, *left_rs, *right_qs, *right_rs; /* left_* already reversed, C2AB:505-517 */
  int32_t w, o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, zdrop, h0, reg_score, q_beg, idx;
  const int8_t *mat; /* 25 entries */
} orc_ext_param_t;

typedef struct {
  int32_t q_beg, q_end;
  int64_t r_beg, r_end;
  int32_t score, true_score, width, idx;
} orc_ext_ret_t;

void orc_extension(const orc_ext_param_t *p, int zdrop_mode, orc_ext_ret_t *ret, int64_t *cells);

/* ---- boundary-2 wire format, C2AB:76-172 (pack) and C2AB:178-190 (result) ---- */
size_t orc_wire_size(int n, const orc_ext_param_t *tasks);
size_t orc_wire_pack(int n, const orc_ext_param_t *tasks, uint8_t *buf, size_t cap);
/* decode a wire batch, run orc_extension per task, emit 10 int16 per task */
int orc_wire_extend(const uint8_t *wire, size_t bytes, const int8_t mat[25], int zdrop, int zdrop_mode,
                    int16_t *out, int64_t *cells);

/* ---- SWAlign / SWAlign2, SW:417-601.  out = {score,tEnd,qEnd,score2,tEnd2,tBeg,qBeg} ---- */
void orc_sw_align(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                  int a, int b, int o_del, int e_del, int o_ins, int e_ins, int xtra, int32_t out[7],
                  int64_t *cells);
void orc_sw_align2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                   int a, int b, int o_del, int e_del, int o_ins, int e_ins, int xtra, int32_t out[7],
                   int64_t *cells);

/* ---- SWGlobal, SW:233-397.  cigar[k] = len<<4 | op (op: 0=M 1=I 2=D).  returns score ---- */
int orc_sw_global(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                  int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar, uint32_t *cigar,
                  int cigar_cap);

/* ---- region record: datatype/MemAlnRegType.scala:26-38 == mem_alnreg_t native/bwamem.h:49-61 ---- */
typedef struct {
  int64_t rb, re;
  int32_t qb, qe, score, truesc, sub, csub, sub_n, w, seedcov, secondary;
  uint64_t hash;
} orc_alnreg_t; /* 64 bytes */

/* memSortAndDedup.  mode ORC_RESCUE_C: native/bwamem.c:394-435 with the klib introsort
 * (native/ksort.h:176-227); mode ORC_RESCUE_SCALA: DEDUP:33-141 (stable sortBy). returns new n */
int orc_sort_dedup(int n, orc_alnreg_t *a, float mask_level_redun, int mode);

typedef struct {
  int32_t low, high, failed;
  int32_t pad_;
  double avg, std;
} orc_pestat_t; /* datatype/MemPeStat.scala:27-31 */

typedef struct {
  int32_t a, b, o_del, e_del, o_ins, e_ins, pen_unpaired, pen_clip5, pen_clip3, w, zdrop, T, flag,
      min_seed_len, max_ins, max_matesw;
  float mask_level_redun;
  int8_t mat[25];
  int8_t pad_[3];
} orc_opt_t; /* the subset of datatype/MemOptType.scala:28-56 the path reads */

void orc_opt_default(orc_opt_t *opt); /* MemOptType defaults + bwaFillScmat, MemOptType.scala:28-73 */

/* infer_dir, native/bwamem_pair.c:27-34 (inlined at PE:1128-1146) */
int orc_infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist);

/* bnsGetSeq, util/BNTSeqUtil.scala:37-79 (== bns_get_seq, native/bntseq.c:355-376): the window [beg,end) of the 2-bit
 * reference in the doubled coordinate space; returns its length (0 when it bridges the strands) and writes at most
 * `cap` bases to out (a longer window returns -length and writes nothing). */
int64_t orc_bns_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, uint8_t *out, int64_t cap);

/*
 * memChainToAlnBatched, C2AB:380-616 (== mem_chain2aln per chain, native/bwamem.c:552-672), for a batch of reads in flat
 * SoA form.  The Scala walks "one seed per read per round"; per read that is exactly: chains in order, seeds of a chain
 * from the longest down (srt, C2AB:366-373), testExtension (C2AB:680-741) against every region the read has so far,
 * checkOverlapping (C2AB:753-787), extension() (C2AB:789-883) on windows cut from bnsGetSeq(rmax) (C2AB:344-364,
 * getMaxSpan C2AB:648-676), computeSeedCoverage (C2AB:891-907).
 *   read_len/read_off[n]   : reads (codes 0..4) in read_pool
 *   chain_cnt[n]           : chains per read;  seed_cnt[sum chain_cnt] : seeds per chain, (read, chain) order
 *   seed_rbeg/qbeg/len[]   : seeds in (read, chain, seedsRefArray) order
 *   out_cnt[n], out_regs[] : regions in creation order (regArrays before memSortAndDedup); returns total, or -needed
 */
int64_t orc_chain2aln_batch(const orc_opt_t *opt, int zdrop_mode, int64_t l_pac, const uint8_t *pac, int n_reads,
                            const int32_t *read_len, const int64_t *read_off, const uint8_t *read_pool,
                            const int32_t *chain_cnt, const int32_t *seed_cnt, const int64_t *seed_rbeg,
                            const int32_t *seed_qbeg, const int32_t *seed_len, int32_t *out_cnt, orc_alnreg_t *out_regs,
                            int64_t out_cap, int64_t *n_ext, int64_t *cells);

/*
 * Batched rescue (boundary 1), flat SoA form of the JNI call's arguments.
 *   seq_len/seq_off[2G]     : mate sequences (codes 0..4) in seq_pool, index 2k+i
 *   reg_cnt[2G], regs[]     : existing regions, concatenated in (k,i,j) order
 *   ref_cnt[2G]             : refSizeArray (PE:1944-1947)
 *   ref_rb/ref_re/ref_len/ref_off[4*R] : per (k,i,j<ref_cnt) x 4 orientations, concatenated; window
 *                             bytes in ref_pool at ref_off (ignored when len==0)
 *   out_cnt[2G], out_regs[] : regions after rescue, (k,i) order; returns total or -needed if cap too small
 * Follows native/bwamem_pair.c:115-228 (mode C) or PE:1335-1369 + PE:1111-1238 (mode Scala).
 * n_sw (optional) counts SWAlign2 calls; cells counts DP cells.
 */
int64_t orc_matesw_group(const orc_opt_t *opt, int64_t l_pac, const orc_pestat_t pes[4], int group_size,
                         const int32_t *seq_len, const int64_t *seq_off, const uint8_t *seq_pool,
                         const int32_t *reg_cnt, const orc_alnreg_t *regs, const int32_t *ref_cnt,
                         const int64_t *ref_rb, const int64_t *ref_re, const int64_t *ref_len,
                         const int64_t *ref_off, const uint8_t *ref_pool, int mode, int32_t *out_cnt,
                         orc_alnreg_t *out_regs, int64_t out_cap, int64_t *n_sw, int64_t *cells);

/* ---- worker2's tail (bpsw_oracle_tail.c): mark-primary, pairing, mapQ, memRegToAln, SAM text ------------------------ */
#define ORC_TAIL_SCALA 0 /* the Scala text: MemMarkPrimarySe.scala, MemSamPe.scala:462-572,1390-1612, MemRegToADAMSAM.scala */
#define ORC_TAIL_C 1     /* the BWA C it was transcribed from: native/bwamem.c, native/bwamem_pair.c, native/bwa.c */

typedef struct {
  float mask_level, mapq_coef_len; /* MemOptType.scala:47,51 */
  int32_t mapq_coef_fac;           /* :52 */
  int32_t pad_;
  char rg_id[64];                  /* SAMHeader.bwaReadGroupID ("" = none): \tRG:Z:<id> behind XS, MemRegToADAMSAM.scala:496-500 */
} orc_tail_opt_t;
void orc_tail_opt_default(orc_tail_opt_t *t);

typedef struct { /* datatype/MemAlnType.scala == mem_aln_t native/bwamem.h:71-80, CIGAR and MD kept beside it */
  int64_t pos;
  int32_t rid, flag, is_rev, mapq, NM, n_cigar, score, sub, md_len, status;
} orc_aln_t; /* 48 bytes */

void orc_mark_primary_se(const orc_opt_t *o, const orc_tail_opt_t *t, int n, orc_alnreg_t *a, int64_t id, int flavour);
int orc_approx_mapq_se(const orc_opt_t *o, const orc_tail_opt_t *t, const orc_alnreg_t *a, int flavour);
/* memPeStat, PE:117-260 (== mem_pestat, native/bwamem_pair.c:50-112): the insert-size statistics between worker1 and worker2 */
void orc_pe_stat(const orc_opt_t *o, const orc_tail_opt_t *t, int64_t l_pac, int n_pairs, const int32_t *reg_cnt,
                 const orc_alnreg_t *regs, int flavour, orc_pestat_t pes[4]);
int orc_mem_pair(const orc_opt_t *o, int64_t l_pac, const orc_pestat_t pes[4], int n0, const orc_alnreg_t *a0, int n1,
                 const orc_alnreg_t *a1, int64_t id, int flavour, int *sub, int *n_sub, int z[2]);
int orc_gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t *pac,
                   int l_query, const uint8_t *query_in, int64_t rb, int64_t re, int flavour, int *score, int *n_cigar, int *NM,
                   uint32_t *cigar, int cigar_cap, char *md, int md_cap, int *md_len);
int orc_bns_pos2rid(int n_seqs, const int64_t *ann_off, int64_t l_pac, int64_t pos_f);
void orc_reg2aln(const orc_opt_t *o, const orc_tail_opt_t *t, int n_seqs, const int64_t *ann_off, const int32_t *ann_len,
                 int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *query, const orc_alnreg_t *ar, int flavour,
                 orc_aln_t *a, uint32_t *cigar, int cigar_cap, char *md, int md_cap);
int64_t orc_sam_pe_batch(const orc_opt_t *o, const orc_tail_opt_t *t, int n_seqs, const int64_t *ann_off, const int32_t *ann_len,
                         const int64_t *ann_name_off, const char *ann_name_pool, int64_t l_pac, const uint8_t *pac,
                         const orc_pestat_t pes[4], int group_size, int64_t id0, const int32_t *read_len, const int64_t *read_off,
                         const uint8_t *read_pool, const char *qual_pool, const int64_t *name_off, const char *name_pool,
                         const int32_t *reg_cnt, orc_alnreg_t *regs, int flavour, char *out_text, int64_t cap, int64_t *out_off,
                         int64_t *n_reg2aln);

#ifdef __cplusplus
}
#endif
#endif
