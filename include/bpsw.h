/*
 * bpsw.h -- C ABI of libbPSW_hip.so, the MI355X (gfx950) batched Smith-Waterman plug-in for
 * CS-BWAMEM (ytchen0323/cloud-scale-bwamem).
 *
 * The library is a drop-in for the reference's two native plug-in points; the JNI symbols the
 * Scala driver binds are exported by the same .so (csrc/bpsw_jni.cpp) and are thin marshalling
 * shims over the functions declared here:
 *
 *   boundary 2  SWExtendFPGAJNI.swExtendFPGAJNI(n, bytes)          -> bpsw_extend_batch
 *               replaces src/main/jni_fpga/sw_extend_fpga.c:116-193 + src/main/alphadata/shm_host.c
 *               (declared at src/main/scala/cs/ucla/edu/bwaspark/jni/SWExtendFPGAJNI.scala:22,
 *                called at  .../worker1/MemChainToAlignBatched.scala:175-176)
 *   boundary 1  MateSWJNI.mateSWJNI(opt,pacLen,pes,n,seqs,regs,refs,refSizes) -> bpsw_matesw_group
 *               replaces src/main/native/jni_mate_sw.c:58-662 + native/bwamem_pair.c:115-228
 *               (declared at .../jni/MateSWJNI.scala:24-25, called at .../worker2/MemSamPe.scala:2091-2092)
 *
 * All pointers are plain host (or, for the *_device entry points, device) pointers with explicit
 * sizes; no C++/torch types cross this boundary.  Every function returns BPSW_OK (0) or a negative
 * error code; bpsw_last_error() gives the text.  There is NO CPU fallback: if no HIP device is
 * usable the calls fail with BPSW_ERR_DEVICE.
 *
 * Thread safety: a bpsw_ctx_t serialises its own calls with an internal mutex; use one context
 * per host thread (the JNI shim keeps one per thread per device) for concurrency.
 */
#ifndef BPSW_H
#define BPSW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPSW_OK 0
#define BPSW_ERR_ARG (-1)      /* malformed argument / wire batch */
#define BPSW_ERR_DEVICE (-2)   /* HIP runtime failure or no gfx950 device */
#define BPSW_ERR_CAPACITY (-3) /* caller's output buffer too small */
#define BPSW_ERR_LIMIT (-4)    /* sequence longer than the kernels support */

#define BPSW_ZDROP_SCALA 0 /* SWUtil.scala:194-199 (default: bit-exact vs the Scala extension()) */
#define BPSW_ZDROP_BWA 1   /* native/ksw.c:455-461 */

#define BPSW_RESCUE_C 0     /* bookkeeping of native/bwamem_pair.c + bwamem.c (what -bPSWJNI 1 does today) */
#define BPSW_RESCUE_SCALA 1 /* bookkeeping of MemSamPe.scala:1111-1238 (what -bPSWJNI 0 does) */

#define BPSW_KSW_XBYTE 0x10000
#define BPSW_KSW_XSTOP 0x20000
#define BPSW_KSW_XSUBO 0x40000
#define BPSW_KSW_XSTART 0x80000

/* kernel limits (checked on the host before every launch) */
#define BPSW_EXT_MAX_QLEN 1023  /* per-side query length of an extension task */
#define BPSW_EXT_MAX_RLEN 4095  /* per-side reference length of an extension task */
#define BPSW_SW_MAX_QLEN 512    /* mate length of a rescue job */
#define BPSW_SW_MAX_TLEN 65535  /* window length of a rescue job */

typedef struct bpsw_ctx bpsw_ctx_t;

/* ---- life cycle ------------------------------------------------------------------------- */
int bpsw_device_count(void);
/* device < 0: pick from BPSW_DEVICES / round robin (INTEGRATION.md "device selection").
 * Side effects on the PROCESS, both documented in INTEGRATION.md: the first bpsw_create on a device asks the HIP runtime for
 * interrupt-driven waits on it (hipSetDeviceFlags(hipDeviceScheduleBlockingSync); BPSW_SPIN_WAIT=1 leaves the runtime's default),
 * and the device phases of all contexts of a device share a pool of min(BPSW_STREAM_POOL = 20, GPU_MAX_HW_QUEUES) streams --
 * GPU_MAX_HW_QUEUES must be in the executor's environment before the runtime initialises (unset: 4 queues, one warning on stderr). */
int bpsw_create(int device, bpsw_ctx_t **out);
void bpsw_destroy(bpsw_ctx_t *ctx);
int bpsw_device_of(const bpsw_ctx_t *ctx);
/* Spark partition -> device (the north_star's "Spark-partition -> device index"; reference: one accelerator per executor,
 * src/main/jni_fpga/sw_extend_fpga.c:116-193).  The devices contexts are spread over are the entries of BPSW_DEVICES
 * ("0,2,3"; an index may repeat; default: every visible device in order); partition p runs on entry p mod count. */
int bpsw_device_slots(void);                  /* number of entries; 0 = no usable device */
int bpsw_device_for_partition(int partition); /* HIP device index of that entry, -1 = no usable device / negative partition */
const char *bpsw_last_error(void); /* thread-local text of the last failing call */
const char *bpsw_version(void); /* "bPSW-hip <major.minor> (gfx950)": structs of this header only ever grow at their end, and the minor
                                   number changes when one does (0.4: bpsw_stats_t::ext_full_relaunches, bpsw_tail_opt_t::rg_id; 0.5: bpsw_stats_t::sw_ring_calls, ext_ring_calls) */

/* ---- scoring that boundary 2 does not transmit (zdrop, mat) ------------------ */
/* defaults: MemOptType (datatype/MemOptType.scala:28-73): a=1 b=4 N=-1, zdrop=100, Scala z-drop parse */
int bpsw_set_ext_scoring(bpsw_ctx_t *ctx, const int8_t mat[25], int zdrop, int zdrop_mode);

/* Which of the kernel's EXACT shortcuts may replace the DP of an extension flank (results are identical either
 * way, this is an A/B and test switch): bit 0 closed form for near-exact flanks, bit 1 single-gap certificate, bit 2 its
 * two-gap-open extension, bit 3 one-base gap at the start of a flank, bit 4 tail-row bound, bit 5 evaluate those forms in the
 * one-task-per-lane sift kernel in front of the extension kernel (where, not which).  mask < 0 or 63: all (default).
 * Applies to bpsw_extend_batch* and bpsw_chain2aln_batch on this context. */
int bpsw_set_ext_shortcuts(bpsw_ctx_t *ctx, int mask);

/* ---- boundary 2: batched seed extension -------------------------------------------------- */
/*
 * wire = header(32 B) | task table (32 B x n) | nibble-packed sequences, exactly the byte[] built
 * by runOnFPGAJNI (MemChainToAlignBatched.scala:76-172).  out receives 10 int16 per task
 * (MemChainToAlignBatched.scala:181-188): idx lo, idx hi, qBeg, qEnd, rBeg, rEnd, score,
 * trueScore, width, 0.  out_len is the capacity of out in int16 units (>= 10*n).
 */
int bpsw_extend_batch(bpsw_ctx_t *ctx, const uint8_t *wire, size_t wire_bytes, int16_t *out, size_t out_len);

/* Single-touch form of bpsw_extend_batch for callers that can fill a buffer themselves (the JNI shim: GetByteArrayRegion of
 * swExtendFPGAJNI's byte[] straight into pinned memory, where the reference does one memcpy into its shared-memory segment,
 * src/main/jni_fpga/sw_extend_fpga.c:146-155).  bpsw_extend_stage returns the context's pinned staging block with room for
 * `bytes`; the caller writes the wire batch there and calls bpsw_extend_commit, which runs the batch without copying it again
 * and returns a VIEW of the 10*n int16 results where the kernel wrote them (*out, *out_len int16; valid until the next call on
 * the context; n == 0 gives *out == NULL).  One commit per stage, of at most the `bytes` that were staged (BPSW_ERR_ARG otherwise: the
 * table scan never reads past the pinned block).  One thread per context, as everywhere. */
int bpsw_extend_stage(bpsw_ctx_t *ctx, size_t bytes, uint8_t **buf);
int bpsw_extend_commit(bpsw_ctx_t *ctx, size_t wire_bytes, const int16_t **out, size_t *out_len);

/* Diagnostics: bpsw_extend_batch that also reports, per task and side (side_how[2 t] left, [2 t + 1] right; 2 n bytes), how the
 * result was produced: 0 = the side is empty, 1 = an exact shortcut (bpsw_set_ext_shortcuts), 2 = the DP was swept.  bench.py
 * uses it to split the useful cell updates per second into "DP run" and "closed form". */
int bpsw_extend_batch_classify(bpsw_ctx_t *ctx, const uint8_t *wire, size_t wire_bytes, int16_t *out, size_t out_len, uint8_t *side_how);

/* Same computation with the wire batch and the result already resident in device memory
 * (used by bench.py; d_wire must be 16-byte aligned).  hip_stream is a hipStream_t or NULL for the context's stream.
 * ASYNCHRONOUS: the device-side table scan, the main launch and the scan's read-back are enqueued back to back and the call
 * returns.  The launch is sized for a fixed geometry (sides <= 256 bases) and checks the scan on the device: a malformed batch,
 * or one with longer sides, is left untouched.  The NEXT call on this context, or bpsw_last_kernel_ms, waits for the launch and
 * (a) returns the deferred BPSW_ERR_ARG / BPSW_ERR_LIMIT of a malformed batch, (b) launches a longer-sided batch again with its
 * real geometry.  d_wire / d_out must stay valid until then.  bpsw_swalign2_batch_device behaves the same way, sizing the
 * launch for the geometry of the previous (verified) call on the context; the first call of a context is synchronous. */
int bpsw_extend_batch_device(bpsw_ctx_t *ctx, const void *d_wire, size_t wire_bytes, int n_tasks,
                             void *d_out, void *hip_stream);

/* Host-side mirror of the Scala packer (MemChainToAlignBatched.scala:76-172) for non-JVM callers.
 * SoA task description; sequences are byte-per-base codes 0..4 in `pool` (left_* already reversed). */
typedef struct {
  int32_t n;
  int32_t o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, w; /* header fields */
  int32_t mat_max;                                             /* max(mat), feeds the maxIns/maxDel shorts */
  const int32_t *left_qlen, *left_rlen, *right_qlen, *right_rlen;
  const int64_t *left_q_off, *left_r_off, *right_q_off, *right_r_off; /* into pool */
  const int32_t *reg_score, *q_beg, *h0, *idx;
  const uint8_t *pool;
} bpsw_ext_tasks_t;
size_t bpsw_wire_size(const bpsw_ext_tasks_t *t);
int bpsw_wire_pack(const bpsw_ext_tasks_t *t, uint8_t *buf, size_t cap, size_t *bytes);

/* Coordinate batches ("wire format 2"): the same call -- and the same JNI symbol, swExtendFPGAJNI(n, bytes) --
 * accepts a batch that names the target flanks by reference coordinates instead of shipping their bases, when the reference
 * is on the device (bpsw_ref_load).  It replaces bnsGetSeq + the leftRs/rightRs copies + their nibble packing
 * (MemChainToAlignBatched.scala:363, 511-517, 534-541, 143-161) on the caller's side and about a third of the bytes per task.
 *   header      as format 1, with byte 7 = BPSW_WIRE_COORDS (format 1 leaves it 0)
 *   task record 40 bytes: the 32 bytes of format 1 (leftQlen, leftRlen, rightQlen, rightRlen int16; word offset of the task's
 *               nibbles; regScore, qBeg, h0 int16; maxIns/maxDel shorts; idx int32) with the seed length (int16) in the slot of
 *               the redundant 16-bit idx (bytes 18-19), then the seed's rBeg in [0, 2*l_pac) as int64 (bytes 32-39)
 *   nibbles     leftQs then rightQs only (leftQs reversed, as in format 1)
 * The left target flank is base(rBeg - 1 - i), i < leftRlen; the right one base(rBeg + len + i), i < rightRlen, with base() the
 * doubled-strand lookup of bnsGetSeq (util/BNTSeqUtil.scala:56-73); a flank must not leave [0, 2*l_pac) nor bridge the strands
 * (getMaxSpan, MemChainToAlignBatched.scala:654-677, guarantees both).  Results are identical to format 1 on the same tasks. */
#define BPSW_WIRE_COORDS 2
typedef struct {
  int32_t n;
  int32_t o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, w; /* header fields */
  int32_t mat_max;
  const int32_t *left_qlen, *left_rlen, *right_qlen, *right_rlen;
  const int64_t *left_q_off, *right_q_off; /* into pool; left flank already reversed */
  const int32_t *reg_score, *q_beg, *h0, *idx, *seed_len;
  const int64_t *seed_rbeg;
  const uint8_t *pool;
} bpsw_ext_coord_tasks_t;
size_t bpsw_wire_coords_size(const bpsw_ext_coord_tasks_t *t);
int bpsw_wire_coords_pack(const bpsw_ext_coord_tasks_t *t, uint8_t *buf, size_t cap, size_t *bytes);

/* ---- boundary 1: pair-end mate-SW rescue -------------------------------------------------- */
typedef struct { /* == mem_alnreg_t native/bwamem.h:49-61 == MemAlnRegType.scala:26-38 */
  int64_t rb, re;
  int32_t qb, qe, score, truesc, sub, csub, sub_n, w, seedcov, secondary;
  uint64_t hash;
} bpsw_alnreg_t; /* 64 bytes */

typedef struct { /* == mem_pestat_t native/bwamem.h:65-69 == MemPeStat.scala:27-31 */
  int32_t low, high, failed, pad_;
  double avg, std;
} bpsw_pestat_t;

typedef struct { /* the MemOptType fields the rescue reads (native/jni_mate_sw.c:177-221) */
  int32_t a, b, o_del, e_del, o_ins, e_ins, pen_unpaired, pen_clip5, pen_clip3, w, zdrop, T, flag,
      min_seed_len, max_ins, max_matesw;
  float mask_level_redun;
  int8_t mat[25];
  int8_t pad_[3];
} bpsw_opt_t;
void bpsw_opt_default(bpsw_opt_t *opt);

/*
 * Raw local-SW jobs (SWUtil.SWAlign2, SWUtil.scala:583-601), one per (anchor, orientation).
 * Job t aligns query bytes q_pool[q_off[t] .. +q_len[t]) -- reverse-complemented on the fly when
 * q_rev[t] != 0 (MemSamPe.scala:1175-1184) -- against t_pool[t_off[t] .. +t_len[t]).
 * out receives 7 int32 per job: score, tEnd, qEnd, score2, tEnd2, tBeg, qBeg.
 */
typedef struct {
  int32_t n;
  int32_t xtra; /* KSW_X* flags | min score, same for all jobs (MemSamPe.scala:1187-1189) */
  const int32_t *q_len, *t_len;
  const int64_t *q_off, *t_off;
  const uint8_t *q_rev;
  const uint8_t *q_pool, *t_pool;
  size_t q_pool_bytes, t_pool_bytes;
} bpsw_sw_jobs_t;
int bpsw_swalign2_batch(bpsw_ctx_t *ctx, const bpsw_opt_t *opt, const bpsw_sw_jobs_t *jobs, int32_t *out);

/* Device-resident form: every array pointer in `jobs` is a device pointer (pools 16-byte aligned). */
int bpsw_swalign2_batch_device(bpsw_ctx_t *ctx, const bpsw_opt_t *opt, const bpsw_sw_jobs_t *jobs,
                               void *d_out, void *hip_stream);

/*
 * The whole boundary-1 call in flat SoA form (what the JNI shim builds from the object arrays):
 *   seq_len/seq_off[2G]   mate sequences (codes 0..4) in seq_pool, index 2k+i
 *   reg_cnt[2G], regs     existing regions concatenated in (k,i,j) order (MemSamPe.scala:1963-1990)
 *   ref_cnt[2G]           refSizeArray (MemSamPe.scala:1944-1947)
 *   ref_rb/re/len/off[4R] per (k,i,j<ref_cnt) x 4 orientations; window bytes at ref_pool+ref_off
 *   out_cnt[2G], out_regs regions after rescue in (k,i,rank) order; *out_total = sum(out_cnt)
 * Returns BPSW_ERR_CAPACITY (with *out_total set to the needed size) if out_cap is too small.
 */
typedef struct {
  int32_t group_size;
  int64_t l_pac;
  bpsw_pestat_t pes[4];
  const int32_t *seq_len;
  const int64_t *seq_off;
  const uint8_t *seq_pool;
  size_t seq_pool_bytes;
  const int32_t *reg_cnt;
  const bpsw_alnreg_t *regs;
  const int32_t *ref_cnt;
  const int64_t *ref_rb, *ref_re, *ref_len, *ref_off;
  const uint8_t *ref_pool;
  size_t ref_pool_bytes;
} bpsw_rescue_group_t;
int bpsw_matesw_group(bpsw_ctx_t *ctx, const bpsw_opt_t *opt, const bpsw_rescue_group_t *g, int mode,
                      int32_t *out_cnt, bpsw_alnreg_t *out_regs, int64_t out_cap, int64_t *out_total);

/* ---- "next" row: banded global alignment -> score + CIGAR ------------------------------ */
/*
 * SWUtil.SWGlobal (SWUtil.scala:233-397 == ksw_global2, native/ksw.c:501-584), the DP behind bwaGenCigar2
 * (MemRegToADAMSAM.scala:738-891).  It is NOT behind either JNI of the reference (the Scala driver calls it
 * directly), so this is an additional export for callers that want CIGARs from the device.
 * Job t aligns q_pool[q_off[t]..+q_len[t]) to t_pool[t_off[t]..+t_len[t]) globally inside the band w[t]
 * (the caller applies the band rule of MemRegToADAMSAM.scala:794-804 and the strand reversal of :764-781).
 * out_score[t]; out_ncigar[t] = number of operations; out_cigar[t*max_cigar ..] = len<<4|op (0=M 1=I 2=D).
 * If out_ncigar[t] > max_cigar that job's operations were not written: resubmit it with a larger max_cigar (<= 512).
 */
#define BPSW_GLOBAL_MAX_QLEN 1023
#define BPSW_GLOBAL_MAX_TLEN 65535
#define BPSW_GLOBAL_MAX_CIGAR 512
typedef struct {
  int32_t n;
  int32_t max_cigar;
  const int32_t *q_len, *t_len, *w;
  const int64_t *q_off, *t_off;
  const uint8_t *q_pool, *t_pool;
  size_t q_pool_bytes, t_pool_bytes;
} bpsw_global_jobs_t;
int bpsw_global_batch(bpsw_ctx_t *ctx, const bpsw_opt_t *opt, const bpsw_global_jobs_t *jobs, int32_t *out_score,
                      int32_t *out_ncigar, uint32_t *out_cigar);

/* ---- "next" row: reference-window extraction on the device ------------------------------ */
/*
 * bnsGetSeq (util/BNTSeqUtil.scala:37-79 == bns_get_seq, native/bntseq.c) with the 2-bit .pac resident in HBM
 * (base k = pac[k>>2] >> ((~k & 3) << 1) & 3; coordinates >= l_pac address the reverse-complement strand).  The
 * reference belongs to the DEVICE of the context it was loaded through and is seen by every context of that device
 * (the JNI shim keeps one context per Spark task thread); loading or unloading must not race with calls in flight on
 * that device.  Once loaded, the callers of bnsGetSeq on the hot path (MemSamPe.scala:1852 for the rescue windows)
 * can send only (rBeg, rEnd) instead of the bytes:
 *   bpsw_sw_jobs_t      with t_pool == NULL : t_off[t] is the window start in the doubled coordinate space,
 *                                             t_len[t] its length; a window may not bridge l_pac.
 *   bpsw_rescue_group_t with ref_pool == NULL: ref_rb/ref_re name the windows (-1,-1 = failed orientation,
 *                                             MemSamPe.scala:1863-1868); ref_len/ref_off are ignored, the length is
 *                                             derived with bnsGetSeq's own rules (clamp to [0, 2*l_pac), 0 when the
 *                                             window bridges the strands) and g->l_pac must equal the loaded length.
 * bpsw_ref_fetch is bnsGetSeq itself (n windows -> bytes), for callers that still want the bases on the host and
 * for the parity tests: out_len[t] = window length after the reference's swap/clamp (0 when bridging); the bases go
 * to out_pool[out_off[t] ..]; returns BPSW_ERR_CAPACITY if a window does not fit before out_pool_bytes.
 */
int bpsw_ref_load(bpsw_ctx_t *ctx, const uint8_t *pac, int64_t l_pac); /* copies (l_pac+3)/4 bytes to the device */
int bpsw_ref_unload(bpsw_ctx_t *ctx);
int64_t bpsw_ref_length(const bpsw_ctx_t *ctx); /* l_pac of the loaded reference, 0 if none */
int bpsw_ref_fetch(bpsw_ctx_t *ctx, int32_t n, const int64_t *beg, const int64_t *end, uint8_t *out_pool,
                   size_t out_pool_bytes, const int64_t *out_off, int64_t *out_len);

/* ---- "next" row: the memChainToAlnBatched round loop on the device ----------------------------- */
/*
 * memChainToAlnBatched (worker1/MemChainToAlignBatched.scala:380-616; per chain == mem_chain2aln, native/bwamem.c:552-672):
 * the caller hands over the reads of a batch with their filtered seed chains and gets back, per read, the regions the
 * Scala leaves in regArrays -- all rounds (testExtension, checkOverlapping, extension with band retries, seed coverage)
 * run on the device, one wavefront per read, with the reference windows read from the 2-bit reference loaded by
 * bpsw_ref_load.  One call per batch replaces one swExtendFPGAJNI call per round plus the Scala-side task packing.
 *   read_len/read_off[n]  : reads (codes 0..4, <= 256 bases) in read_pool
 *   chain_cnt[n]          : chains per read (0 == chainsFilteredArray(i) null)
 *   seed_cnt[sum chains]  : seeds per chain, (read, chain) order
 *   seed_rbeg/qbeg/len[]  : MemSeedType fields in (read, chain, seedsRefArray) order; a chain lies on one strand
 *   out_cnt[n], out_regs  : regions per read in creation order (sub, csub, sub_n, secondary, hash = 0 as in the Scala);
 *                           out_cap must be >= the total number of seeds.  With BPSW_C2A_SORT_DEDUP the lists are
 *                           passed through memSortAndDedup (what bwaMemWorker1Batched returns,
 *                           BWAMemWorker1Batched.scala:128-133), C flavour by default, Scala flavour with
 *                           BPSW_C2A_DEDUP_SCALA.
 * zdrop_mode selects the z-drop parse as in bpsw_set_ext_scoring.  Needs e_del, e_ins >= 1 and 1 <= w <= 254.
 */
#define BPSW_C2A_SORT_DEDUP 1
#define BPSW_C2A_DEDUP_SCALA 2
typedef struct {
  int32_t n_reads;
  const int32_t *read_len;
  const int64_t *read_off;
  const uint8_t *read_pool;
  size_t read_pool_bytes;
  const int32_t *chain_cnt;
  const int32_t *seed_cnt;
  const int64_t *seed_rbeg;
  const int32_t *seed_qbeg, *seed_len;
} bpsw_chains_t;
int bpsw_chain2aln_batch(bpsw_ctx_t *ctx, const bpsw_opt_t *opt, const bpsw_chains_t *batch, int zdrop_mode, int flags,
                         int32_t *out_cnt, bpsw_alnreg_t *out_regs, int64_t out_cap, int64_t *out_total);

/* ---- worker2's tail: everything after the rescue ------------------------------------------
 *
 * memSamPeGroupRest (worker2/MemSamPe.scala:1390-1612 == mem_sam_pe after the rescue, native/bwamem_pair.c:385-452):
 * memMarkPrimarySe, memPair, the mapQ arithmetic, memRegToAln (band inference, up to three banded global alignments,
 * NM/MD, position, clipping) and the SAM text (memAlnToSAM, worker2/MemRegToADAMSAM.scala:328-560).  Not behind either
 * JNI of the reference (the Scala calls them directly), so these are additional exports like bpsw_global_batch.
 * The global alignments, NM/MD and the coordinates run on the GPU against the reference loaded with bpsw_ref_load
 * (bpsw_reg2aln.hip); the per-pair bookkeeping and the text are host code (bpsw_tail.cpp).
 *
 * flavour: where the Scala text and the C it was transcribed from differ (band rule of bwaGenCigar2,
 * mapQ at len == mapQCoefLen, hash ordering and parent index in memMarkPrimarySe, flag folding in memAlnToSAM) the
 * Scala form is BPSW_TAIL_SCALA (default) and the C form BPSW_TAIL_C.
 */
#define BPSW_TAIL_SCALA 0
#define BPSW_TAIL_C 1
#define BPSW_MEM_F_NOPAIRING 0x4 /* bits of bpsw_opt_t.flag the tail reads (native/bwamem.h:14-19) */
#define BPSW_MEM_F_ALL 0x8
#define BPSW_MEM_F_NO_MULTI 0x10
#define BPSW_R2A_MAX_QLEN 1024 /* read length */
#define BPSW_R2A_MAX_RLEN 4096 /* re - rb of a region */

typedef struct { /* the MemOptType fields only the tail reads (datatype/MemOptType.scala:47-52) */
  float mask_level, mapq_coef_len;
  int32_t mapq_coef_fac;
  int32_t flavour;
  char rg_id[64]; /* read-group ID ("" = none): every SAM line gets \tRG:Z:<id> behind XS (worker2/MemRegToADAMSAM.scala:496-500 ==
                     native/bwamem.c:815; the ID is what SAMHeader.bwaSetReadGroup / bwa_set_rg cut out of the -R line) */
} bpsw_tail_opt_t;
void bpsw_tail_opt_default(bpsw_tail_opt_t *t);

/* The contig table of the reference loaded with bpsw_ref_load (bntann1_t offset / len / name, native/bntseq.h:40-46 ==
 * datatype/BNTSeqType.scala); shared by every context of the device like the reference itself.  names: n_seqs
 * NUL-terminated strings back to back, or NULL (then the SAM text uses "ctgN"). */
int bpsw_bns_load(bpsw_ctx_t *ctx, int32_t n_seqs, const int64_t *offset, const int32_t *len, const char *names);

#define BPSW_ALN_OK 0
#define BPSW_ALN_XREF 1     /* bwaFixXref2 could not repair the hit (the Scala asserts, R2S:196-199; the C exits) */
#define BPSW_ALN_NOCIGAR 2  /* bwaGenCigar2 returned null (window out of range / bridging the strands) */
#define BPSW_ALN_OVERFLOW 3 /* more CIGAR operations / MD bytes than the kernel stages: not produced */
typedef struct { /* MemAlnType (datatype/MemAlnType.scala) == mem_aln_t (native/bwamem.h:71-80); CIGAR and MD beside it */
  int64_t pos;
  int32_t rid, flag, is_rev, mapq, NM, n_cigar, score, sub, md_len, status;
} bpsw_aln_t; /* 48 bytes */

/* memRegToAln (worker2/MemRegToADAMSAM.scala:172-313 == mem_reg2aln, native/bwamem.c:949-1021) for n (read, region) jobs.
 * regs[j].rb < 0 or .re < 0: the unmapped record.  out_cigar: max_cigar words per job (len<<4|op, op MIDS=0123);
 * out_md: max_md bytes per job (text, no NUL).  A job whose n_cigar > max_cigar or md_len > max_md has to be resubmitted
 * with more room (nothing is truncated silently: its status is BPSW_ALN_OVERFLOW only past the kernel's own staging). */
typedef struct {
  int32_t n;
  int32_t max_cigar, max_md;
  const int32_t *read_len;
  const int64_t *read_off;
  const uint8_t *read_pool; /* codes 0..4 */
  size_t read_pool_bytes;
  const bpsw_alnreg_t *regs;
} bpsw_reg2aln_jobs_t;
int bpsw_reg2aln_batch(bpsw_ctx_t *ctx, const bpsw_opt_t *opt, const bpsw_tail_opt_t *topt, const bpsw_reg2aln_jobs_t *jobs,
                       bpsw_aln_t *out, uint32_t *out_cigar, uint8_t *out_md);

/* The tail over a group of pairs.  Arrays are indexed 2k+i (pair k, end i); regions in (k, i, j) order as
 * bpsw_matesw_group leaves them.  id0: pair id of the first pair (the reference hashes id0 + k).
 * Output: out_text receives the SAM lines (read 2k+i: out_text[out_off[2k+i] .. out_off[2k+i+1]); a read with
 * supplementary hits has several lines); BPSW_ERR_CAPACITY with *out_needed set when text_cap is too small (out_text then
 * holds a part of the text, nothing is written past text_cap, out_off is complete: call again with *out_needed bytes).
 * out_regs (optional, same size as regs): the region lists as the tail leaves them (sorted, sub/sub_n/secondary/hash). */
typedef struct {
  int32_t group_size;
  int64_t id0;
  bpsw_pestat_t pes[4];
  const int32_t *read_len;
  const int64_t *read_off;
  const uint8_t *read_pool;
  const uint8_t *qual_pool; /* same offsets as read_pool, or NULL */
  size_t read_pool_bytes;
  const int64_t *name_off;  /* group_size + 1 offsets into name_pool */
  const char *name_pool;
  const int32_t *reg_cnt;
  const bpsw_alnreg_t *regs;
} bpsw_pairs_t;
int bpsw_sam_pe_batch(bpsw_ctx_t *ctx, const bpsw_opt_t *opt, const bpsw_tail_opt_t *topt, const bpsw_pairs_t *g, char *out_text,
                      size_t text_cap, int64_t *out_off, size_t *out_needed, bpsw_alnreg_t *out_regs);
/* Host-only pieces of the tail and of the rescue bookkeeping, exported so that they can be used and tested without a device
 * (no context, no GPU): memMarkPrimarySe (sorts regs in place and fills sub / sub_n / secondary / hash), memApproxMapqSe,
 * memPair (out5 = {score, sub, n_sub, z0, z1}) and memSortAndDedup (mode BPSW_RESCUE_C / _SCALA; returns the new count). */
int bpsw_mark_primary_se(const bpsw_opt_t *opt, const bpsw_tail_opt_t *topt, int32_t n, bpsw_alnreg_t *regs, int64_t id);
int bpsw_approx_mapq_se(const bpsw_opt_t *opt, const bpsw_tail_opt_t *topt, const bpsw_alnreg_t *reg);
int bpsw_mem_pair(const bpsw_opt_t *opt, const bpsw_tail_opt_t *topt, int64_t l_pac, const bpsw_pestat_t pes[4], int32_t n0,
                  const bpsw_alnreg_t *regs0, int32_t n1, const bpsw_alnreg_t *regs1, int64_t id, int32_t out5[5]);
int bpsw_sort_dedup(int32_t n, bpsw_alnreg_t *regs, float mask_level_redun, int mode);
/* HARNESS ONLY -- outside both boundaries (the driver's job), kept because the tests and tools that build worker2
 * inputs need the pes[] the driver would have computed; a maintainer wiring the library into the aligner does not bind it.
 * memPeStat (worker2/MemSamPe.scala:117-260 == mem_pestat, native/bwamem_pair.c:50-112): the insert-size statistics the driver
 * computes between worker1 and worker2, from the region lists of n_pairs pairs in (pair, end, j) order.  Nothing is printed. */
int bpsw_pe_stat(const bpsw_opt_t *opt, const bpsw_tail_opt_t *topt, int64_t l_pac, int32_t n_pairs, const int32_t *reg_cnt,
                 const bpsw_alnreg_t *regs, bpsw_pestat_t pes[4]);

/* worker2 in one call: the rescue of boundary 1 (anchors and windows as memSamPeGroupJNIPrepare builds them,
 * worker2/MemSamPe.scala:1895-2000, windows named by coordinates of the reference loaded with bpsw_ref_load; rescue_mode as in
 * bpsw_matesw_group) followed by the tail above.  g->regs are the region lists BEFORE the rescue (what worker1 hands over).
 * out_reg_cnt (2*group_size, optional) / out_regs (optional): the lists after the rescue and the tail's bookkeeping. */
int bpsw_worker2_batch(bpsw_ctx_t *ctx, const bpsw_opt_t *opt, const bpsw_tail_opt_t *topt, const bpsw_pairs_t *g, int rescue_mode,
                       char *out_text, size_t text_cap, int64_t *out_off, size_t *out_needed, int32_t *out_reg_cnt,
                       bpsw_alnreg_t *out_regs, int64_t out_regs_cap, int64_t *out_regs_total);
/* The tail off the calling thread: a library-owned pool of n_workers tail workers on one device (each with a context of its own).
 * The reference runs memSamPeGroupRest in the partition's task thread (worker2/MemSamPe.scala:1390-1612 under FastMap.scala:266-293);
 * with the pool that thread only ENQUEUES its groups and collects the text later -- plan, kernel and text of different groups overlap.
 * bpsw_tail_pool_submit copies opt, topt and *g by value and returns a ticket at once; everything they and the out_* arguments point
 * to stays the caller's and must stay valid and unchanged until the ticket is collected.  rescue_mode: BPSW_RESCUE_C / BPSW_RESCUE_SCALA
 * = bpsw_worker2_batch (rescue + tail), BPSW_TAIL_POOL_TAIL_ONLY = bpsw_sam_pe_batch (out_reg_cnt / out_regs_cap unused; out_regs as
 * there).  bpsw_tail_pool_wait blocks until the ticket's group is done and returns ITS return code (BPSW_ERR_CAPACITY with *out_needed /
 * *out_regs_total set, exactly as the direct calls; the worker's error text becomes the collecting thread's bpsw_last_error());
 * tickets may be collected in any order, each once.  bpsw_tail_pool_destroy runs what is still queued, then joins the workers. */
typedef struct bpsw_tail_pool bpsw_tail_pool_t;
#define BPSW_TAIL_POOL_TAIL_ONLY (-1)
int bpsw_tail_pool_create(int device, int n_workers, bpsw_tail_pool_t **out);
void bpsw_tail_pool_destroy(bpsw_tail_pool_t *pool);
int bpsw_tail_pool_submit(bpsw_tail_pool_t *pool, const bpsw_opt_t *opt, const bpsw_tail_opt_t *topt, const bpsw_pairs_t *g,
                          int rescue_mode, char *out_text, size_t text_cap, int64_t *out_off, int32_t *out_reg_cnt,
                          bpsw_alnreg_t *out_regs, int64_t out_regs_cap, int64_t *ticket);
int bpsw_tail_pool_wait(bpsw_tail_pool_t *pool, int64_t ticket, size_t *out_needed, int64_t *out_regs_total);
int bpsw_tail_pool_workers(const bpsw_tail_pool_t *pool);
/* the most recent tail call on this context: kernel_ms = reg2aln_kernel launches (hipEvents on the launch stream), n_jobs =
 * jobs they carried, host_ms = {plan, device round trip (staging, copies, kernel), emit} of bpsw_sam_pe_batch */
int bpsw_last_tail_times(bpsw_ctx_t *ctx, float *kernel_ms, int32_t *n_jobs, double host_ms[3]);

/* ---- statistics (the buckets of profiling/SWBatchTimeBreakdown.scala:25-39, device flavoured) -- */
typedef struct {
  uint64_t ext_calls, ext_tasks, ext_wire_bytes;
  uint64_t sw_calls, sw_jobs, sw_speculated, sw_replayed_rounds, sw_wasted;
  double ext_h2d_ms, ext_kernel_ms, ext_d2h_ms;
  double sw_h2d_ms, sw_kernel_ms, sw_d2h_ms, sw_host_ms;
  /* wall-clock phases of the host-buffer entry points, summed over calls (ms): staging copy in, waiting for a device
   * stream of the pool, device phase (H2D + kernel + D2H until the stream is idle), staging copy out; for the group
   * rescue: speculation, job packing, waiting, device phase, replay, output copy */
  double ext_host_in_ms, ext_wait_ms, ext_dev_ms, ext_host_out_ms;
  double grp_plan_ms, grp_pack_ms, grp_wait_ms, grp_dev_ms, grp_replay_ms, grp_out_ms;
  uint64_t grp_calls, grp_pairs;
  uint64_t ext_full_relaunches; /* extension calls whose short kernel deferred tasks, so that the full kernel was launched behind it after all */
  uint64_t sw_ring_calls;       /* SW batches that went through the device's submission ring (resident kernel) instead of a launch of their own;
                                   for those sw_kernel_ms is the batch's span on the device clock: first job pair taken -> last one finished */
  uint64_t ext_ring_calls;      /* extension batches (small ones: no long flank, too few tasks for the sift kernel) that went through the
                                   device's extension ring; ext_kernel_ms likewise: first task taken -> last one finished */
} bpsw_stats_t;
int bpsw_get_stats(bpsw_ctx_t *ctx, bpsw_stats_t *out);
int bpsw_reset_stats(bpsw_ctx_t *ctx);
/* duration in ms of the most recent extend / swalign kernel launch on this context, measured with
 * hipEvents on the launch stream.  Synchronises the stream and resolves the asynchronous device entries: their deferred
 * errors are returned here. */
int bpsw_last_kernel_ms(bpsw_ctx_t *ctx, float *ext_ms, float *sw_ms);
/* The submission ring of the context's device (round 5; csrc/bpsw_ring.h): the SW batches of bpsw_swalign2_batch / bpsw_matesw_group /
 * mateSWJNI of ALL contexts of a device are descriptors of one resident kernel instead of launches of their own.  epochs = launches of
 * that kernel (it ends by itself BPSW_RING_IDLE_US after its last batch), submitted = batches appended, carried = batches a closing
 * epoch handed to its successor; epochs_ms / epochs_timed = summed duration and number of the epochs that are OVER, each timed by two HIP
 * events around the resident kernel's launch on its stream (what a kernel trace reports as that kernel's duration).  Diagnostics; any
 * pointer may be null. */
int bpsw_ring_stats(bpsw_ctx_t *ctx, uint64_t *epochs, uint64_t *submitted, uint64_t *carried, double *epochs_ms, uint64_t *epochs_timed);
/* The rings' integrity tripwire (csrc/bpsw_ring.cpp).  A batch's results and its completion word reach host memory as separate writes
 * from many wavefronts; before a caller publishes a batch it poisons every record of its result block with a value no kernel writes, and
 * after the completion word it looks at every record again.  checked = records looked at, faults = records that still held the poison when
 * the completion word was visible (process-wide; expected: 0 -- a fault is reported on stderr once, the call waits up to 5 ms for the
 * records and fails with BPSW_ERR_DEVICE if they do not arrive; an extension batch is run again through a launch instead).  Returns 1
 * when the check is on (default), 0 when BPSW_RING_INTEGRITY=0 switched it off.  Either pointer may be null. */
int bpsw_ring_integrity(uint64_t *checked, uint64_t *faults);
/* A gauge: the number of SW batches (bpsw_swalign2_batch / bpsw_matesw_group / mateSWJNI rounds, of any context) that are in their device
 * phase on `device` right now -- what the "lone caller takes a launch of its own" rule of the SW entry points looks at
 * (BPSW_RING_LONE_LAUNCH).  Diagnostics (an executor's metrics page; tests/test_ring_gpu.py waits on it to make
 * "a batch that has company" a deterministic state instead of a matter of thread timing).  -1 for a device index outside 0..63. */
int bpsw_sw_batches_in_flight(int device);

#ifdef __cplusplus
}
#endif
#endif
