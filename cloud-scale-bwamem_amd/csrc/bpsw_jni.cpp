// bpsw_jni.cpp -- the JNI symbols the unmodified CS-BWAMEM Scala driver binds, as thin marshalling
// shims over the C ABI (include/bpsw.h).
//
//   Java_cs_ucla_edu_bwaspark_jni_SWExtendFPGAJNI_swExtendFPGAJNI  replaces jni_fpga/sw_extend_fpga.c:116-193
//   Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_mateSWJNI              replaces native/jni_mate_sw.c:58-662
//   Java_cs_ucla_edu_bwaspark_jni_HelloWorld_helloWorld            replaces native/jni_hello_world.c:23-26
//   Java_cs_ucla_edu_bwaspark_jni_SWExtendFPGAJNI_chainToAlnJNI    NEW: the whole round loop of
//                                                                  memChainToAlnBatched in one call, primitive arrays only
//   Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_mateSWFlatJNI          NEW (round 4): boundary 1 with primitive arrays in and one long[] out
//   Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_loadPacJNI             NEW (needs one line of Scala, see
//                                                                  INTEGRATION.md): puts the 2-bit reference on every
//                                                                  visible device; mateSWJNI then accepts RefSWType
//                                                                  objects whose ref0..ref3 are null (coordinates only)
//   Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_loadBnsJNI             NEW: the contig table next to the reference
//   Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailJNI           NEW: memSamPeGroupRest for a group of
//                                                                  pairs in one call, primitive arrays in, SAM text out
//
// Differences from the reference glue that a JVM can observe: nothing is printed, the JVM is never
// exit()ed or assert()ed, a device failure surfaces as a java.lang.RuntimeException (Spark retries the
// task), local references are bounded with Push/PopLocalFrame so -sbatch can be thousands of pairs, and
// arrays are read with Get*ArrayRegion (no pinning).  No JVM exists in the build image, so this file
// is compile- and link-checked, and exercised through a fake JNIEnv table in tests/test_jni_shim.py.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <time.h>
#include <exception>
#include <vector>

#include "bpsw.h"
#include "bpsw_rescue_skip.h"
#include "jni_min.h"

namespace {

// One context per (thread, device): Spark runs several task threads per executor JVM and the reference
// native code is re-entrant, so calls from different threads must not serialise on one stream.  The context is released by
// the thread_local destructor when the task thread exits (stream, arenas, pinned staging).
struct ThreadCtx {
  bpsw_ctx_t* ctx = nullptr;
  int device = -1;
  int partition = -1, slot = -1;  // what the last call on this thread saw (bpsw_jni_thread_info, for the tests)
  ~ThreadCtx() {
    if (ctx) bpsw_destroy(ctx);
  }
};
thread_local ThreadCtx t_ctx;

void clear_pending(JNIEnv* env) {
  if (jni::ExceptionCheck(env)) jni::ExceptionClear(env);
}

// Spark partition -> device (north_star: "Spark-partition -> device index").  The partition id is not in
// either JNI signature; org.apache.spark.TaskContext.get().partitionId() is reachable through JNIEnv.  The class and the two
// method IDs are resolved once per process (a global reference keeps the class, and with it the IDs, alive).
struct SparkIds {
  jclass cls = nullptr;
  jmethodID get = nullptr, pid = nullptr;
};
std::mutex g_ids_mu;
std::atomic<const SparkIds*> g_spark{nullptr};
std::atomic<bool> g_no_spark{false};  // TaskContext is not on the class path (a harness, not an executor): do not ask again

int spark_partition_id(JNIEnv* env) {
  const SparkIds* ids = g_spark.load(std::memory_order_acquire);
  if (!ids) {
    if (g_no_spark.load(std::memory_order_relaxed)) return -1;
    std::lock_guard<std::mutex> lk(g_ids_mu);
    ids = g_spark.load(std::memory_order_acquire);
    if (!ids) {
      jclass cls = jni::FindClass(env, "org/apache/spark/TaskContext");
      if (!cls || jni::ExceptionCheck(env)) { clear_pending(env); g_no_spark.store(true); return -1; }
      SparkIds* n = new SparkIds();
      n->get = jni::GetStaticMethodID(env, cls, "get", "()Lorg/apache/spark/TaskContext;");
      n->pid = (n->get && !jni::ExceptionCheck(env)) ? jni::GetMethodID(env, cls, "partitionId", "()I") : nullptr;
      if (!n->get || !n->pid || jni::ExceptionCheck(env)) { clear_pending(env); delete n; return -1; }
      n->cls = (jclass)jni::NewGlobalRef(env, cls);
      jni::DeleteLocalRef(env, cls);
      if (!n->cls) { clear_pending(env); delete n; return -1; }
      g_spark.store(n, std::memory_order_release);
      ids = n;
    }
  }
  jobject tc = jni::CallStaticObjectMethod(env, ids->cls, ids->get);
  if (!tc || jni::ExceptionCheck(env)) { clear_pending(env); return -1; }
  const jint id = jni::CallIntMethod(env, tc, ids->pid);
  jni::DeleteLocalRef(env, tc);
  if (jni::ExceptionCheck(env)) { clear_pending(env); return -1; }
  return (int)id;
}

bpsw_ctx_t* thread_context(JNIEnv* env) {
  const int part = spark_partition_id(env);
  const int want = bpsw_device_for_partition(part);  // entry (partition mod count) of BPSW_DEVICES; -1 without a TaskContext
  t_ctx.partition = part;
  t_ctx.slot = part >= 0 && bpsw_device_slots() > 0 ? part % bpsw_device_slots() : -1;
  if (t_ctx.ctx && (want < 0 || want == t_ctx.device)) return t_ctx.ctx;
  if (t_ctx.ctx) { bpsw_destroy(t_ctx.ctx); t_ctx.ctx = nullptr; }
  // want < 0: no TaskContext (harness) -> bpsw_create spreads threads round-robin over BPSW_DEVICES
  if (bpsw_create(want, &t_ctx.ctx) != BPSW_OK) return nullptr;
  t_ctx.device = bpsw_device_of(t_ctx.ctx);
  return t_ctx.ctx;
}

double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
}
// wall time of the last JNI call on this thread, split at the C ABI: marshalling in, the bpsw_* call, marshalling out (bpsw_jni_last_times)
struct ShimTimes { double in_us, call_us, out_us, units; };
thread_local ShimTimes t_times = {0, 0, 0, 0};
thread_local int t_mate_path = 0;  // what the last mateSWJNI on this thread did: 1 the lazy path, 2 the eager one (bpsw_jni_last_mate_path, for the tests)

void throw_runtime(JNIEnv* env, const std::string& msg) {
  clear_pending(env);
  jclass rte = jni::FindClass(env, "java/lang/RuntimeException");
  if (rte) jni::ThrowNew(env, rte, msg.c_str());
}

struct OptIds {
  jfieldID a, b, oDel, eDel, oIns, eIns, penUnpaired, penClip5, penClip3, w, zdrop, T, flag, minSeedLen, maxIns,
      maxMatesw, maskLevelRedun, mat;
};
struct RegIds {
  jfieldID rBeg, rEnd, qBeg, qEnd, score, trueScore, sub, csub, subNum, width, seedCov, secondary, hash;
};

bool load_reg_ids(JNIEnv* env, jclass c, RegIds* r) {  // MemAlnRegType.scala:26-38, native/jni_mate_sw.c:131-143
  r->rBeg = jni::GetFieldID(env, c, "rBeg", "J"); r->rEnd = jni::GetFieldID(env, c, "rEnd", "J");
  r->qBeg = jni::GetFieldID(env, c, "qBeg", "I"); r->qEnd = jni::GetFieldID(env, c, "qEnd", "I");
  r->score = jni::GetFieldID(env, c, "score", "I"); r->trueScore = jni::GetFieldID(env, c, "trueScore", "I");
  r->sub = jni::GetFieldID(env, c, "sub", "I"); r->csub = jni::GetFieldID(env, c, "csub", "I");
  r->subNum = jni::GetFieldID(env, c, "subNum", "I"); r->width = jni::GetFieldID(env, c, "width", "I");
  r->seedCov = jni::GetFieldID(env, c, "seedCov", "I"); r->secondary = jni::GetFieldID(env, c, "secondary", "I");
  r->hash = jni::GetFieldID(env, c, "hash", "J");
  return r->rBeg && r->rEnd && r->qBeg && r->qEnd && r->score && r->trueScore && r->sub && r->csub && r->subNum &&
         r->width && r->seedCov && r->secondary && r->hash && !jni::ExceptionCheck(env);
}

// Everything mateSWJNI looks up by name (native/jni_mate_sw.c:102-143 does the same ~55 GetFieldID + 6 FindClass on every
// call): resolved on the first call, classes held as global references so that the field IDs stay valid.  FindClass from a
// native method consults the class loader of the calling class (MateSWJNI's), the same on every call.
struct MateIds {
  jclass regCls, optCls, pesCls, mateCls, seqCls, refCls;
  jfieldID opt_int[16], opt_mlr, opt_mat;
  jfieldID pes_low, pes_high, pes_failed, pes_avg, pes_std;
  jfieldID seq_rid, seq_pid, seq_len, seq_trans;
  RegIds rf;
  jfieldID mRid, mPid, mReg, mAln;
  jfieldID rRid, rPid, rReg, rB, rE, rL, rRef[4];
};
std::atomic<const MateIds*> g_mate{nullptr};

// nullptr: a class or field is missing; the JVM's NoClassDefFoundError / NoSuchFieldError is pending
const MateIds* mate_ids(JNIEnv* env) {
  const MateIds* ids = g_mate.load(std::memory_order_acquire);
  if (ids) return ids;
  std::lock_guard<std::mutex> lk(g_ids_mu);
  ids = g_mate.load(std::memory_order_acquire);
  if (ids) return ids;
  MateIds m;
  memset(&m, 0, sizeof m);
  jclass regCls = jni::FindClass(env, "cs/ucla/edu/bwaspark/datatype/MemAlnRegType");
  jclass optCls = regCls ? jni::FindClass(env, "cs/ucla/edu/bwaspark/datatype/MemOptType") : nullptr;
  jclass pesCls = optCls ? jni::FindClass(env, "cs/ucla/edu/bwaspark/datatype/MemPeStat") : nullptr;
  jclass mateCls = pesCls ? jni::FindClass(env, "cs/ucla/edu/bwaspark/jni/MateSWType") : nullptr;
  jclass seqCls = mateCls ? jni::FindClass(env, "cs/ucla/edu/bwaspark/jni/SeqSWType") : nullptr;
  jclass refCls = seqCls ? jni::FindClass(env, "cs/ucla/edu/bwaspark/jni/RefSWType") : nullptr;
  if (!refCls) return nullptr;
  static const char* opt_names[16] = {"a", "b", "oDel", "eDel", "oIns", "eIns", "penUnpaired", "penClip5", "penClip3", "w",
                                      "zdrop", "T", "flag", "minSeedLen", "maxIns", "maxMatesw"};  // native/jni_mate_sw.c:102-128
  for (int i = 0; i < 16; ++i)
    if (!(m.opt_int[i] = jni::GetFieldID(env, optCls, opt_names[i], "I"))) return nullptr;
  m.opt_mlr = jni::GetFieldID(env, optCls, "maskLevelRedun", "F");
  m.opt_mat = jni::GetFieldID(env, optCls, "mat", "[B");
  m.pes_low = jni::GetFieldID(env, pesCls, "low", "I"); m.pes_high = jni::GetFieldID(env, pesCls, "high", "I");
  m.pes_failed = jni::GetFieldID(env, pesCls, "failed", "I");
  m.pes_avg = jni::GetFieldID(env, pesCls, "avg", "D"); m.pes_std = jni::GetFieldID(env, pesCls, "std", "D");
  m.seq_rid = jni::GetFieldID(env, seqCls, "readIdx", "I"); m.seq_pid = jni::GetFieldID(env, seqCls, "pairIdx", "I");
  m.seq_len = jni::GetFieldID(env, seqCls, "seqLength", "I"); m.seq_trans = jni::GetFieldID(env, seqCls, "seqTrans", "[B");
  if (!load_reg_ids(env, regCls, &m.rf)) return nullptr;
  m.mRid = jni::GetFieldID(env, mateCls, "readIdx", "I"); m.mPid = jni::GetFieldID(env, mateCls, "pairIdx", "I");
  m.mReg = jni::GetFieldID(env, mateCls, "regIdx", "I");
  m.mAln = jni::GetFieldID(env, mateCls, "alnReg", "Lcs/ucla/edu/bwaspark/datatype/MemAlnRegType;");
  m.rRid = jni::GetFieldID(env, refCls, "readIdx", "I"); m.rPid = jni::GetFieldID(env, refCls, "pairIdx", "I");
  m.rReg = jni::GetFieldID(env, refCls, "regIdx", "I");
  m.rB = jni::GetFieldID(env, refCls, "rBegArray", "[J"); m.rE = jni::GetFieldID(env, refCls, "rEndArray", "[J");
  m.rL = jni::GetFieldID(env, refCls, "lenArray", "[J");
  static const char* ref_names[4] = {"ref0", "ref1", "ref2", "ref3"};
  for (int r = 0; r < 4; ++r) m.rRef[r] = jni::GetFieldID(env, refCls, ref_names[r], "[B");
  if (!m.opt_mlr || !m.opt_mat || !m.pes_low || !m.pes_high || !m.pes_failed || !m.pes_avg || !m.pes_std || !m.seq_rid ||
      !m.seq_pid || !m.seq_len || !m.seq_trans || !m.mRid || !m.mPid || !m.mReg || !m.mAln || !m.rRid || !m.rPid || !m.rReg ||
      !m.rB || !m.rE || !m.rL || !m.rRef[0] || !m.rRef[1] || !m.rRef[2] || !m.rRef[3] || jni::ExceptionCheck(env))
    return nullptr;
  m.regCls = (jclass)jni::NewGlobalRef(env, regCls); m.optCls = (jclass)jni::NewGlobalRef(env, optCls);
  m.pesCls = (jclass)jni::NewGlobalRef(env, pesCls); m.mateCls = (jclass)jni::NewGlobalRef(env, mateCls);
  m.seqCls = (jclass)jni::NewGlobalRef(env, seqCls); m.refCls = (jclass)jni::NewGlobalRef(env, refCls);
  if (!m.regCls || !m.optCls || !m.pesCls || !m.mateCls || !m.seqCls || !m.refCls) return nullptr;  // OutOfMemoryError pending
  const MateIds* pub = new MateIds(m);
  g_mate.store(pub, std::memory_order_release);
  return pub;
}

// Byte pool that is never value-initialised: a std::vector<uint8_t>::resize would zero-fill what GetByteArrayRegion is about to
// overwrite (a second touch of every sequence and window byte of a call).
struct BytePool {
  uint8_t* p = nullptr;
  size_t n = 0, cap = 0;
  ~BytePool() { free(p); }
  void clear() { n = 0; }
  uint8_t* grow(size_t add) {
    if (n + add > cap) {
      size_t want = cap ? cap : (size_t)1 << 20;
      while (want < n + add) want <<= 1;
      p = static_cast<uint8_t*>(realloc(p, want));
      cap = want;
    }
    uint8_t* at = p + n;
    n += add;
    return at;
  }
  void pad16() {  // keep every sequence 16-byte aligned for the device; the pad bytes are zero
    const size_t pad = (16 - (n & 15)) & 15;
    if (pad) memset(grow(pad), 0, pad);
  }
};
// What a mateSWJNI call builds for bpsw_matesw_group, kept per thread and reused: a call of 4 096 pairs moves ~10 MB through these,
// and allocating (and zero-filling) them afresh on every call was a measurable part of the shim.
struct MateScratch {
  std::vector<int32_t> seq_len, reg_cnt, ref_cnt, out_cnt, tmp_cnt;
  std::vector<int64_t> seq_off, at, base, ref_rb, ref_re, ref_len, ref_off;
  std::vector<long> where;
  std::vector<bpsw_alnreg_t> regs, tmp, out;
  BytePool seq_pool, ref_pool;
};
thread_local MateScratch t_ms;

void read_bytes(JNIEnv* env, jbyteArray arr, BytePool& pool, int64_t* off, int32_t* len) {
  *off = (int64_t)pool.n;
  *len = 0;
  if (!arr) return;
  const jsize n = jni::GetArrayLength(env, arr);
  if (n > 0) jni::GetByteArrayRegion(env, arr, 0, n, reinterpret_cast<jbyte*>(pool.grow((size_t)n)));
  *len = n;
  pool.pad16();
}


// ---- mateSWJNI without the object walk over pairs that need nothing (round 5) -------------------------------------------------------
// The reference's contract hands over every region, every mate and every rescue window of a group as objects (native/jni_mate_sw.c:
// 239-518), and 96 % of a 4 096-pair call through this shim was reading and rebuilding them -- while nine pairs in ten are properly
// paired, need no SW at all and come back exactly as they went in.  So: (1) one light pass over MateSWType[] reads what the skip test
// needs of a region (its end, rBeg, score: 6 JNI calls instead of 18) and checks that the array is in (pair, end, rank) order with
// regIdx = rank, as memSamPeGroupJNIPrepare builds it (MemSamPe.scala:1962-1990); (2) the skip test of the library itself
// (bpsw_rescue_skip.h) selects the pairs that may need a job; (3) only THEIR regions (all fields), mates and windows are unmarshalled,
// into a group of their own, and bpsw_matesw_group runs on that; (4) the result array holds new objects for those pairs and, for every
// other end, the caller's own MateSWType objects -- they already carry (readIdx, pairIdx, regIdx = rank) and the unchanged region.
// The Scala caller rebuilds its lists from the returned array and drops the input arrays (MemSamPe.scala:2010-2044), so sharing the
// objects is safe.  Anything that does not look like what memSamPeGroupJNIPrepare builds (SeqSWType[] not one per end in order,
// RefSWType[] not in (pair, end, anchor) order, regIdx != rank, a malformed object) returns 1: the caller takes the eager path below,
// which accepts any order and raises the errors.  BPSW_JNI_LAZY=0: always the eager path.
struct LightReg { int64_t rb; int32_t score, e; };
struct LazyScratch {
  std::vector<LightReg> light;
  std::vector<int64_t> reg_at, base;
  std::vector<int32_t> reg_cnt, ref_cnt, touched;
  std::vector<uint8_t> is_touched;
};
thread_local LazyScratch t_lazy;

void read_region(JNIEnv* env, jobject a, const RegIds& rf, bpsw_alnreg_t* r) {
  r->rb = jni::GetLongField(env, a, rf.rBeg); r->re = jni::GetLongField(env, a, rf.rEnd);
  r->qb = jni::GetIntField(env, a, rf.qBeg); r->qe = jni::GetIntField(env, a, rf.qEnd);
  r->score = jni::GetIntField(env, a, rf.score); r->truesc = jni::GetIntField(env, a, rf.trueScore);
  r->sub = jni::GetIntField(env, a, rf.sub); r->csub = jni::GetIntField(env, a, rf.csub);
  r->sub_n = jni::GetIntField(env, a, rf.subNum); r->w = jni::GetIntField(env, a, rf.width);
  r->seedcov = jni::GetIntField(env, a, rf.seedCov); r->secondary = jni::GetIntField(env, a, rf.secondary);
  r->hash = (uint64_t)jni::GetLongField(env, a, rf.hash);
}
jobject new_mate_object(JNIEnv* env, const MateIds* ids, jint k, jint i, jint rank, const bpsw_alnreg_t& r) {
  const RegIds& rf = ids->rf;
  jobject m = jni::AllocObject(env, ids->mateCls), a = jni::AllocObject(env, ids->regCls);
  if (!m || !a) return nullptr;
  jni::SetIntField(env, m, ids->mRid, k); jni::SetIntField(env, m, ids->mPid, i); jni::SetIntField(env, m, ids->mReg, rank);
  jni::SetLongField(env, a, rf.rBeg, r.rb); jni::SetLongField(env, a, rf.rEnd, r.re);
  jni::SetIntField(env, a, rf.qBeg, r.qb); jni::SetIntField(env, a, rf.qEnd, r.qe);
  jni::SetIntField(env, a, rf.score, r.score); jni::SetIntField(env, a, rf.trueScore, r.truesc);
  jni::SetIntField(env, a, rf.sub, r.sub); jni::SetIntField(env, a, rf.csub, r.csub);
  jni::SetIntField(env, a, rf.subNum, r.sub_n); jni::SetIntField(env, a, rf.width, r.w);
  jni::SetIntField(env, a, rf.seedCov, r.seedcov); jni::SetIntField(env, a, rf.secondary, r.secondary);
  jni::SetLongField(env, a, rf.hash, (jlong)r.hash);
  jni::SetObjectField(env, m, ids->mAln, a);
  return m;
}

// 0: done (*ret_out set), 1: take the eager path, -1: a Java exception is pending
int mate_sw_lazy(JNIEnv* env, const MateIds* ids, const bpsw_opt_t& opt, const bpsw_rescue_group_t& g0, jint groupSize, jobjectArray seqArr,
                 jobjectArray mateArr, jobjectArray refArr, jintArray refSizeArr, double t0, jobjectArray* ret_out) {
  constexpr jint FRAME = 512;   // objects per local frame
  LazyScratch& L = t_lazy;
  MateScratch& ms = t_ms;
  const size_t ends = 2 * (size_t)groupSize;
  const jsize n_mate = jni::GetArrayLength(env, mateArr), n_seq = jni::GetArrayLength(env, seqArr), n_ref = jni::GetArrayLength(env, refArr);
  if ((size_t)n_seq != ends || (size_t)jni::GetArrayLength(env, refSizeArr) < ends) return 1;
  L.ref_cnt.assign(ends ? ends : 1, 0);
  if (ends) jni::GetIntArrayRegion(env, refSizeArr, 0, (jsize)ends, L.ref_cnt.data());
  L.base.assign(ends + 1, 0);
  for (size_t e = 0; e < ends; ++e) {
    if (L.ref_cnt[e] < 0) return 1;
    L.base[e + 1] = L.base[e] + L.ref_cnt[e];
  }
  if (L.base[ends] != (int64_t)n_ref) return 1;
  // ---- (1) the light pass ----
  L.light.resize((size_t)n_mate);
  L.reg_cnt.assign(ends ? ends : 1, 0);
  {
    long prev_e = -1;
    jint rank = 0;
    for (jsize s0 = 0; s0 < n_mate; s0 += FRAME) {
      if (jni::PushLocalFrame(env, 2 * FRAME + 8) != JNI_OK) return -1;
      const jsize s1 = s0 + FRAME < n_mate ? s0 + FRAME : n_mate;
      for (jsize s = s0; s < s1; ++s) {
        jobject o = jni::GetObjectArrayElement(env, mateArr, s);
        if (!o) { jni::PopLocalFrame(env, nullptr); return 1; }
        const jint k = jni::GetIntField(env, o, ids->mRid), i = jni::GetIntField(env, o, ids->mPid);
        if (k < 0 || k >= groupSize || i < 0 || i > 1) { jni::PopLocalFrame(env, nullptr); return 1; }
        const long e = 2l * k + i;
        rank = e == prev_e ? rank + 1 : 0;
        if (e < prev_e || jni::GetIntField(env, o, ids->mReg) != rank) { jni::PopLocalFrame(env, nullptr); return 1; }
        prev_e = e;
        jobject a = jni::GetObjectField(env, o, ids->mAln);
        if (!a) { jni::PopLocalFrame(env, nullptr); return 1; }
        LightReg& lr = L.light[(size_t)s];
        lr.rb = jni::GetLongField(env, a, ids->rf.rBeg);
        lr.score = jni::GetIntField(env, a, ids->rf.score);
        lr.e = (int32_t)e;
        ++L.reg_cnt[(size_t)e];
      }
      jni::PopLocalFrame(env, nullptr);
    }
  }
  L.reg_at.assign(ends + 1, 0);
  for (size_t e = 0; e < ends; ++e) L.reg_at[e + 1] = L.reg_at[e] + L.reg_cnt[e];
  // ---- (2) which pairs may need a job: the library's own skip test on (rBeg, score); window validity and empty mates are left to
  // the library (a superset of the pairs it will really touch) ----
  L.touched.clear();
  L.is_touched.assign((size_t)groupSize + 1, 0);
  if ((opt.flag & 0x20) == 0) {  // MEM_F_NO_RESCUE
    int32_t low[4], high[4];
    int failed_mask = 0;
    for (int r = 0; r < 4; ++r) { low[r] = g0.pes[r].low; high[r] = g0.pes[r].high; failed_mask |= (g0.pes[r].failed ? 1 : 0) << r; }
    const char* compat = getenv("BPSW_MATESW_COMPAT");
    const bool scala = compat && strcmp(compat, "scala") == 0;
    for (jint k = 0; k < groupSize; ++k) {
      bool touched = false;
      for (int i = 0; i < 2 && !touched; ++i) {
        const size_t e = 2 * (size_t)k + (size_t)i, mate = e ^ 1;
        if (L.reg_cnt[e] == 0 || L.ref_cnt[e] == 0) continue;
        const LightReg* init = L.light.data() + L.reg_at[e];
        const LightReg* minit = L.light.data() + L.reg_at[mate];
        const int thr = init[0].score - opt.pen_unpaired;
        int j = 0;
        for (int ai = 0; ai < L.reg_cnt[e] && !touched; ++ai) {
          if (!(init[ai].score >= thr)) continue;
          if (j >= opt.max_matesw || j >= L.ref_cnt[e]) break;
          int skip[4];
          bpsw::rescue_skip_flags(g0.l_pac, low, high, failed_mask, scala, init[ai].rb, &minit->rb, sizeof(LightReg), (size_t)L.reg_cnt[mate], skip);
          if (skip[0] + skip[1] + skip[2] + skip[3] != 4) touched = true;
          ++j;
        }
      }
      if (touched) { L.touched.push_back(k); L.is_touched[(size_t)k] = 1; }
    }
  }
  const size_t nt = L.touched.size();
  // ---- (3) the touched pairs as a group of their own ----
  const size_t rends = 2 * nt;
  std::vector<int32_t>&seq_len = ms.seq_len, &reg_cnt = ms.reg_cnt, &ref_cnt = ms.ref_cnt;
  std::vector<int64_t>& seq_off = ms.seq_off;
  seq_len.assign(rends ? rends : 1, 0); reg_cnt.assign(rends ? rends : 1, 0); ref_cnt.assign(rends ? rends : 1, 0); seq_off.assign(rends ? rends : 1, 0);
  BytePool &seq_pool = ms.seq_pool, &ref_pool = ms.ref_pool;
  seq_pool.clear(); ref_pool.clear();
  std::vector<bpsw_alnreg_t>& regs = ms.regs;
  regs.clear();
  std::vector<int64_t>&ref_rb = ms.ref_rb, &ref_re = ms.ref_re, &ref_len = ms.ref_len, &ref_off = ms.ref_off;
  size_t rows = 0;
  for (size_t ti = 0; ti < nt; ++ti)
    for (int i = 0; i < 2; ++i) rows += (size_t)L.ref_cnt[2 * (size_t)L.touched[ti] + (size_t)i];
  ref_rb.assign(4 * rows, -1); ref_re.assign(4 * rows, -1); ref_len.assign(4 * rows, 0); ref_off.assign(4 * rows, 0);
  size_t coord_windows = 0, byte_windows = 0, row_at = 0;
  for (size_t ti = 0; ti < nt; ++ti) {
    const jint k = L.touched[ti];
    for (int i = 0; i < 2; ++i) {
      const size_t e = 2 * (size_t)k + (size_t)i, re = 2 * ti + (size_t)i;
      if (jni::PushLocalFrame(env, 32 + 2 * L.reg_cnt[e] + 16 * L.ref_cnt[e]) != JNI_OK) return -1;
      {  // the mate (SeqSWType[e] is end e: checked)
        jobject o = jni::GetObjectArrayElement(env, seqArr, (jsize)e);
        if (!o || jni::GetIntField(env, o, ids->seq_rid) != k || jni::GetIntField(env, o, ids->seq_pid) != i) { jni::PopLocalFrame(env, nullptr); return 1; }
        jbyteArray bytes = (jbyteArray)jni::GetObjectField(env, o, ids->seq_trans);
        int32_t got = 0;
        read_bytes(env, bytes, seq_pool, &seq_off[re], &got);
        const jint declared = jni::GetIntField(env, o, ids->seq_len);
        seq_len[re] = declared < got ? declared : got;
      }
      for (int64_t s = L.reg_at[e]; s < L.reg_at[e + 1]; ++s) {  // the regions, all fields
        jobject o = jni::GetObjectArrayElement(env, mateArr, (jsize)s);
        jobject a = o ? jni::GetObjectField(env, o, ids->mAln) : nullptr;
        if (!a) { jni::PopLocalFrame(env, nullptr); return 1; }
        regs.emplace_back();
        read_region(env, a, ids->rf, &regs.back());
      }
      reg_cnt[re] = L.reg_cnt[e];
      ref_cnt[re] = L.ref_cnt[e];
      for (int32_t j = 0; j < L.ref_cnt[e]; ++j, ++row_at) {  // the anchors' windows (RefSWType[base[e] + j] is anchor j of end e: checked)
        jobject o = jni::GetObjectArrayElement(env, refArr, (jsize)(L.base[e] + j));
        if (!o || jni::GetIntField(env, o, ids->rRid) != k || jni::GetIntField(env, o, ids->rPid) != i || jni::GetIntField(env, o, ids->rReg) != j) {
          jni::PopLocalFrame(env, nullptr); return 1;
        }
        const size_t x = 4 * row_at;
        jlongArray ab = (jlongArray)jni::GetObjectField(env, o, ids->rB), ae = (jlongArray)jni::GetObjectField(env, o, ids->rE);
        jlongArray al = (jlongArray)jni::GetObjectField(env, o, ids->rL);
        if (!ab || !ae || !al || jni::GetArrayLength(env, ab) < 4 || jni::GetArrayLength(env, ae) < 4 || jni::GetArrayLength(env, al) < 4) {
          jni::PopLocalFrame(env, nullptr); return 1;
        }
        jni::GetLongArrayRegion(env, ab, 0, 4, (jlong*)&ref_rb[x]);
        jni::GetLongArrayRegion(env, ae, 0, 4, (jlong*)&ref_re[x]);
        jni::GetLongArrayRegion(env, al, 0, 4, (jlong*)&ref_len[x]);
        for (int r = 0; r < 4; ++r) {
          if (ref_rb[x + r] < 0 && ref_re[x + r] < 0) continue;  // failed orientation (MemSamPe.scala:1863-1868)
          jbyteArray bytes = (jbyteArray)jni::GetObjectField(env, o, ids->rRef[r]);
          if (!bytes && ref_len[x + r] != 0) { ++coord_windows; continue; }
          ++byte_windows;
          if (ref_len[x + r] <= 0) continue;
          int32_t got = 0;
          read_bytes(env, bytes, ref_pool, &ref_off[x + r], &got);
          if (got < ref_len[x + r]) { jni::PopLocalFrame(env, nullptr); return 1; }
        }
      }
      jni::PopLocalFrame(env, nullptr);
    }
  }
  if (coord_windows > 0 && byte_windows > 0) return 1;
  if (seq_pool.n == 0) memset(seq_pool.grow(16), 0, 16);
  if (ref_pool.n == 0) memset(ref_pool.grow(16), 0, 16);
  bpsw_rescue_group_t g = g0;
  g.group_size = (int32_t)nt;
  g.seq_len = seq_len.data(); g.seq_off = seq_off.data(); g.seq_pool = seq_pool.p; g.seq_pool_bytes = seq_pool.n;
  g.reg_cnt = reg_cnt.data(); g.regs = regs.data(); g.ref_cnt = ref_cnt.data();
  g.ref_rb = ref_rb.data(); g.ref_re = ref_re.data(); g.ref_len = ref_len.data(); g.ref_off = ref_off.data();
  g.ref_pool = ref_pool.p; g.ref_pool_bytes = ref_pool.n;
  if (coord_windows > 0) { g.ref_pool = nullptr; g.ref_pool_bytes = 0; g.ref_len = nullptr; g.ref_off = nullptr; }
  std::vector<int32_t>& out_cnt = ms.out_cnt;
  std::vector<bpsw_alnreg_t>& out = ms.out;
  out_cnt.assign(rends ? rends : 1, 0);
  int64_t total_r = 0;
  const double t1 = now_us();
  if (nt) {
    bpsw_ctx_t* ctx = thread_context(env);
    if (!ctx) { throw_runtime(env, std::string("bPSW: no usable HIP device: ") + bpsw_last_error()); return -1; }
    out.resize(regs.size() + rows + 16);
    const char* compat = getenv("BPSW_MATESW_COMPAT");
    const int mode = (compat && strcmp(compat, "scala") == 0) ? BPSW_RESCUE_SCALA : BPSW_RESCUE_C;
    int rc = bpsw_matesw_group(ctx, &opt, &g, mode, out_cnt.data(), out.data(), (int64_t)out.size(), &total_r);
    if (rc == BPSW_ERR_CAPACITY) {
      out.resize((size_t)total_r);
      rc = bpsw_matesw_group(ctx, &opt, &g, mode, out_cnt.data(), out.data(), (int64_t)out.size(), &total_r);
    }
    if (rc != BPSW_OK) { throw_runtime(env, std::string("bPSW: mateSWJNI: ") + bpsw_last_error()); return -1; }
  } else if (!thread_context(env)) {  // (a call that needs no device still fails without one, like every other: no silent CPU path)
    throw_runtime(env, std::string("bPSW: no usable HIP device: ") + bpsw_last_error());
    return -1;
  }
  // ---- (4) the result: (pair, end, rank) order; untouched ends keep the caller's objects ----
  const double t2 = now_us();
  int64_t total = total_r;
  for (jint k = 0; k < groupSize; ++k)
    if (!L.is_touched[(size_t)k]) total += L.reg_at[2 * (size_t)k + 2] - L.reg_at[2 * (size_t)k];
  jobjectArray ret = jni::NewObjectArray(env, (jsize)total, ids->mateCls, nullptr);
  if (!ret) return -1;
  int64_t at = 0, rat = 0;
  size_t ti = 0;
  jint in_frame = 0;
  if (jni::PushLocalFrame(env, 2 * FRAME + 8) != JNI_OK) return -1;
  for (jint k = 0; k < groupSize; ++k) {
    if (L.is_touched[(size_t)k]) {
      for (int i = 0; i < 2; ++i)
        for (int32_t rank = 0; rank < out_cnt[2 * ti + (size_t)i]; ++rank, ++at, ++rat) {
          jobject m = new_mate_object(env, ids, k, i, rank, out[(size_t)rat]);
          if (!m) { jni::PopLocalFrame(env, nullptr); return -1; }
          jni::SetObjectArrayElement(env, ret, (jsize)at, m);
          in_frame += 2;
        }
      ++ti;
    } else {
      for (int64_t s = L.reg_at[2 * (size_t)k]; s < L.reg_at[2 * (size_t)k + 2]; ++s, ++at) {
        jobject o = jni::GetObjectArrayElement(env, mateArr, (jsize)s);
        jni::SetObjectArrayElement(env, ret, (jsize)at, o);
        ++in_frame;
      }
    }
    if (in_frame >= FRAME) {
      jni::PopLocalFrame(env, nullptr);
      if (jni::PushLocalFrame(env, 2 * FRAME + 8) != JNI_OK) return -1;
      in_frame = 0;
    }
  }
  jni::PopLocalFrame(env, nullptr);
  t_times = {t1 - t0, t2 - t1, now_us() - t2, (double)total};
  t_mate_path = 1;
  *ret_out = ret;
  return 0;
}

}  // namespace

extern "C" {

// What the last JNI call on the calling thread resolved (for the tests; not a JNI symbol): out = {partition id seen (-1: no
// TaskContext), entry of BPSW_DEVICES it maps to, HIP device of the thread's context}; returns the context as an integer
// (two threads never share one) or 0 when the thread has none.
JNIEXPORT uint64_t bpsw_jni_thread_info(int32_t out[3]) {
  if (out) { out[0] = t_ctx.partition; out[1] = t_ctx.slot; out[2] = t_ctx.device; }
  return (uint64_t)(uintptr_t)t_ctx.ctx;
}

// Wall time of the last swExtendFPGAJNI / mateSWJNI call on the calling thread (for the shim micro-benchmark; not a JNI symbol):
// out = {marshalling in (us), the C ABI call (us), marshalling out (us), units (tasks / regions returned)}.
JNIEXPORT int bpsw_jni_last_mate_path(void) { return t_mate_path; }
JNIEXPORT void bpsw_jni_last_times(double out[4]) {
  if (out) { out[0] = t_times.in_us; out[1] = t_times.call_us; out[2] = t_times.out_us; out[3] = t_times.units; }
}

JNIEXPORT void JNICALL Java_cs_ucla_edu_bwaspark_jni_HelloWorld_helloWorld(JNIEnv*, jobject) {
  printf("Hello World from %s (%d HIP device(s))\n", bpsw_version(), bpsw_device_count());
}

// ---- boundary 2 ------------------------------------------------------------------------------------
JNIEXPORT jshortArray JNICALL Java_cs_ucla_edu_bwaspark_jni_SWExtendFPGAJNI_swExtendFPGAJNI(JNIEnv* env, jobject,
                                                                                            jint retTaskNum,
                                                                                            jbyteArray arrayIn) {
  try {
  if (!arrayIn || retTaskNum < 0) { throw_runtime(env, "bPSW: swExtendFPGAJNI: bad arguments"); return nullptr; }
  const double t0 = now_us();
  const jsize bytes = jni::GetArrayLength(env, arrayIn);
  bpsw_ctx_t* ctx = thread_context(env);
  if (!ctx) { throw_runtime(env, std::string("bPSW: no usable HIP device: ") + bpsw_last_error()); return nullptr; }
  // Single touch: the JVM's bytes go straight into the context's pinned staging block, from where the copy engine reads them
  // (the reference does one memcpy into its shared-memory segment, src/main/jni_fpga/sw_extend_fpga.c:146-155); the results are
  // handed to SetShortArrayRegion from the pinned block the kernel wrote them to.  No heap allocation, no second copy.
  uint8_t* stage = nullptr;
  if (bpsw_extend_stage(ctx, (size_t)bytes, &stage) != BPSW_OK) { throw_runtime(env, std::string("bPSW: swExtendFPGAJNI: ") + bpsw_last_error()); return nullptr; }
  if (bytes > 0) jni::GetByteArrayRegion(env, arrayIn, 0, bytes, reinterpret_cast<jbyte*>(stage));
  const double t1 = now_us();
  const int16_t* res = nullptr;
  size_t res_len = 0;
  const int rc = bpsw_extend_commit(ctx, (size_t)bytes, &res, &res_len);
  if (rc != BPSW_OK) { throw_runtime(env, std::string("bPSW: swExtendFPGAJNI: ") + bpsw_last_error()); return nullptr; }
  if (res_len > (size_t)retTaskNum) { throw_runtime(env, "bPSW: swExtendFPGAJNI: retTaskNum smaller than 10 shorts per task"); return nullptr; }
  const double t2 = now_us();
  jshortArray ret = jni::NewShortArray(env, retTaskNum);  // (a new Java array is zero-filled by the JVM)
  if (!ret) return nullptr;  // OutOfMemoryError already pending
  if (res_len > 0) jni::SetShortArrayRegion(env, ret, 0, (jsize)res_len, res);
  t_times = {t1 - t0, t2 - t1, now_us() - t2, (double)(res_len / 10)};
  return ret;
  } catch (const std::exception& e) {  // nothing C++ may unwind into the JVM (std::bad_alloc of a scratch vector, ...)
    throw_runtime(env, std::string("bPSW: swExtendFPGAJNI: ") + e.what());
    return nullptr;
  }
}

// ---- boundary 1 ------------------------------------------------------------------------------------
JNIEXPORT jobjectArray JNICALL Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_mateSWJNI(
    JNIEnv* env, jobject, jobject optObj, jlong pacLen, jobjectArray pesArr, jint groupSize, jobjectArray seqArr,
    jobjectArray mateArr, jobjectArray refArr, jintArray refSizeArr) {
  try {
  if (!optObj || !pesArr || !seqArr || !mateArr || !refArr || !refSizeArr || groupSize < 0) {
    throw_runtime(env, "bPSW: mateSWJNI: bad arguments");
    return nullptr;
  }
  const double t0 = now_us();
  const MateIds* ids = mate_ids(env);
  if (!ids) return nullptr;  // NoClassDefFoundError / NoSuchFieldError pending
  jclass regCls = ids->regCls, mateCls = ids->mateCls;
  // ---- MemOptType (native/jni_mate_sw.c:102-128, 177-221) ----
  bpsw_opt_t opt;
  bpsw_opt_default(&opt);
  {
    int32_t* dst[16] = {&opt.a, &opt.b, &opt.o_del, &opt.e_del, &opt.o_ins, &opt.e_ins, &opt.pen_unpaired, &opt.pen_clip5,
                        &opt.pen_clip3, &opt.w, &opt.zdrop, &opt.T, &opt.flag, &opt.min_seed_len, &opt.max_ins, &opt.max_matesw};
    for (int i = 0; i < 16; ++i) *dst[i] = jni::GetIntField(env, optObj, ids->opt_int[i]);
    jfieldID mlr = ids->opt_mlr, matId = ids->opt_mat;
    opt.mask_level_redun = jni::GetFloatField(env, optObj, mlr);
    jbyteArray matArr = (jbyteArray)jni::GetObjectField(env, optObj, matId);
    if (!matArr || jni::GetArrayLength(env, matArr) < 25) { throw_runtime(env, "bPSW: mateSWJNI: opt.mat must hold 25 bytes"); return nullptr; }
    jni::GetByteArrayRegion(env, matArr, 0, 25, reinterpret_cast<jbyte*>(opt.mat));
    jni::DeleteLocalRef(env, matArr);
  }

  bpsw_rescue_group_t g;
  memset(&g, 0, sizeof g);
  g.group_size = groupSize;
  g.l_pac = pacLen;
  {  // ---- MemPeStat[4] (native/jni_mate_sw.c:225-236) ----
    jfieldID low = ids->pes_low, high = ids->pes_high, failed = ids->pes_failed, avg = ids->pes_avg, sd = ids->pes_std;
    for (int r = 0; r < 4; ++r) {
      jobject o = jni::GetObjectArrayElement(env, pesArr, r);
      if (!o) { throw_runtime(env, "bPSW: mateSWJNI: pes must hold 4 MemPeStat"); return nullptr; }
      g.pes[r].low = jni::GetIntField(env, o, low); g.pes[r].high = jni::GetIntField(env, o, high);
      g.pes[r].failed = jni::GetIntField(env, o, failed);
      g.pes[r].avg = jni::GetDoubleField(env, o, avg); g.pes[r].std = jni::GetDoubleField(env, o, sd);
      jni::DeleteLocalRef(env, o);
    }
  }
  {  // the lazy path (above): only the pairs that may need a job are unmarshalled; 1 = the arrays are not in the order it relies on
    static const bool lazy_on = !(getenv("BPSW_JNI_LAZY") && atoi(getenv("BPSW_JNI_LAZY")) == 0);
    if (lazy_on) {
      jobjectArray lazy_ret = nullptr;
      const int st = mate_sw_lazy(env, ids, opt, g, groupSize, seqArr, mateArr, refArr, refSizeArr, t0, &lazy_ret);
      if (st == 0) return lazy_ret;
      if (st < 0) return nullptr;
    }
  }
  const size_t ends = 2 * (size_t)groupSize;
  MateScratch& ms = t_ms;
  std::vector<int32_t>&seq_len = ms.seq_len, &reg_cnt = ms.reg_cnt, &ref_cnt = ms.ref_cnt;
  std::vector<int64_t>& seq_off = ms.seq_off;
  seq_len.assign(ends, 0); reg_cnt.assign(ends, 0); ref_cnt.assign(ends, 0); seq_off.assign(ends, 0);
  BytePool &seq_pool = ms.seq_pool, &ref_pool = ms.ref_pool;
  seq_pool.clear(); ref_pool.clear();
  auto end_index = [&](jint k, jint i) -> long { return (k < 0 || k >= groupSize || i < 0 || i > 1) ? -1 : 2l * k + i; };

  {  // ---- SeqSWType[] (native/jni_mate_sw.c:258-278) ----
    jfieldID rid = ids->seq_rid, pid = ids->seq_pid, slen = ids->seq_len, strans = ids->seq_trans;
    const jsize n = jni::GetArrayLength(env, seqArr);
    for (jsize s = 0; s < n; ++s) {
      jobject o = jni::GetObjectArrayElement(env, seqArr, s);
      const long e = o ? end_index(jni::GetIntField(env, o, rid), jni::GetIntField(env, o, pid)) : -1;
      if (e < 0) { throw_runtime(env, "bPSW: mateSWJNI: SeqSWType index outside the group"); return nullptr; }
      jbyteArray bytes = (jbyteArray)jni::GetObjectField(env, o, strans);
      int32_t got = 0;
      read_bytes(env, bytes, seq_pool, &seq_off[(size_t)e], &got);
      const jint declared = jni::GetIntField(env, o, slen);
      seq_len[(size_t)e] = declared < got ? declared : got;
      if (bytes) jni::DeleteLocalRef(env, bytes);
      jni::DeleteLocalRef(env, o);
    }
  }
  const RegIds& rf = ids->rf;
  jfieldID mRid = ids->mRid, mPid = ids->mPid, mReg = ids->mReg, mAln = ids->mAln;

  std::vector<bpsw_alnreg_t>& regs = ms.regs;
  {  // ---- MateSWType[] -> regions grouped by (k,i) in arrival order (native/jni_mate_sw.c:300-345) ----
    const jsize n = jni::GetArrayLength(env, mateArr);
    std::vector<bpsw_alnreg_t>& tmp = ms.tmp;
    std::vector<long>& where = ms.where;
    tmp.resize((size_t)n); where.resize((size_t)n);
    for (jsize s = 0; s < n; ++s) {
      jobject o = jni::GetObjectArrayElement(env, mateArr, s);
      const long e = o ? end_index(jni::GetIntField(env, o, mRid), jni::GetIntField(env, o, mPid)) : -1;
      jobject a = e >= 0 ? jni::GetObjectField(env, o, mAln) : nullptr;
      if (!a) { throw_runtime(env, "bPSW: mateSWJNI: malformed MateSWType"); return nullptr; }
      bpsw_alnreg_t& r = tmp[(size_t)s];
      r.rb = jni::GetLongField(env, a, rf.rBeg); r.re = jni::GetLongField(env, a, rf.rEnd);
      r.qb = jni::GetIntField(env, a, rf.qBeg); r.qe = jni::GetIntField(env, a, rf.qEnd);
      r.score = jni::GetIntField(env, a, rf.score); r.truesc = jni::GetIntField(env, a, rf.trueScore);
      r.sub = jni::GetIntField(env, a, rf.sub); r.csub = jni::GetIntField(env, a, rf.csub);
      r.sub_n = jni::GetIntField(env, a, rf.subNum); r.w = jni::GetIntField(env, a, rf.width);
      r.seedcov = jni::GetIntField(env, a, rf.seedCov); r.secondary = jni::GetIntField(env, a, rf.secondary);
      r.hash = (uint64_t)jni::GetLongField(env, a, rf.hash);
      where[(size_t)s] = e;
      ++reg_cnt[(size_t)e];
      jni::DeleteLocalRef(env, a);
      jni::DeleteLocalRef(env, o);
    }
    std::vector<int64_t>& at = ms.at;
    at.assign(ends + 1, 0);
    for (size_t e = 0; e < ends; ++e) at[e + 1] = at[e] + reg_cnt[e];
    regs.resize((size_t)n);
    for (jsize s = 0; s < n; ++s) regs[(size_t)at[(size_t)where[(size_t)s]]++] = tmp[(size_t)s];
  }
  std::vector<int64_t>&ref_rb = ms.ref_rb, &ref_re = ms.ref_re, &ref_len = ms.ref_len, &ref_off = ms.ref_off;
  size_t coord_windows = 0, byte_windows = 0;
  {  // ---- refSizeArray + RefSWType[] (native/jni_mate_sw.c:352-518) ----
    if (jni::GetArrayLength(env, refSizeArr) < (jsize)ends) { throw_runtime(env, "bPSW: mateSWJNI: refSizeArray too short"); return nullptr; }
    if (ends) jni::GetIntArrayRegion(env, refSizeArr, 0, (jsize)ends, ref_cnt.data());
    std::vector<int64_t>& base = ms.base;
    base.assign(ends + 1, 0);
    for (size_t e = 0; e < ends; ++e) {
      if (ref_cnt[e] < 0) { throw_runtime(env, "bPSW: mateSWJNI: negative refSizeArray entry"); return nullptr; }
      base[e + 1] = base[e] + ref_cnt[e];
    }
    const size_t rows = (size_t)base[ends];
    ref_rb.assign(4 * rows, -1); ref_re.assign(4 * rows, -1); ref_len.assign(4 * rows, 0); ref_off.assign(4 * rows, 0);
    jfieldID rRid = ids->rRid, rPid = ids->rPid, rReg = ids->rReg, rB = ids->rB, rE = ids->rE, rL = ids->rL;
    const jfieldID* rRef = ids->rRef;
    const jsize n = jni::GetArrayLength(env, refArr);
    for (jsize s = 0; s < n; ++s) {
      if (jni::PushLocalFrame(env, 16) != JNI_OK) return nullptr;
      jobject o = jni::GetObjectArrayElement(env, refArr, s);
      const long e = o ? end_index(jni::GetIntField(env, o, rRid), jni::GetIntField(env, o, rPid)) : -1;
      const jint j = o ? jni::GetIntField(env, o, rReg) : -1;
      if (e < 0 || j < 0 || j >= ref_cnt[(size_t)e]) { jni::PopLocalFrame(env, nullptr); throw_runtime(env, "bPSW: mateSWJNI: RefSWType index outside refSizeArray"); return nullptr; }
      const size_t x = 4 * (size_t)(base[(size_t)e] + j);
      jlongArray ab = (jlongArray)jni::GetObjectField(env, o, rB), ae = (jlongArray)jni::GetObjectField(env, o, rE);
      jlongArray al = (jlongArray)jni::GetObjectField(env, o, rL);
      if (!ab || !ae || !al || jni::GetArrayLength(env, ab) < 4 || jni::GetArrayLength(env, ae) < 4 || jni::GetArrayLength(env, al) < 4) {
        jni::PopLocalFrame(env, nullptr); throw_runtime(env, "bPSW: mateSWJNI: RefSWType arrays must hold 4 longs"); return nullptr;
      }
      jni::GetLongArrayRegion(env, ab, 0, 4, (jlong*)&ref_rb[x]);
      jni::GetLongArrayRegion(env, ae, 0, 4, (jlong*)&ref_re[x]);
      jni::GetLongArrayRegion(env, al, 0, 4, (jlong*)&ref_len[x]);
      for (int r = 0; r < 4; ++r) {
        if (ref_rb[x + r] < 0 && ref_re[x + r] < 0) continue;  // failed orientation: rBeg=rEnd=-1, ref=null (MemSamPe.scala:1863-1868)
        jbyteArray bytes = (jbyteArray)jni::GetObjectField(env, o, rRef[r]);
        if (!bytes && ref_len[x + r] != 0) { ++coord_windows; continue; }  // named by (rBeg, rEnd) only: read from the device-resident reference
        ++byte_windows;
        if (ref_len[x + r] <= 0) continue;          // bnsGetSeq returned nothing (window bridging the strands)
        int32_t got = 0;
        read_bytes(env, bytes, ref_pool, &ref_off[x + r], &got);
        if (got < ref_len[x + r]) { jni::PopLocalFrame(env, nullptr); throw_runtime(env, "bPSW: mateSWJNI: reference window shorter than lenArray"); return nullptr; }
      }
      jni::PopLocalFrame(env, nullptr);
    }
  }
  if (seq_pool.n == 0) memset(seq_pool.grow(16), 0, 16);
  if (ref_pool.n == 0) memset(ref_pool.grow(16), 0, 16);
  g.seq_len = seq_len.data(); g.seq_off = seq_off.data(); g.seq_pool = seq_pool.p; g.seq_pool_bytes = seq_pool.n;
  g.reg_cnt = reg_cnt.data(); g.regs = regs.data(); g.ref_cnt = ref_cnt.data();
  g.ref_rb = ref_rb.data(); g.ref_re = ref_re.data(); g.ref_len = ref_len.data(); g.ref_off = ref_off.data();
  g.ref_pool = ref_pool.p; g.ref_pool_bytes = ref_pool.n;
  if (coord_windows > 0) {
    if (byte_windows > 0) { throw_runtime(env, "bPSW: mateSWJNI: RefSWType windows must all carry bytes or all be coordinates"); return nullptr; }
    g.ref_pool = nullptr; g.ref_pool_bytes = 0; g.ref_len = nullptr; g.ref_off = nullptr;
  }

  bpsw_ctx_t* ctx = thread_context(env);
  if (!ctx) { throw_runtime(env, std::string("bPSW: no usable HIP device: ") + bpsw_last_error()); return nullptr; }
  std::vector<int32_t>& out_cnt = ms.out_cnt;
  std::vector<bpsw_alnreg_t>& out = ms.out;
  out_cnt.resize(ends ? ends : 1);
  out.resize(regs.size() + 4 * ref_rb.size() / 4 + 16);
  int64_t total = 0;
  const double t1 = now_us();
  const char* compat = getenv("BPSW_MATESW_COMPAT");
  const int mode = (compat && strcmp(compat, "scala") == 0) ? BPSW_RESCUE_SCALA : BPSW_RESCUE_C;
  int rc = bpsw_matesw_group(ctx, &opt, &g, mode, out_cnt.data(), out.data(), (int64_t)out.size(), &total);
  if (rc == BPSW_ERR_CAPACITY) {
    out.resize((size_t)total);
    rc = bpsw_matesw_group(ctx, &opt, &g, mode, out_cnt.data(), out.data(), (int64_t)out.size(), &total);
  }
  if (rc != BPSW_OK) { throw_runtime(env, std::string("bPSW: mateSWJNI: ") + bpsw_last_error()); return nullptr; }

  // ---- result: MateSWType[] in (k, i, rank) order (native/jni_mate_sw.c:548-591) ----
  const double t2 = now_us();
  jobjectArray ret = jni::NewObjectArray(env, (jsize)total, mateCls, nullptr);
  if (!ret) return nullptr;
  int64_t at = 0;
  for (size_t e = 0; e < ends; ++e)
    for (int32_t rank = 0; rank < out_cnt[e]; ++rank, ++at) {
      if (jni::PushLocalFrame(env, 8) != JNI_OK) return nullptr;
      const bpsw_alnreg_t& r = out[(size_t)at];
      jobject m = jni::AllocObject(env, mateCls), a = jni::AllocObject(env, regCls);
      if (!m || !a) { jni::PopLocalFrame(env, nullptr); return nullptr; }
      jni::SetIntField(env, m, mRid, (jint)(e >> 1)); jni::SetIntField(env, m, mPid, (jint)(e & 1)); jni::SetIntField(env, m, mReg, rank);
      jni::SetLongField(env, a, rf.rBeg, r.rb); jni::SetLongField(env, a, rf.rEnd, r.re);
      jni::SetIntField(env, a, rf.qBeg, r.qb); jni::SetIntField(env, a, rf.qEnd, r.qe);
      jni::SetIntField(env, a, rf.score, r.score); jni::SetIntField(env, a, rf.trueScore, r.truesc);
      jni::SetIntField(env, a, rf.sub, r.sub); jni::SetIntField(env, a, rf.csub, r.csub);
      jni::SetIntField(env, a, rf.subNum, r.sub_n); jni::SetIntField(env, a, rf.width, r.w);
      jni::SetIntField(env, a, rf.seedCov, r.seedcov); jni::SetIntField(env, a, rf.secondary, r.secondary);
      jni::SetLongField(env, a, rf.hash, (jlong)r.hash);
      jni::SetObjectField(env, m, mAln, a);
      jni::SetObjectArrayElement(env, ret, (jsize)at, m);
      jni::PopLocalFrame(env, nullptr);
    }
  t_times = {t1 - t0, t2 - t1, now_us() - t2, (double)total};
  t_mate_path = 2;
  return ret;
  } catch (const std::exception& e) {  // nothing C++ may unwind into the JVM (std::bad_alloc of a scratch vector, ...)
    throw_runtime(env, std::string("bPSW: mateSWJNI: ") + e.what());
    return nullptr;
  }
}

// ---- boundary 1 with flat arrays (round 4) -------------------------------------------------------------------------------------
// mateSWJNI's contract (native/jni_mate_sw.c:239-518, 560-591) costs one GetObjectArrayElement and ~13 Get<Type>Field per region on
// the way in and an AllocObject pair per region on the way out: 96 % of a 4 096-pair call (breakdown.jni_shim_fake_env).  This entry
// takes what memSamPeGroupJNIPrepare (worker2/MemSamPe.scala:1895-2000) holds in its loops as primitive arrays -- the same values in
// the same (k, i, j) order, no objects -- and returns one long[]; the Scala edit is in INTEGRATION.md section 1e.
// Scala side (jni/MateSWJNI.scala):
//   @native def mateSWFlatJNI(optInts: Array[Int], maskLevelRedun: Float, mat: Array[Byte], pacLen: Long, pes: Array[Double],
//                             groupSize: Int, seqLen: Array[Int], seqs: Array[Byte], regCnt: Array[Int], regLongs: Array[Long],
//                             regInts: Array[Int], refCnt: Array[Int], refRb: Array[Long], refRe: Array[Long], refLen: Array[Long],
//                             refBytes: Array[Byte]): Array[Long]
// optInts  = (a, b, oDel, eDel, oIns, eIns, penUnpaired, penClip5, penClip3, w, zdrop, T, flag, minSeedLen, maxIns, maxMatesw);
// pes      = per orientation (low, high, failed, avg, std): 20 doubles;
// seqLen / regCnt / refCnt are indexed 2k+i (refCnt = refSizeArray, MemSamPe.scala:1944-1947); seqs = the reads' codes back to back;
// regLongs = (rBeg, rEnd, hash), regInts = (qBeg, qEnd, score, trueScore, sub, csub, subNum, width, seedCov, secondary) per region in
//            (k, i, j) order; refRb / refRe / refLen hold 4 entries (the orientations) per (k, i, j < refCnt) row in the same order
//            (-1, -1, 0 = failed orientation, MemSamPe.scala:1863-1868); refBytes = the windows of positive length back to back in
//            that order.  refLen == null and refBytes == null: the windows are named by (rBeg, rEnd) alone and read from the
//            reference loaded with loadPacJNI -- then nothing but the reads' own bytes travels.
// Returns long[2 G + 8 R']: the region counts per end, then per region after the rescue, in (k, i, rank) order, 8 longs:
//   rBeg, rEnd, hash, qBeg | qEnd << 32, score | trueScore << 32, sub | csub << 32, subNum | width << 32, seedCov | secondary << 32
// (the low halves are the 32 low bits of the first field).
JNIEXPORT jlongArray JNICALL Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_mateSWFlatJNI(
    JNIEnv* env, jobject, jintArray optInts, jfloat maskLevelRedun, jbyteArray matArr, jlong pacLen, jdoubleArray pesArr, jint groupSize,
    jintArray seqLenArr, jbyteArray seqsArr, jintArray regCntArr, jlongArray regLongsArr, jintArray regIntsArr, jintArray refCntArr,
    jlongArray refRbArr, jlongArray refReArr, jlongArray refLenArr, jbyteArray refBytesArr) {
  try {
  if (groupSize < 0 || groupSize > 0x3fffffff) { throw_runtime(env, "bPSW: mateSWFlatJNI: groupSize out of range"); return nullptr; }
  const jsize ends_j = 2 * groupSize;
  if (!optInts || !matArr || !pesArr || !seqLenArr || !seqsArr || !regCntArr || !regLongsArr || !regIntsArr || !refCntArr || !refRbArr ||
      !refReArr || groupSize < 0 || (refLenArr == nullptr) != (refBytesArr == nullptr) || jni::GetArrayLength(env, optInts) < 16 ||
      jni::GetArrayLength(env, matArr) < 25 || jni::GetArrayLength(env, pesArr) < 20 || jni::GetArrayLength(env, seqLenArr) < ends_j ||
      jni::GetArrayLength(env, regCntArr) < ends_j || jni::GetArrayLength(env, refCntArr) < ends_j) {
    throw_runtime(env, "bPSW: mateSWFlatJNI: bad arguments");
    return nullptr;
  }
  const double t0 = now_us();
  bpsw_opt_t opt;
  bpsw_opt_default(&opt);
  {
    jint oi[16];
    jni::GetIntArrayRegion(env, optInts, 0, 16, oi);
    int32_t* dst[16] = {&opt.a, &opt.b, &opt.o_del, &opt.e_del, &opt.o_ins, &opt.e_ins, &opt.pen_unpaired, &opt.pen_clip5,
                        &opt.pen_clip3, &opt.w, &opt.zdrop, &opt.T, &opt.flag, &opt.min_seed_len, &opt.max_ins, &opt.max_matesw};
    for (int i = 0; i < 16; ++i) *dst[i] = oi[i];
    opt.mask_level_redun = maskLevelRedun;
    jni::GetByteArrayRegion(env, matArr, 0, 25, reinterpret_cast<jbyte*>(opt.mat));
  }
  bpsw_rescue_group_t g;
  memset(&g, 0, sizeof g);
  g.group_size = groupSize;
  g.l_pac = pacLen;
  {
    jdouble pe[20];
    jni::GetDoubleArrayRegion(env, pesArr, 0, 20, pe);
    for (int r = 0; r < 4; ++r) {
      g.pes[r].low = (int32_t)pe[5 * r]; g.pes[r].high = (int32_t)pe[5 * r + 1]; g.pes[r].failed = (int32_t)pe[5 * r + 2];
      g.pes[r].avg = pe[5 * r + 3]; g.pes[r].std = pe[5 * r + 4];
    }
  }
  const size_t ends = (size_t)ends_j;
  MateScratch& ms = t_ms;
  std::vector<int32_t>&seq_len = ms.seq_len, &reg_cnt = ms.reg_cnt, &ref_cnt = ms.ref_cnt;
  std::vector<int64_t>&seq_off = ms.seq_off, &ref_rb = ms.ref_rb, &ref_re = ms.ref_re, &ref_len = ms.ref_len, &ref_off = ms.ref_off;
  seq_len.resize(ends + 1); reg_cnt.resize(ends + 1); ref_cnt.resize(ends + 1); seq_off.resize(ends + 1);
  if (ends) {
    jni::GetIntArrayRegion(env, seqLenArr, 0, ends_j, seq_len.data());
    jni::GetIntArrayRegion(env, regCntArr, 0, ends_j, reg_cnt.data());
    jni::GetIntArrayRegion(env, refCntArr, 0, ends_j, ref_cnt.data());
  }
  int64_t seq_bytes = 0, n_regs = 0, rows = 0;
  for (size_t e = 0; e < ends; ++e) {
    if (seq_len[e] < 0 || reg_cnt[e] < 0 || ref_cnt[e] < 0) { throw_runtime(env, "bPSW: mateSWFlatJNI: negative length or count"); return nullptr; }
    seq_off[e] = seq_bytes; seq_bytes += seq_len[e]; n_regs += reg_cnt[e]; rows += ref_cnt[e];
  }
  if ((int64_t)jni::GetArrayLength(env, seqsArr) < seq_bytes || (int64_t)jni::GetArrayLength(env, regLongsArr) < 3 * n_regs ||
      (int64_t)jni::GetArrayLength(env, regIntsArr) < 10 * n_regs || (int64_t)jni::GetArrayLength(env, refRbArr) < 4 * rows ||
      (int64_t)jni::GetArrayLength(env, refReArr) < 4 * rows || (refLenArr && (int64_t)jni::GetArrayLength(env, refLenArr) < 4 * rows)) {
    throw_runtime(env, "bPSW: mateSWFlatJNI: an array is shorter than its table says");
    return nullptr;
  }
  BytePool &seq_pool = ms.seq_pool, &ref_pool = ms.ref_pool;
  seq_pool.clear(); ref_pool.clear();
  if (seq_bytes) jni::GetByteArrayRegion(env, seqsArr, 0, (jsize)seq_bytes, reinterpret_cast<jbyte*>(seq_pool.grow((size_t)seq_bytes)));
  memset(seq_pool.grow(16), 0, 16);
  std::vector<bpsw_alnreg_t>& regs = ms.regs;
  regs.resize((size_t)n_regs + 1);
  {
    std::vector<int64_t>& rl = ms.at;    // (scratch vectors of the object-array entry, reused)
    std::vector<int32_t>& ri = ms.out_cnt;
    rl.resize((size_t)(3 * n_regs) + 1); ri.resize((size_t)(10 * n_regs) + 1);
    if (n_regs) {
      jni::GetLongArrayRegion(env, regLongsArr, 0, (jsize)(3 * n_regs), reinterpret_cast<jlong*>(rl.data()));
      jni::GetIntArrayRegion(env, regIntsArr, 0, (jsize)(10 * n_regs), ri.data());
    }
    for (int64_t j = 0; j < n_regs; ++j) {
      bpsw_alnreg_t& a = regs[(size_t)j];
      const int32_t* v = ri.data() + 10 * j;
      a.rb = rl[(size_t)(3 * j)]; a.re = rl[(size_t)(3 * j + 1)]; a.hash = (uint64_t)rl[(size_t)(3 * j + 2)];
      a.qb = v[0]; a.qe = v[1]; a.score = v[2]; a.truesc = v[3]; a.sub = v[4]; a.csub = v[5]; a.sub_n = v[6]; a.w = v[7]; a.seedcov = v[8];
      a.secondary = v[9];
    }
  }
  ref_rb.resize((size_t)(4 * rows) + 1); ref_re.resize((size_t)(4 * rows) + 1);
  if (rows) {
    jni::GetLongArrayRegion(env, refRbArr, 0, (jsize)(4 * rows), reinterpret_cast<jlong*>(ref_rb.data()));
    jni::GetLongArrayRegion(env, refReArr, 0, (jsize)(4 * rows), reinterpret_cast<jlong*>(ref_re.data()));
  }
  if (refLenArr) {
    ref_len.resize((size_t)(4 * rows) + 1); ref_off.resize((size_t)(4 * rows) + 1);
    if (rows) jni::GetLongArrayRegion(env, refLenArr, 0, (jsize)(4 * rows), reinterpret_cast<jlong*>(ref_len.data()));
    int64_t ref_bytes = 0;
    for (int64_t x = 0; x < 4 * rows; ++x) {
      if (ref_len[(size_t)x] < 0) { throw_runtime(env, "bPSW: mateSWFlatJNI: negative window length"); return nullptr; }
      ref_off[(size_t)x] = ref_bytes;
      ref_bytes += ref_len[(size_t)x];
    }
    if ((int64_t)jni::GetArrayLength(env, refBytesArr) < ref_bytes) { throw_runtime(env, "bPSW: mateSWFlatJNI: refBytes shorter than refLen says"); return nullptr; }
    if (ref_bytes) jni::GetByteArrayRegion(env, refBytesArr, 0, (jsize)ref_bytes, reinterpret_cast<jbyte*>(ref_pool.grow((size_t)ref_bytes)));
    memset(ref_pool.grow(16), 0, 16);
    g.ref_len = ref_len.data(); g.ref_off = ref_off.data(); g.ref_pool = ref_pool.p; g.ref_pool_bytes = ref_pool.n;
  }
  g.seq_len = seq_len.data(); g.seq_off = seq_off.data(); g.seq_pool = seq_pool.p; g.seq_pool_bytes = seq_pool.n;
  g.reg_cnt = reg_cnt.data(); g.regs = regs.data(); g.ref_cnt = ref_cnt.data(); g.ref_rb = ref_rb.data(); g.ref_re = ref_re.data();

  bpsw_ctx_t* ctx = thread_context(env);
  if (!ctx) { throw_runtime(env, std::string("bPSW: no usable HIP device: ") + bpsw_last_error()); return nullptr; }
  std::vector<int32_t>& out_cnt = ms.tmp_cnt;
  std::vector<bpsw_alnreg_t>& out = ms.out;
  out_cnt.resize(ends ? ends : 1);
  out.resize((size_t)n_regs + (size_t)rows + 16);
  int64_t total = 0;
  const double t1 = now_us();
  const char* compat = getenv("BPSW_MATESW_COMPAT");
  const int mode = (compat && strcmp(compat, "scala") == 0) ? BPSW_RESCUE_SCALA : BPSW_RESCUE_C;
  int rc = bpsw_matesw_group(ctx, &opt, &g, mode, out_cnt.data(), out.data(), (int64_t)out.size(), &total);
  if (rc == BPSW_ERR_CAPACITY) {
    out.resize((size_t)total);
    rc = bpsw_matesw_group(ctx, &opt, &g, mode, out_cnt.data(), out.data(), (int64_t)out.size(), &total);
  }
  if (rc != BPSW_OK) { throw_runtime(env, std::string("bPSW: mateSWFlatJNI: ") + bpsw_last_error()); return nullptr; }

  const double t2 = now_us();
  const int64_t n_out = (int64_t)ends + 8 * total;
  if (n_out > 0x7fffffffLL) { throw_runtime(env, "bPSW: mateSWFlatJNI: result exceeds a Java array; use a smaller group"); return nullptr; }
  std::vector<int64_t>& res = ms.base;
  res.resize((size_t)n_out + 1);
  for (size_t e = 0; e < ends; ++e) res[e] = out_cnt[e];
  const auto pack = [](int32_t lo, int32_t hi) { return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo); };
  for (int64_t j = 0; j < total; ++j) {
    const bpsw_alnreg_t& r = out[(size_t)j];
    int64_t* o = res.data() + ends + 8 * j;
    o[0] = r.rb; o[1] = r.re; o[2] = (int64_t)r.hash; o[3] = pack(r.qb, r.qe); o[4] = pack(r.score, r.truesc); o[5] = pack(r.sub, r.csub);
    o[6] = pack(r.sub_n, r.w); o[7] = pack(r.seedcov, r.secondary);
  }
  jlongArray ret = jni::NewLongArray(env, (jsize)n_out);
  if (!ret) return nullptr;  // OutOfMemoryError already pending
  if (n_out) jni::SetLongArrayRegion(env, ret, 0, (jsize)n_out, reinterpret_cast<const jlong*>(res.data()));
  t_times = {t1 - t0, t2 - t1, now_us() - t2, (double)total};
  return ret;
  } catch (const std::exception& e) {  // nothing C++ may unwind into the JVM (std::bad_alloc of a scratch vector, ...)
    throw_runtime(env, std::string("bPSW: mateSWFlatJNI: ") + e.what());
    return nullptr;
  }
}

// ---- reference on the device -------------------------------------------------------------
// Scala side (one line in jni/MateSWJNI.scala):  @native def loadPacJNI(pac: Array[Byte], pacLen: Long): Int
// Call once per executor JVM before the first mateSWJNI; returns the number of devices that now hold the reference.
JNIEXPORT jint JNICALL Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_loadPacJNI(JNIEnv* env, jobject, jbyteArray pacArr, jlong pacLen) {
  try {
  if (!pacArr || pacLen < 1) { throw_runtime(env, "bPSW: loadPacJNI: bad arguments"); return 0; }
  const jsize bytes = jni::GetArrayLength(env, pacArr);
  if ((int64_t)bytes < (pacLen + 3) / 4) { throw_runtime(env, "bPSW: loadPacJNI: pac shorter than (pacLen+3)/4 bytes"); return 0; }
  std::vector<uint8_t> pac((size_t)bytes);
  jni::GetByteArrayRegion(env, pacArr, 0, bytes, reinterpret_cast<jbyte*>(pac.data()));
  const int ndev = bpsw_device_count();
  if (ndev <= 0) { throw_runtime(env, std::string("bPSW: no usable HIP device: ") + bpsw_last_error()); return 0; }
  int loaded = 0;
  for (int d = 0; d < ndev; ++d) {  // any task thread may land on any device (partition -> device)
    bpsw_ctx_t* c = nullptr;
    int rc = bpsw_create(d, &c);
    if (rc == BPSW_OK) rc = bpsw_ref_load(c, pac.data(), pacLen);
    if (c) bpsw_destroy(c);
    if (rc != BPSW_OK) { throw_runtime(env, std::string("bPSW: loadPacJNI: ") + bpsw_last_error()); return loaded; }
    ++loaded;
  }
  return loaded;
  } catch (const std::exception& e) {  // nothing C++ may unwind into the JVM (std::bad_alloc of a scratch vector, ...)
    throw_runtime(env, std::string("bPSW: loadPacJNI: ") + e.what());
    return 0;
  }
}

// ---- the whole memChainToAlnBatched round loop in one call -------------------------------------
// Scala side (jni/SWExtendFPGAJNI.scala, next to swExtendFPGAJNI):
//   @native def chainToAlnJNI(optInts: Array[Int], mat: Array[Byte], readLen: Array[Int], reads: Array[Byte],
//                             chainCnt: Array[Int], seedCnt: Array[Int], seedRBeg: Array[Long], seedQBeg: Array[Int],
//                             seedLen: Array[Int]): Array[Long]
// optInts = (a, b, oDel, eDel, oIns, eIns, penClip5, penClip3, w, zdrop); reads = the reads back to back (codes 0..4).
// Result: n longs (regions per read) followed by 8 longs per region in (read, creation) order:
//   rBeg, rEnd, qBeg, qEnd, score, trueScore, width, seedCov     (what memChainToAlnBatched leaves in regArrays).
// Needs loadPacJNI first.  BPSW_ZDROP=bwa selects the BWA z-drop parse, default is the Scala one (SWUtil.scala:194-199).
JNIEXPORT jlongArray JNICALL Java_cs_ucla_edu_bwaspark_jni_SWExtendFPGAJNI_chainToAlnJNI(
    JNIEnv* env, jobject, jintArray optInts, jbyteArray matArr, jintArray readLenArr, jbyteArray readsArr, jintArray chainCntArr,
    jintArray seedCntArr, jlongArray seedRBegArr, jintArray seedQBegArr, jintArray seedLenArr) {
  try {
  if (!optInts || !matArr || !readLenArr || !readsArr || !chainCntArr || !seedCntArr || !seedRBegArr || !seedQBegArr || !seedLenArr ||
      jni::GetArrayLength(env, optInts) < 10 || jni::GetArrayLength(env, matArr) < 25) {
    throw_runtime(env, "bPSW: chainToAlnJNI: bad arguments");
    return nullptr;
  }
  bpsw_opt_t opt;
  bpsw_opt_default(&opt);
  jint oi[10];
  jni::GetIntArrayRegion(env, optInts, 0, 10, oi);
  opt.a = oi[0]; opt.b = oi[1]; opt.o_del = oi[2]; opt.e_del = oi[3]; opt.o_ins = oi[4]; opt.e_ins = oi[5];
  opt.pen_clip5 = oi[6]; opt.pen_clip3 = oi[7]; opt.w = oi[8]; opt.zdrop = oi[9];
  jni::GetByteArrayRegion(env, matArr, 0, 25, reinterpret_cast<jbyte*>(opt.mat));
  const jsize n = jni::GetArrayLength(env, readLenArr);
  if (jni::GetArrayLength(env, chainCntArr) < n) { throw_runtime(env, "bPSW: chainToAlnJNI: chainCnt shorter than readLen"); return nullptr; }
  std::vector<int32_t> read_len((size_t)n), chain_cnt((size_t)n);
  if (n) { jni::GetIntArrayRegion(env, readLenArr, 0, n, read_len.data()); jni::GetIntArrayRegion(env, chainCntArr, 0, n, chain_cnt.data()); }
  std::vector<int64_t> read_off((size_t)n);
  int64_t at = 0, nchains = 0;
  for (jsize r = 0; r < n; ++r) {
    if (read_len[(size_t)r] < 0 || chain_cnt[(size_t)r] < 0) { throw_runtime(env, "bPSW: chainToAlnJNI: negative length or count"); return nullptr; }
    read_off[(size_t)r] = at; at += read_len[(size_t)r]; nchains += chain_cnt[(size_t)r];
  }
  const jsize pool_bytes = jni::GetArrayLength(env, readsArr);
  if ((int64_t)pool_bytes < at || (int64_t)jni::GetArrayLength(env, seedCntArr) < nchains) { throw_runtime(env, "bPSW: chainToAlnJNI: reads or seedCnt too short"); return nullptr; }
  std::vector<uint8_t> pool((size_t)pool_bytes + 16);
  if (pool_bytes) jni::GetByteArrayRegion(env, readsArr, 0, pool_bytes, reinterpret_cast<jbyte*>(pool.data()));
  std::vector<int32_t> seed_cnt((size_t)nchains + 1);
  if (nchains) jni::GetIntArrayRegion(env, seedCntArr, 0, (jsize)nchains, seed_cnt.data());
  int64_t nseeds = 0;
  for (int64_t k = 0; k < nchains; ++k) {
    if (seed_cnt[(size_t)k] < 0) { throw_runtime(env, "bPSW: chainToAlnJNI: negative seed count"); return nullptr; }
    nseeds += seed_cnt[(size_t)k];
  }
  if ((int64_t)jni::GetArrayLength(env, seedRBegArr) < nseeds || (int64_t)jni::GetArrayLength(env, seedQBegArr) < nseeds ||
      (int64_t)jni::GetArrayLength(env, seedLenArr) < nseeds) { throw_runtime(env, "bPSW: chainToAlnJNI: seed arrays too short"); return nullptr; }
  std::vector<int64_t> seed_rbeg((size_t)nseeds + 1);
  std::vector<int32_t> seed_qbeg((size_t)nseeds + 1), seed_len((size_t)nseeds + 1);
  if (nseeds) {
    jni::GetLongArrayRegion(env, seedRBegArr, 0, (jsize)nseeds, (jlong*)seed_rbeg.data());
    jni::GetIntArrayRegion(env, seedQBegArr, 0, (jsize)nseeds, seed_qbeg.data());
    jni::GetIntArrayRegion(env, seedLenArr, 0, (jsize)nseeds, seed_len.data());
  }
  bpsw_chains_t b;
  memset(&b, 0, sizeof b);
  b.n_reads = n; b.read_len = read_len.data(); b.read_off = read_off.data(); b.read_pool = pool.data(); b.read_pool_bytes = (size_t)pool_bytes;
  b.chain_cnt = chain_cnt.data(); b.seed_cnt = seed_cnt.data(); b.seed_rbeg = seed_rbeg.data(); b.seed_qbeg = seed_qbeg.data(); b.seed_len = seed_len.data();
  bpsw_ctx_t* ctx = thread_context(env);
  if (!ctx) { throw_runtime(env, std::string("bPSW: no usable HIP device: ") + bpsw_last_error()); return nullptr; }
  std::vector<int32_t> out_cnt((size_t)n + 1);
  std::vector<bpsw_alnreg_t> out((size_t)nseeds + 1);
  int64_t total = 0;
  const char* zd = getenv("BPSW_ZDROP");
  const int zmode = (zd && strcmp(zd, "bwa") == 0) ? BPSW_ZDROP_BWA : BPSW_ZDROP_SCALA;
  if (bpsw_chain2aln_batch(ctx, &opt, &b, zmode, 0, out_cnt.data(), out.data(), (int64_t)out.size(), &total) != BPSW_OK) {
    throw_runtime(env, std::string("bPSW: chainToAlnJNI: ") + bpsw_last_error());
    return nullptr;
  }
  std::vector<jlong> flat((size_t)n + 8 * (size_t)total);
  for (jsize r = 0; r < n; ++r) flat[(size_t)r] = out_cnt[(size_t)r];
  for (int64_t k = 0; k < total; ++k) {
    jlong* o = flat.data() + (size_t)n + 8 * (size_t)k;
    const bpsw_alnreg_t& g = out[(size_t)k];
    o[0] = g.rb; o[1] = g.re; o[2] = g.qb; o[3] = g.qe; o[4] = g.score; o[5] = g.truesc; o[6] = g.w; o[7] = g.seedcov;
  }
  jlongArray ret = jni::NewLongArray(env, (jsize)flat.size());
  if (!ret) return nullptr;  // OutOfMemoryError already pending
  if (!flat.empty()) jni::SetLongArrayRegion(env, ret, 0, (jsize)flat.size(), flat.data());
  return ret;
  } catch (const std::exception& e) {  // nothing C++ may unwind into the JVM (std::bad_alloc of a scratch vector, ...)
    throw_runtime(env, std::string("bPSW: chainToAlnJNI: ") + e.what());
    return nullptr;
  }
}

// ---- worker2's tail -----------------------------------------------------------------------------
// Scala side (jni/MateSWJNI.scala):
//   @native def loadBnsJNI(offset: Array[Long], len: Array[Int], names: Array[Byte]): Int     // names NUL-terminated, back to back
// Call once per executor JVM after loadPacJNI (bns.anns(i).offset / .len / .name); returns the number of devices loaded.
JNIEXPORT jint JNICALL Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_loadBnsJNI(JNIEnv* env, jobject, jlongArray offArr, jintArray lenArr,
                                                                          jbyteArray namesArr) {
  try {
  if (!offArr || !lenArr) { throw_runtime(env, "bPSW: loadBnsJNI: bad arguments"); return 0; }
  const jsize n = jni::GetArrayLength(env, offArr);
  if (n < 1 || jni::GetArrayLength(env, lenArr) != n) { throw_runtime(env, "bPSW: loadBnsJNI: offset and len must have one entry per contig"); return 0; }
  std::vector<jlong> off((size_t)n);
  std::vector<jint> len((size_t)n);
  jni::GetLongArrayRegion(env, offArr, 0, n, off.data());
  jni::GetIntArrayRegion(env, lenArr, 0, n, len.data());
  std::vector<char> names;
  if (namesArr) {
    const jsize nb = jni::GetArrayLength(env, namesArr);
    names.resize((size_t)nb + 1, 0);
    if (nb > 0) jni::GetByteArrayRegion(env, namesArr, 0, nb, reinterpret_cast<jbyte*>(names.data()));
    jsize zeros = 0;
    for (jsize i = 0; i < nb; ++i) zeros += names[(size_t)i] == 0;
    if (zeros < n) { throw_runtime(env, "bPSW: loadBnsJNI: fewer NUL-terminated names than contigs"); return 0; }
  }
  const int ndev = bpsw_device_count();
  if (ndev <= 0) { throw_runtime(env, std::string("bPSW: no usable HIP device: ") + bpsw_last_error()); return 0; }
  int loaded = 0;
  for (int d = 0; d < ndev; ++d) {
    bpsw_ctx_t* c = nullptr;
    int rc = bpsw_create(d, &c);
    if (rc == BPSW_OK) rc = bpsw_bns_load(c, n, reinterpret_cast<const int64_t*>(off.data()), len.data(), namesArr ? names.data() : nullptr);
    if (c) bpsw_destroy(c);
    if (rc != BPSW_OK) { throw_runtime(env, std::string("bPSW: loadBnsJNI: ") + bpsw_last_error()); return loaded; }
    ++loaded;
  }
  return loaded;
  } catch (const std::exception& e) {  // nothing C++ may unwind into the JVM (std::bad_alloc of a scratch vector, ...)
    throw_runtime(env, std::string("bPSW: loadBnsJNI: ") + e.what());
    return 0;
  }
}

// Scala side (jni/MateSWJNI.scala); replaces the body of memSamPeGroupRest (worker2/MemSamPe.scala:1390-1612):
//   @native def samPeTailJNI(optInts: Array[Int], reals: Array[Double], mat: Array[Byte], id: Long, readLen: Array[Int],
//                            reads: Array[Byte], quals: Array[Byte], nameLen: Array[Int], names: Array[Byte], regCnt: Array[Int],
//                            regLongs: Array[Long], regInts: Array[Int], outOff: Array[Long]): Array[Byte]
// optInts = (a, b, oDel, eDel, oIns, eIns, penUnpaired, w, T, flag, minSeedLen, mapQCoefFac);
// reals   = (maskLevel, mapQCoefLen) then per orientation (low, high, failed, avg, std);
// readLen / regCnt are indexed 2k+i; reads / quals / names back to back (quals may be null: '*');
// regLongs = (rBeg, rEnd) and regInts = (qBeg, qEnd, score, trueScore, sub, csub, subNum, width, seedCov, secondary) per region,
// in (k, i, j) order as mateSWJNI returns them.  Returns the SAM text of the group; the text of read 2k+i is
// [outOff(2k+i), outOff(2k+i+1)) (outOff has 2*groupSize + 1 entries).  BPSW_TAIL_COMPAT=c selects the C flavour.
}  // extern "C" (the tail's marshalling is shared by three entries: C++ linkage)

namespace {

// one group of samPeTailJNI's arguments in flat native buffers; lives on the stack of the synchronous entry, on the heap behind a ticket
struct TailCall {
  bpsw_opt_t opt;
  bpsw_tail_opt_t topt;
  bpsw_pairs_t g;
  std::vector<int32_t> read_len, reg_cnt;
  std::vector<int64_t> read_off, name_off, out_off;
  std::vector<uint8_t> reads, quals;
  std::vector<char> names, text;
  std::vector<bpsw_alnreg_t> regs;
  jsize n2 = 0;
  // (the asynchronous pair) the pool the group went to and its ticket there
  bpsw_tail_pool_t* pool = nullptr;
  int64_t ticket = 0;
};

// false: a RuntimeException is pending
bool tail_unmarshal(JNIEnv* env, const char* who, jintArray optInts, jdoubleArray realsArr, jbyteArray matArr, jlong id0, jintArray readLenArr,
                    jbyteArray readsArr, jbyteArray qualsArr, jintArray nameLenArr, jbyteArray namesArr, jintArray regCntArr,
                    jlongArray regLongsArr, jintArray regIntsArr, TailCall& tc) {
  const std::string pre = std::string("bPSW: ") + who + ": ";
  if (!optInts || !realsArr || !matArr || !readLenArr || !readsArr || !nameLenArr || !namesArr || !regCntArr || !regLongsArr || !regIntsArr ||
      jni::GetArrayLength(env, optInts) < 12 || jni::GetArrayLength(env, realsArr) < 22 || jni::GetArrayLength(env, matArr) < 25) {
    throw_runtime(env, pre + "bad arguments");
    return false;
  }
  bpsw_opt_t& opt = tc.opt;
  bpsw_tail_opt_t& topt = tc.topt;
  bpsw_opt_default(&opt);
  bpsw_tail_opt_default(&topt);
  jint oi[12];
  jni::GetIntArrayRegion(env, optInts, 0, 12, oi);
  opt.a = oi[0]; opt.b = oi[1]; opt.o_del = oi[2]; opt.e_del = oi[3]; opt.o_ins = oi[4]; opt.e_ins = oi[5]; opt.pen_unpaired = oi[6];
  opt.w = oi[7]; opt.T = oi[8]; opt.flag = oi[9]; opt.min_seed_len = oi[10]; topt.mapq_coef_fac = oi[11];
  jdouble re[22];
  jni::GetDoubleArrayRegion(env, realsArr, 0, 22, re);
  topt.mask_level = (float)re[0]; topt.mapq_coef_len = (float)re[1];
  jni::GetByteArrayRegion(env, matArr, 0, 25, reinterpret_cast<jbyte*>(opt.mat));
  const char* compat = getenv("BPSW_TAIL_COMPAT");
  topt.flavour = (compat && (compat[0] == 'c' || compat[0] == 'C')) ? BPSW_TAIL_C : BPSW_TAIL_SCALA;
  // the read group of the run (samHeader.bwaReadGroupID, FastMap.scala:109-114: fixed by the -R line before any worker starts)
  if (const char* rg = getenv("BPSW_READ_GROUP_ID")) {
    // (the reference writes samHeader.bwaReadGroupID whatever its length: a truncated ID would give RG:Z tags that match no @RG line)
    if (strlen(rg) >= sizeof(topt.rg_id)) { throw_runtime(env, pre + "BPSW_READ_GROUP_ID is longer than 63 characters"); return false; }
    strncpy(topt.rg_id, rg, sizeof(topt.rg_id) - 1);
  }
  const jsize n2 = jni::GetArrayLength(env, readLenArr);
  const jsize G = n2 / 2;
  tc.n2 = n2;
  if ((n2 & 1) || jni::GetArrayLength(env, regCntArr) != n2 || jni::GetArrayLength(env, nameLenArr) != G) {
    throw_runtime(env, pre + "array lengths do not describe groupSize pairs");
    return false;
  }
  std::vector<int32_t> name_len((size_t)G);
  tc.read_len.resize((size_t)n2); tc.reg_cnt.resize((size_t)n2);
  if (n2) { jni::GetIntArrayRegion(env, readLenArr, 0, n2, tc.read_len.data()); jni::GetIntArrayRegion(env, regCntArr, 0, n2, tc.reg_cnt.data()); }
  if (G) jni::GetIntArrayRegion(env, nameLenArr, 0, G, name_len.data());
  tc.read_off.resize((size_t)n2); tc.name_off.assign((size_t)G + 1, 0);
  int64_t at = 0, n_regs = 0;
  for (jsize r = 0; r < n2; ++r) {
    if (tc.read_len[(size_t)r] < 0 || tc.reg_cnt[(size_t)r] < 0) { throw_runtime(env, pre + "negative length or count"); return false; }
    tc.read_off[(size_t)r] = at; at += tc.read_len[(size_t)r]; n_regs += tc.reg_cnt[(size_t)r];
  }
  for (jsize k = 0; k < G; ++k) {
    if (name_len[(size_t)k] < 0) { throw_runtime(env, pre + "negative name length"); return false; }
    tc.name_off[(size_t)k + 1] = tc.name_off[(size_t)k] + name_len[(size_t)k];
  }
  const int64_t names_bytes = tc.name_off[(size_t)G];
  if ((int64_t)jni::GetArrayLength(env, readsArr) < at || (qualsArr && (int64_t)jni::GetArrayLength(env, qualsArr) < at) ||
      (int64_t)jni::GetArrayLength(env, namesArr) < names_bytes || (int64_t)jni::GetArrayLength(env, regLongsArr) < 2 * n_regs ||
      (int64_t)jni::GetArrayLength(env, regIntsArr) < 10 * n_regs) {
    throw_runtime(env, pre + "a pool is shorter than its table says");
    return false;
  }
  tc.reads.resize((size_t)at + 16);
  if (at) jni::GetByteArrayRegion(env, readsArr, 0, (jsize)at, reinterpret_cast<jbyte*>(tc.reads.data()));
  if (qualsArr) { tc.quals.resize((size_t)at + 16); if (at) jni::GetByteArrayRegion(env, qualsArr, 0, (jsize)at, reinterpret_cast<jbyte*>(tc.quals.data())); }
  tc.names.resize((size_t)names_bytes + 1);
  if (names_bytes) jni::GetByteArrayRegion(env, namesArr, 0, (jsize)names_bytes, reinterpret_cast<jbyte*>(tc.names.data()));
  std::vector<jlong> rl((size_t)(2 * n_regs) + 1);
  std::vector<jint> ri((size_t)(10 * n_regs) + 1);
  if (n_regs) { jni::GetLongArrayRegion(env, regLongsArr, 0, (jsize)(2 * n_regs), rl.data()); jni::GetIntArrayRegion(env, regIntsArr, 0, (jsize)(10 * n_regs), ri.data()); }
  tc.regs.resize((size_t)n_regs + 1);
  for (int64_t j = 0; j < n_regs; ++j) {
    bpsw_alnreg_t& a = tc.regs[(size_t)j];
    const jint* v = ri.data() + 10 * j;
    a.rb = rl[(size_t)(2 * j)]; a.re = rl[(size_t)(2 * j + 1)];
    a.qb = v[0]; a.qe = v[1]; a.score = v[2]; a.truesc = v[3]; a.sub = v[4]; a.csub = v[5]; a.sub_n = v[6]; a.w = v[7]; a.seedcov = v[8];
    a.secondary = v[9]; a.hash = 0;
  }
  bpsw_pairs_t& g = tc.g;
  memset(&g, 0, sizeof g);
  g.group_size = G; g.id0 = id0;
  for (int r = 0; r < 4; ++r) {
    g.pes[r].low = (int32_t)re[2 + 5 * r]; g.pes[r].high = (int32_t)re[3 + 5 * r]; g.pes[r].failed = (int32_t)re[4 + 5 * r];
    g.pes[r].avg = re[5 + 5 * r]; g.pes[r].std = re[6 + 5 * r];
  }
  g.read_len = tc.read_len.data(); g.read_off = tc.read_off.data(); g.read_pool = tc.reads.data(); g.qual_pool = qualsArr ? tc.quals.data() : nullptr;
  g.read_pool_bytes = (size_t)at; g.name_off = tc.name_off.data(); g.name_pool = tc.names.data(); g.reg_cnt = tc.reg_cnt.data(); g.regs = tc.regs.data();
  tc.out_off.assign((size_t)n2 + 1, 0);
  tc.text.resize((size_t)G * 1400 + 1024);
  return true;
}

// the group's text as a byte[] + outOff filled; nullptr with an exception pending
jbyteArray tail_result(JNIEnv* env, const char* who, const TailCall& tc, size_t need, jlongArray outOffArr) {
  if (need > 0x7fffffffULL) { throw_runtime(env, std::string("bPSW: ") + who + ": SAM text of the group exceeds 2 GiB; use a smaller group"); return nullptr; }
  jbyteArray ret = jni::NewByteArray(env, (jsize)need);
  if (!ret) return nullptr;  // OutOfMemoryError already pending
  if (need) jni::SetByteArrayRegion(env, ret, 0, (jsize)need, reinterpret_cast<const jbyte*>(tc.text.data()));
  jni::SetLongArrayRegion(env, outOffArr, 0, tc.n2 + 1, reinterpret_cast<const jlong*>(tc.out_off.data()));
  return ret;
}

// The library's tail pools, one per device, made by the first samPeTailSubmitJNI that needs one (BPSW_TAIL_POOL_WORKERS workers, default
// 8) and kept for the life of the process; the groups in flight, by handle.
std::mutex g_tail_mu;
std::map<int, bpsw_tail_pool_t*> g_tail_pools;
std::map<int64_t, std::unique_ptr<TailCall> > g_tail_calls;
int64_t g_tail_next = 1;

bpsw_tail_pool_t* tail_pool_of(int device) {
  std::lock_guard<std::mutex> lk(g_tail_mu);
  auto it = g_tail_pools.find(device);
  if (it != g_tail_pools.end()) return it->second;
  int w = getenv("BPSW_TAIL_POOL_WORKERS") ? atoi(getenv("BPSW_TAIL_POOL_WORKERS")) : 8;
  w = w < 1 ? 1 : (w > 64 ? 64 : w);
  bpsw_tail_pool_t* p = nullptr;
  if (bpsw_tail_pool_create(device, w, &p) != BPSW_OK) return nullptr;
  g_tail_pools[device] = p;
  return p;
}

}  // namespace

extern "C" {

JNIEXPORT jbyteArray JNICALL Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailJNI(
    JNIEnv* env, jobject, jintArray optInts, jdoubleArray realsArr, jbyteArray matArr, jlong id0, jintArray readLenArr, jbyteArray readsArr,
    jbyteArray qualsArr, jintArray nameLenArr, jbyteArray namesArr, jintArray regCntArr, jlongArray regLongsArr, jintArray regIntsArr,
    jlongArray outOffArr) {
  try {
  TailCall tc;
  if (!tail_unmarshal(env, "samPeTailJNI", optInts, realsArr, matArr, id0, readLenArr, readsArr, qualsArr, nameLenArr, namesArr, regCntArr,
                      regLongsArr, regIntsArr, tc)) return nullptr;
  if (!outOffArr || jni::GetArrayLength(env, outOffArr) < tc.n2 + 1) { throw_runtime(env, "bPSW: samPeTailJNI: outOff needs 2*groupSize + 1 entries"); return nullptr; }
  bpsw_ctx_t* ctx = thread_context(env);
  if (!ctx) { throw_runtime(env, std::string("bPSW: no usable HIP device: ") + bpsw_last_error()); return nullptr; }
  size_t need = 0;
  int rc = bpsw_sam_pe_batch(ctx, &tc.opt, &tc.topt, &tc.g, tc.text.data(), tc.text.size(), tc.out_off.data(), &need, nullptr);
  if (rc == BPSW_ERR_CAPACITY && need > tc.text.size()) {
    tc.text.resize(need + 16);
    rc = bpsw_sam_pe_batch(ctx, &tc.opt, &tc.topt, &tc.g, tc.text.data(), tc.text.size(), tc.out_off.data(), &need, nullptr);
  }
  if (rc != BPSW_OK) { throw_runtime(env, std::string("bPSW: samPeTailJNI: ") + bpsw_last_error()); return nullptr; }
  return tail_result(env, "samPeTailJNI", tc, need, outOffArr);
  } catch (const std::exception& e) {  // nothing C++ may unwind into the JVM (std::bad_alloc of a scratch vector, ...)
    throw_runtime(env, std::string("bPSW: samPeTailJNI: ") + e.what());
    return nullptr;
  }
}

// The same call in two halves (round 5): the task thread only ENQUEUES its groups -- the arguments are copied into native buffers behind
// a handle, and one of the library's tail workers of the thread's device (bpsw_tail_pool_*, include/bpsw.h) does the plan, the kernel
// and the text -- and collects the text later, so that the groups of a partition overlap instead of queueing behind one another on the
// partition's own thread (worker2/MemSamPe.scala:2099 under FastMap.scala:266-293).
//   @native def samPeTailSubmitJNI(<the arguments of samPeTailJNI without outOff>): Long
//   @native def samPeTailCollectJNI(handle: Long, outOff: Array[Long]): Array[Byte]      // blocks for that group; each handle once
JNIEXPORT jlong JNICALL Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailSubmitJNI(
    JNIEnv* env, jobject, jintArray optInts, jdoubleArray realsArr, jbyteArray matArr, jlong id0, jintArray readLenArr, jbyteArray readsArr,
    jbyteArray qualsArr, jintArray nameLenArr, jbyteArray namesArr, jintArray regCntArr, jlongArray regLongsArr, jintArray regIntsArr) {
  try {
  std::unique_ptr<TailCall> tc(new TailCall());
  if (!tail_unmarshal(env, "samPeTailSubmitJNI", optInts, realsArr, matArr, id0, readLenArr, readsArr, qualsArr, nameLenArr, namesArr, regCntArr,
                      regLongsArr, regIntsArr, *tc)) return 0;
  bpsw_ctx_t* ctx = thread_context(env);  // (the partition -> device choice of every other entry)
  if (!ctx) { throw_runtime(env, std::string("bPSW: no usable HIP device: ") + bpsw_last_error()); return 0; }
  tc->pool = tail_pool_of(bpsw_device_of(ctx));
  if (!tc->pool) { throw_runtime(env, std::string("bPSW: samPeTailSubmitJNI: ") + bpsw_last_error()); return 0; }
  if (bpsw_tail_pool_submit(tc->pool, &tc->opt, &tc->topt, &tc->g, BPSW_TAIL_POOL_TAIL_ONLY, tc->text.data(), tc->text.size(), tc->out_off.data(),
                            nullptr, nullptr, 0, &tc->ticket) != BPSW_OK) {
    throw_runtime(env, std::string("bPSW: samPeTailSubmitJNI: ") + bpsw_last_error());
    return 0;
  }
  std::lock_guard<std::mutex> lk(g_tail_mu);
  const int64_t h = g_tail_next++;
  g_tail_calls[h] = std::move(tc);
  return (jlong)h;
  } catch (const std::exception& e) {
    throw_runtime(env, std::string("bPSW: samPeTailSubmitJNI: ") + e.what());
    return 0;
  }
}

JNIEXPORT jbyteArray JNICALL Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailCollectJNI(JNIEnv* env, jobject, jlong handle, jlongArray outOffArr) {
  try {
  std::unique_ptr<TailCall> tc;
  {
    std::lock_guard<std::mutex> lk(g_tail_mu);
    auto it = g_tail_calls.find((int64_t)handle);
    if (it == g_tail_calls.end()) { throw_runtime(env, "bPSW: samPeTailCollectJNI: unknown handle (never issued, or collected already)"); return nullptr; }
    tc = std::move(it->second);
    g_tail_calls.erase(it);
  }
  size_t need = 0;
  int rc = bpsw_tail_pool_wait(tc->pool, tc->ticket, &need, nullptr);
  if (rc == BPSW_ERR_CAPACITY && need > tc->text.size()) {  // (a group whose text outgrew the first guess: once more, sized)
    tc->text.resize(need + 16);
    rc = bpsw_tail_pool_submit(tc->pool, &tc->opt, &tc->topt, &tc->g, BPSW_TAIL_POOL_TAIL_ONLY, tc->text.data(), tc->text.size(), tc->out_off.data(),
                               nullptr, nullptr, 0, &tc->ticket);
    if (rc == BPSW_OK) rc = bpsw_tail_pool_wait(tc->pool, tc->ticket, &need, nullptr);
  }
  if (rc != BPSW_OK) { throw_runtime(env, std::string("bPSW: samPeTailCollectJNI: ") + bpsw_last_error()); return nullptr; }
  if (!outOffArr || jni::GetArrayLength(env, outOffArr) < tc->n2 + 1) { throw_runtime(env, "bPSW: samPeTailCollectJNI: outOff needs 2*groupSize + 1 entries"); return nullptr; }
  return tail_result(env, "samPeTailCollectJNI", *tc, need, outOffArr);
  } catch (const std::exception& e) {
    throw_runtime(env, std::string("bPSW: samPeTailCollectJNI: ") + e.what());
    return nullptr;
  }
}

// A handle that will not be collected (the Spark task failed or was killed between submit and collect): wait for the group's ticket -- a
// tail worker may still be writing into the handle's buffers -- and drop it.  Call it from the `finally` of the code that holds handles;
// without it the whole TailCall (reads, qualities, names, regions, text: several MB per group) and the pool's job entry stay for the life
// of the executor JVM (advisor, round 5).  Unknown handles (collected already, never issued) are ignored; returns 1 when one was dropped.
//   @native def samPeTailCancelJNI(handle: Long): Int
JNIEXPORT jint JNICALL Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailCancelJNI(JNIEnv* env, jobject, jlong handle) {
  try {
  std::unique_ptr<TailCall> tc;
  {
    std::lock_guard<std::mutex> lk(g_tail_mu);
    auto it = g_tail_calls.find((int64_t)handle);
    if (it == g_tail_calls.end()) return 0;
    tc = std::move(it->second);
    g_tail_calls.erase(it);
  }
  size_t need = 0;
  (void)bpsw_tail_pool_wait(tc->pool, tc->ticket, &need, nullptr);  // (whatever it returns: the worker is done with the buffers afterwards)
  return 1;
  } catch (const std::exception& e) {
    throw_runtime(env, std::string("bPSW: samPeTailCancelJNI: ") + e.what());
    return 0;
  }
}

}  // extern "C"
