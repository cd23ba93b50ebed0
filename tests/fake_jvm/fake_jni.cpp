// fake_jni.cpp -- TEST INFRASTRUCTURE: a miniature JVM object model behind a real JNIEnv function table, so the
// JNI shim of libbPSW_hip.so (csrc/bpsw_jni.cpp) can be driven end to end without a JVM (none exists in the image).
// Only the table slots the shim uses are populated; any other slot aborts loudly.  Objects carry their fields by name,
// exactly what GetFieldID/Get*Field need.  The two drivers at the bottom build the Scala-side argument graphs
// (SeqSWType / MateSWType / RefSWType / MemOptType / MemPeStat, jni/*.scala, datatype/*.scala) from flat arrays,
// call the exported Java_* symbol, and flatten what comes back.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "jni_min.h"

namespace {

// Field names are interned to small integers once (GetFieldID does it for the shim, the drivers below for the harness), and an
// object keeps its fields in slot-indexed vectors: a Get<Type>Field is an index, as in a JVM (an offset), not a string-keyed map
// lookup -- the shim's micro-benchmark (bpsw_hip/jnishim.py) times the shim, not this harness.
int intern_field(const std::string& name);
template <class T>
struct Slots {
  std::vector<T> v;
  std::vector<uint8_t> has;
  T& at_id(int id) {
    if ((size_t)id >= v.size()) { v.resize((size_t)id + 1, T()); has.resize((size_t)id + 1, 0); }
    has[(size_t)id] = 1;
    return v[(size_t)id];
  }
  const T* find_id(int id) const { return ((size_t)id < v.size() && has[(size_t)id]) ? &v[(size_t)id] : nullptr; }
  T& operator[](const std::string& name) { return at_id(intern_field(name)); }  // harness side: by name
};
struct FObj {
  std::string cls;  // class name, or "[B" "[S" "[I" "[J" "[L" for arrays, or "class" for a jclass handle
  Slots<int64_t> ints;
  Slots<double> dbls;
  Slots<FObj*> objs;
  std::vector<int8_t> bytes;
  std::vector<int16_t> shorts;
  std::vector<int32_t> ia;
  std::vector<int64_t> la;
  std::vector<double> da;
  std::vector<FObj*> elems;
};
struct FField { std::string name, sig; int id; };

// What a real JVM keeps for the life of the process: loaded classes and their field / method IDs.  The shim caches both
// (global references + IDs) across calls, so they must outlive the per-call object heap below.  `lookups` counts FindClass /
// GetFieldID / GetMethodID calls, so a test can see that a second call resolves nothing by name.
struct ClassWorld {
  std::mutex mu;
  std::vector<std::unique_ptr<FObj>> classes_heap;
  std::vector<std::unique_ptr<FField>> fields;
  std::map<std::string, FObj*> classes;
  std::atomic<long> lookups{0};
  std::atomic<long> global_refs{0};
};
ClassWorld g_world;
int intern_field(const std::string& name) {
  static std::mutex mu;
  static std::map<std::string, int> ids;
  std::lock_guard<std::mutex> lk(mu);
  auto it = ids.find(name);
  if (it != ids.end()) return it->second;
  const int id = (int)ids.size();
  ids[name] = id;
  return id;
}

struct Jvm {
  std::vector<std::unique_ptr<FObj>> heap;
  bool pending = false;
  std::string pending_msg;
  int partition = -1;  // >= 0: org.apache.spark.TaskContext.get().partitionId() answers this
  long calls[JNI_SLOT_COUNT] = {0};
  FObj* alloc(const std::string& cls) { heap.emplace_back(new FObj()); heap.back()->cls = cls; return heap.back().get(); }
};
thread_local Jvm* g_vm = nullptr;  // one fake JVM per driver call, one driver call per thread

const char* kKnown[] = {"cs/ucla/edu/bwaspark/datatype/MemAlnRegType", "cs/ucla/edu/bwaspark/datatype/MemOptType",
                        "cs/ucla/edu/bwaspark/datatype/MemPeStat", "cs/ucla/edu/bwaspark/jni/MateSWType",
                        "cs/ucla/edu/bwaspark/jni/SeqSWType", "cs/ucla/edu/bwaspark/jni/RefSWType",
                        "java/lang/RuntimeException"};

FObj* O(jobject o) { return reinterpret_cast<FObj*>(o); }
jobject J(FObj* o) { return reinterpret_cast<jobject>(o); }

jclass f_FindClass(JNIEnv*, const char* name) {
  g_vm->calls[JNI_SLOT_FindClass]++;
  bool ok = false;
  for (const char* k : kKnown) ok |= strcmp(k, name) == 0;
  if (strcmp(name, "org/apache/spark/TaskContext") == 0) ok = true;  // on the class path; get() is null outside a task
  if (!ok) { g_vm->pending = true; g_vm->pending_msg = std::string("NoClassDefFoundError: ") + name; return nullptr; }
  g_world.lookups++;
  std::lock_guard<std::mutex> lk(g_world.mu);
  auto it = g_world.classes.find(name);
  if (it != g_world.classes.end()) return J(it->second);
  g_world.classes_heap.emplace_back(new FObj());
  FObj* c = g_world.classes_heap.back().get();
  c->cls = std::string("class:") + name;
  g_world.classes[name] = c;
  return J(c);
}
jobject f_NewGlobalRef(JNIEnv*, jobject o) {  // only classes are ever made global by the shim; they live in g_world already
  if (o && O(o)->cls.rfind("class:", 0) != 0) { fprintf(stderr, "fake_jni: NewGlobalRef of a non-class object\n"); abort(); }
  g_world.global_refs++;
  return o;
}
void f_DeleteGlobalRef(JNIEnv*, jobject) { g_world.global_refs--; }
jint f_ThrowNew(JNIEnv*, jclass c, const char* msg) {
  g_vm->pending = true;
  g_vm->pending_msg = O(c)->cls.substr(6) + ": " + msg;
  return 0;
}
void f_ExceptionClear(JNIEnv*) { g_vm->pending = false; }
jboolean f_ExceptionCheck(JNIEnv*) { return g_vm->pending ? 1 : 0; }
jint f_PushLocalFrame(JNIEnv*, jint) { g_vm->calls[JNI_SLOT_PushLocalFrame]++; return JNI_OK; }
jobject f_PopLocalFrame(JNIEnv*, jobject r) { g_vm->calls[JNI_SLOT_PopLocalFrame]++; return r; }
void f_DeleteLocalRef(JNIEnv*, jobject) {}
jobject f_AllocObject(JNIEnv*, jclass c) { return J(g_vm->alloc(O(c)->cls.substr(6))); }
jfieldID f_GetFieldID(JNIEnv*, jclass, const char* name, const char* sig) {
  g_vm->calls[JNI_SLOT_GetFieldID]++;
  g_world.lookups++;
  std::lock_guard<std::mutex> lk(g_world.mu);
  g_world.fields.emplace_back(new FField{name, sig, intern_field(name)});
  return reinterpret_cast<jfieldID>(g_world.fields.back().get());
}
const FField* F(jfieldID f) { return reinterpret_cast<const FField*>(f); }
jobject f_GetObjectField(JNIEnv*, jobject o, jfieldID f) {
  FObj* const* p = O(o)->objs.find_id(F(f)->id);
  return p ? J(*p) : nullptr;
}
jint f_GetIntField(JNIEnv*, jobject o, jfieldID f) { const int64_t* p = O(o)->ints.find_id(F(f)->id); return p ? (jint)*p : 0; }
jlong f_GetLongField(JNIEnv*, jobject o, jfieldID f) { const int64_t* p = O(o)->ints.find_id(F(f)->id); return p ? (jlong)*p : 0; }
jfloat f_GetFloatField(JNIEnv*, jobject o, jfieldID f) { const double* p = O(o)->dbls.find_id(F(f)->id); return p ? (jfloat)*p : 0.f; }
jdouble f_GetDoubleField(JNIEnv*, jobject o, jfieldID f) { const double* p = O(o)->dbls.find_id(F(f)->id); return p ? *p : 0.0; }
void f_SetObjectField(JNIEnv*, jobject o, jfieldID f, jobject v) { O(o)->objs.at_id(F(f)->id) = O(v); }
void f_SetIntField(JNIEnv*, jobject o, jfieldID f, jint v) { O(o)->ints.at_id(F(f)->id) = v; }
void f_SetLongField(JNIEnv*, jobject o, jfieldID f, jlong v) { O(o)->ints.at_id(F(f)->id) = v; }
jmethodID f_GetMethodID(JNIEnv*, jclass, const char* name, const char*) {
  g_world.lookups++;
  std::lock_guard<std::mutex> lk(g_world.mu);
  g_world.fields.emplace_back(new FField{name, "()"});
  return reinterpret_cast<jmethodID>(g_world.fields.back().get());
}
jmethodID f_GetStaticMethodID(JNIEnv* e, jclass c, const char* name, const char* sig) { return f_GetMethodID(e, c, name, sig); }
jobject f_CallStaticObjectMethod(JNIEnv*, jclass, jmethodID, ...) {  // TaskContext.get(): null when the thread runs no task
  return g_vm->partition >= 0 ? J(g_vm->alloc("org/apache/spark/TaskContext")) : nullptr;
}
jint f_CallIntMethod(JNIEnv*, jobject, jmethodID, ...) { return g_vm->partition; }
jsize f_GetArrayLength(JNIEnv*, jarray a) {
  FObj* o = O(a);
  if (o->cls == "[B") return (jsize)o->bytes.size();
  if (o->cls == "[S") return (jsize)o->shorts.size();
  if (o->cls == "[I") return (jsize)o->ia.size();
  if (o->cls == "[J") return (jsize)o->la.size();
  if (o->cls == "[D") return (jsize)o->da.size();
  return (jsize)o->elems.size();
}
jobjectArray f_NewObjectArray(JNIEnv*, jsize n, jclass, jobject) {
  FObj* a = g_vm->alloc("[L");
  a->elems.assign((size_t)n, nullptr);
  return J(a);
}
jobject f_GetObjectArrayElement(JNIEnv*, jobjectArray a, jsize i) {
  return (i < 0 || (size_t)i >= O(a)->elems.size()) ? nullptr : J(O(a)->elems[(size_t)i]);
}
void f_SetObjectArrayElement(JNIEnv*, jobjectArray a, jsize i, jobject v) { O(a)->elems[(size_t)i] = O(v); }
jshortArray f_NewShortArray(JNIEnv*, jsize n) {
  FObj* a = g_vm->alloc("[S");
  a->shorts.assign((size_t)n, 0);
  return J(a);
}
jlongArray f_NewLongArray(JNIEnv*, jsize n) {
  FObj* a = g_vm->alloc("[J");
  a->la.assign((size_t)n, 0);
  return (jlongArray)J(a);
}
void f_SetLongArrayRegion(JNIEnv*, jlongArray a, jsize s, jsize l, const jlong* b) { memcpy(O(a)->la.data() + s, b, 8 * (size_t)l); }
void f_GetByteArrayRegion(JNIEnv*, jbyteArray a, jsize s, jsize l, jbyte* b) { memcpy(b, O(a)->bytes.data() + s, (size_t)l); }
void f_GetIntArrayRegion(JNIEnv*, jintArray a, jsize s, jsize l, jint* b) { memcpy(b, O(a)->ia.data() + s, 4 * (size_t)l); }
void f_GetLongArrayRegion(JNIEnv*, jlongArray a, jsize s, jsize l, jlong* b) { memcpy(b, O(a)->la.data() + s, 8 * (size_t)l); }
jbyteArray f_NewByteArray(JNIEnv*, jsize n) {
  FObj* a = g_vm->alloc("[B");
  a->bytes.assign((size_t)n, 0);
  return (jbyteArray)J(a);
}
void f_SetByteArrayRegion(JNIEnv*, jbyteArray a, jsize s, jsize l, const jbyte* b) { memcpy(O(a)->bytes.data() + s, b, (size_t)l); }
void f_GetDoubleArrayRegion(JNIEnv*, jdoubleArray a, jsize s, jsize l, jdouble* b) { memcpy(b, O(a)->da.data() + s, 8 * (size_t)l); }
void f_SetShortArrayRegion(JNIEnv*, jshortArray a, jsize s, jsize l, const jshort* b) { memcpy(O(a)->shorts.data() + s, b, 2 * (size_t)l); }

void unpopulated() {
  fprintf(stderr, "fake_jni: the shim called a JNI slot this fake does not implement\n");
  abort();
}

struct Env {
  JNINativeInterface_ table;
  const JNINativeInterface_* env;  // JNIEnv = pointer to the table
  Env() {
    for (auto& s : table.slot) s = reinterpret_cast<void*>(&unpopulated);
#define SET(name) table.slot[JNI_SLOT_##name] = reinterpret_cast<void*>(&f_##name)
    SET(FindClass); SET(ThrowNew); SET(ExceptionClear); SET(ExceptionCheck); SET(PushLocalFrame); SET(PopLocalFrame);
    SET(DeleteLocalRef); SET(AllocObject); SET(GetFieldID); SET(GetObjectField); SET(GetIntField); SET(GetLongField);
    SET(GetFloatField); SET(GetDoubleField); SET(SetObjectField); SET(SetIntField); SET(SetLongField); SET(GetMethodID);
    SET(GetStaticMethodID); SET(CallStaticObjectMethod); SET(CallIntMethod); SET(GetArrayLength); SET(NewObjectArray);
    SET(GetObjectArrayElement); SET(SetObjectArrayElement); SET(NewShortArray); SET(GetByteArrayRegion);
    SET(GetIntArrayRegion); SET(GetLongArrayRegion); SET(SetShortArrayRegion); SET(NewLongArray); SET(SetLongArrayRegion);
    SET(NewByteArray); SET(SetByteArrayRegion); SET(GetDoubleArrayRegion); SET(NewGlobalRef); SET(DeleteGlobalRef);
#undef SET
    env = &table;
  }
};

FObj* byte_array(const uint8_t* p, size_t n) {
  FObj* a = g_vm->alloc("[B");
  a->bytes.assign(reinterpret_cast<const int8_t*>(p), reinterpret_cast<const int8_t*>(p) + n);
  return a;
}

FObj* int_array(const int32_t* p, size_t n) {
  FObj* a = g_vm->alloc("[I");
  a->ia.assign(p, p + n);
  return a;
}
FObj* long_array(const int64_t* p, size_t n) {
  FObj* a = g_vm->alloc("[J");
  a->la.assign(p, p + n);
  return a;
}

FObj* double_array(const double* p, size_t n) {
  FObj* a = g_vm->alloc("[D");
  a->da.assign(p, p + n);
  return a;
}

struct FlatReg {
  int64_t rb, re;
  int32_t qb, qe, score, truesc, sub, csub, sub_n, w, seedcov, secondary;
  uint64_t hash;
};
FObj* reg_object(const FlatReg& r) {  // MemAlnRegType.scala:26-38
  FObj* a = g_vm->alloc("cs/ucla/edu/bwaspark/datatype/MemAlnRegType");
  a->ints["rBeg"] = r.rb; a->ints["rEnd"] = r.re; a->ints["qBeg"] = r.qb; a->ints["qEnd"] = r.qe;
  a->ints["score"] = r.score; a->ints["trueScore"] = r.truesc; a->ints["sub"] = r.sub; a->ints["csub"] = r.csub;
  a->ints["subNum"] = r.sub_n; a->ints["width"] = r.w; a->ints["seedCov"] = r.seedcov; a->ints["secondary"] = r.secondary;
  a->ints["hash"] = (int64_t)r.hash;
  return a;
}

void* load_symbol(const char* lib, const char* sym, char* err, size_t errcap) {
  void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
  if (!h) { snprintf(err, errcap, "dlopen: %s", dlerror()); return nullptr; }
  void* f = dlsym(h, sym);
  if (!f) snprintf(err, errcap, "dlsym: %s", dlerror());
  return f;
}

}  // namespace

extern "C" {

// returns 0 on success, 1 if the shim left a Java exception pending (text in err), -1 on harness failure
int fake_jvm_extend(const char* lib, int partition, const uint8_t* wire, int wire_bytes, int ret_task_num, int16_t* out,
                    char* err, int errcap) {
  Jvm vm;
  g_vm = &vm;
  vm.partition = partition;
  Env e;
  typedef jshortArray (*Fn)(JNIEnv*, jobject, jint, jbyteArray);
  Fn fn = (Fn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_SWExtendFPGAJNI_swExtendFPGAJNI", err, (size_t)errcap);
  if (!fn) return -1;
  FObj* self = vm.alloc("cs/ucla/edu/bwaspark/jni/SWExtendFPGAJNI");
  jshortArray r = fn(&e.env, J(self), ret_task_num, J(byte_array(wire, (size_t)wire_bytes)));
  if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
  if (!r || (int)O(r)->shorts.size() != ret_task_num) { snprintf(err, (size_t)errcap, "bad result array"); return -1; }
  memcpy(out, O(r)->shorts.data(), 2 * (size_t)ret_task_num);
  return 0;
}

int fake_jvm_matesw(const char* lib, int partition, const int32_t opt_ints[16], float mask_level_redun, const int8_t mat[25],
                    int64_t l_pac, const double* pes /*4x5: low high failed avg std*/, int group_size, const int32_t* seq_len,
                    const int64_t* seq_off, const uint8_t* seq_pool, const int32_t* reg_cnt, const FlatReg* regs,
                    const int32_t* ref_cnt, const int64_t* ref_rb, const int64_t* ref_re, const int64_t* ref_len,
                    const int64_t* ref_off, const uint8_t* ref_pool, int32_t* out_cnt, FlatReg* out_regs, int64_t out_cap,
                    int64_t* out_total, long* local_frames, char* err, int errcap, const uint8_t* pac /* may be null */) {
  Jvm vm;
  g_vm = &vm;
  vm.partition = partition;
  Env e;
  if (pac) {  // the driver loads the reference once, then sends coordinates only
    typedef jint (*LoadFn)(JNIEnv*, jobject, jbyteArray, jlong);
    LoadFn load = (LoadFn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_loadPacJNI", err, (size_t)errcap);
    if (!load) return -1;
    FObj* self0 = vm.alloc("cs/ucla/edu/bwaspark/jni/MateSWJNI");
    const jint nd = load(&e.env, J(self0), (jbyteArray)J(byte_array(pac, (size_t)((l_pac + 3) / 4))), (jlong)l_pac);
    if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
    if (nd < 1) { snprintf(err, (size_t)errcap, "loadPacJNI loaded no device"); return -1; }
  }
  typedef jobjectArray (*Fn)(JNIEnv*, jobject, jobject, jlong, jobjectArray, jint, jobjectArray, jobjectArray, jobjectArray, jintArray);
  Fn fn = (Fn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_mateSWJNI", err, (size_t)errcap);
  if (!fn) return -1;
  // MemOptType (datatype/MemOptType.scala:28-56)
  static const char* names[16] = {"a", "b", "oDel", "eDel", "oIns", "eIns", "penUnpaired", "penClip5", "penClip3", "w", "zdrop",
                                  "T", "flag", "minSeedLen", "maxIns", "maxMatesw"};
  FObj* opt = vm.alloc("cs/ucla/edu/bwaspark/datatype/MemOptType");
  for (int i = 0; i < 16; ++i) opt->ints[names[i]] = opt_ints[i];
  opt->dbls["maskLevelRedun"] = mask_level_redun;
  opt->objs["mat"] = byte_array(reinterpret_cast<const uint8_t*>(mat), 25);
  FObj* pes_arr = vm.alloc("[L");
  for (int r = 0; r < 4; ++r) {
    FObj* p = vm.alloc("cs/ucla/edu/bwaspark/datatype/MemPeStat");
    p->ints["low"] = (int64_t)pes[5 * r]; p->ints["high"] = (int64_t)pes[5 * r + 1]; p->ints["failed"] = (int64_t)pes[5 * r + 2];
    p->dbls["avg"] = pes[5 * r + 3]; p->dbls["std"] = pes[5 * r + 4];
    pes_arr->elems.push_back(p);
  }
  FObj *seqs = vm.alloc("[L"), *mates = vm.alloc("[L"), *refs = vm.alloc("[L"), *ref_sizes = vm.alloc("[I");
  int64_t reg_at = 0, row = 0;
  for (int k = 0; k < group_size; ++k)  // memSamPeGroupJNIPrepare emits all refs first (MemSamPe.scala:1931-1950)
    for (int i = 0; i < 2; ++i) {
      const int e2 = 2 * k + i;
      ref_sizes->ia.push_back(ref_cnt[e2]);
      for (int j = 0; j < ref_cnt[e2]; ++j, ++row) {
        FObj* r = vm.alloc("cs/ucla/edu/bwaspark/jni/RefSWType");
        r->ints["readIdx"] = k; r->ints["pairIdx"] = i; r->ints["regIdx"] = j;
        FObj *rb = vm.alloc("[J"), *re = vm.alloc("[J"), *ln = vm.alloc("[J");
        for (int o = 0; o < 4; ++o) {
          static const char* rn[4] = {"ref0", "ref1", "ref2", "ref3"};
          rb->la.push_back(ref_rb[4 * row + o]); re->la.push_back(ref_re[4 * row + o]);
          if (pac) {  // coordinates only: lenArray = rEnd - rBeg, no bytes
            ln->la.push_back(ref_rb[4 * row + o] < 0 ? 0 : ref_re[4 * row + o] - ref_rb[4 * row + o]);
            r->objs[rn[o]] = nullptr;
            continue;
          }
          ln->la.push_back(ref_len[4 * row + o]);
          r->objs[rn[o]] = ref_len[4 * row + o] > 0 ? byte_array(ref_pool + ref_off[4 * row + o], (size_t)ref_len[4 * row + o]) : nullptr;
        }
        r->objs["rBegArray"] = rb; r->objs["rEndArray"] = re; r->objs["lenArray"] = ln;
        refs->elems.push_back(r);
      }
    }
  for (int k = 0; k < group_size; ++k)  // then seqs and regions in (k,i,j) order (MemSamPe.scala:1962-1990)
    for (int i = 0; i < 2; ++i) {
      const int e2 = 2 * k + i;
      FObj* s = vm.alloc("cs/ucla/edu/bwaspark/jni/SeqSWType");
      s->ints["readIdx"] = k; s->ints["pairIdx"] = i; s->ints["seqLength"] = seq_len[e2];
      s->objs["seqTrans"] = byte_array(seq_pool + seq_off[e2], (size_t)seq_len[e2]);
      seqs->elems.push_back(s);
      for (int j = 0; j < reg_cnt[e2]; ++j, ++reg_at) {
        FObj* m = vm.alloc("cs/ucla/edu/bwaspark/jni/MateSWType");
        m->ints["readIdx"] = k; m->ints["pairIdx"] = i; m->ints["regIdx"] = j;
        m->objs["alnReg"] = reg_object(regs[reg_at]);
        mates->elems.push_back(m);
      }
    }
  // FAKE_JVM_SHUFFLE=1: the arrays in an order memSamPeGroupJNIPrepare never produces (RefSWType[] and SeqSWType[] reversed) -- the contract
  // keys every object by its own (readIdx, pairIdx, regIdx), so any order is legal; the shim's lazy path must notice and step aside
  if (getenv("FAKE_JVM_SHUFFLE") && atoi(getenv("FAKE_JVM_SHUFFLE")) != 0) {
    std::reverse(refs->elems.begin(), refs->elems.end());
    std::reverse(seqs->elems.begin(), seqs->elems.end());
  }
  FObj* self = vm.alloc("cs/ucla/edu/bwaspark/jni/MateSWJNI");
  jobjectArray r = fn(&e.env, J(self), J(opt), (jlong)l_pac, J(pes_arr), group_size, J(seqs), J(mates), J(refs), J(ref_sizes));
  if (local_frames) *local_frames = vm.calls[JNI_SLOT_PushLocalFrame] - vm.calls[JNI_SLOT_PopLocalFrame];
  if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
  if (!r) { snprintf(err, (size_t)errcap, "null result"); return -1; }
  // mateSWArrayToAlnRegPairArray (MemSamPe.scala:2010-2044): group by (readIdx, pairIdx) in arrival order
  memset(out_cnt, 0, sizeof(int32_t) * 2 * (size_t)group_size);
  *out_total = (int64_t)O(r)->elems.size();
  if (*out_total > out_cap) { snprintf(err, (size_t)errcap, "out_cap too small"); return -1; }
  int64_t at = 0;
  int last_e = -1, last_rank = -1;
  for (FObj* m : O(r)->elems) {
    if (!m || m->cls != "cs/ucla/edu/bwaspark/jni/MateSWType" || !m->objs["alnReg"]) { snprintf(err, (size_t)errcap, "malformed MateSWType"); return -1; }
    const int e2 = 2 * (int)m->ints["readIdx"] + (int)m->ints["pairIdx"];
    if (e2 < last_e || (e2 == last_e && (int)m->ints["regIdx"] != last_rank + 1) || (e2 != last_e && m->ints["regIdx"] != 0)) {
      snprintf(err, (size_t)errcap, "result not in (k,i,rank) order"); return -1;
    }
    last_rank = (int)m->ints["regIdx"]; last_e = e2;
    out_cnt[e2]++;
    FObj* a = m->objs["alnReg"];
    FlatReg& o = out_regs[at++];
    o.rb = a->ints["rBeg"]; o.re = a->ints["rEnd"]; o.qb = (int32_t)a->ints["qBeg"]; o.qe = (int32_t)a->ints["qEnd"];
    o.score = (int32_t)a->ints["score"]; o.truesc = (int32_t)a->ints["trueScore"]; o.sub = (int32_t)a->ints["sub"];
    o.csub = (int32_t)a->ints["csub"]; o.sub_n = (int32_t)a->ints["subNum"]; o.w = (int32_t)a->ints["width"];
    o.seedcov = (int32_t)a->ints["seedCov"]; o.secondary = (int32_t)a->ints["secondary"]; o.hash = (uint64_t)a->ints["hash"];
  }
  return 0;
}

// mateSWFlatJNI (round 4): what the flattened memSamPeGroupJNIPrepare of INTEGRATION.md 1e hands over -- primitive arrays only.
// Same inputs and outputs as fake_jvm_matesw (the object-array entry), so a test can hold the two against each other.
int fake_jvm_matesw_flat(const char* lib, int partition, const int32_t opt_ints[16], float mask_level_redun, const int8_t mat[25],
                         int64_t l_pac, const double* pes /*4x5*/, int group_size, const int32_t* seq_len, const int64_t* seq_off,
                         const uint8_t* seq_pool, const int32_t* reg_cnt, const FlatReg* regs, const int32_t* ref_cnt, const int64_t* ref_rb,
                         const int64_t* ref_re, const int64_t* ref_len, const int64_t* ref_off, const uint8_t* ref_pool, int32_t* out_cnt,
                         FlatReg* out_regs, int64_t out_cap, int64_t* out_total, char* err, int errcap, const uint8_t* pac /* may be null */) {
  Jvm vm;
  g_vm = &vm;
  vm.partition = partition;
  Env e;
  FObj* self = vm.alloc("cs/ucla/edu/bwaspark/jni/MateSWJNI");
  if (pac) {
    typedef jint (*LoadFn)(JNIEnv*, jobject, jbyteArray, jlong);
    LoadFn load = (LoadFn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_loadPacJNI", err, (size_t)errcap);
    if (!load) return -1;
    const jint nd = load(&e.env, J(self), (jbyteArray)J(byte_array(pac, (size_t)((l_pac + 3) / 4))), (jlong)l_pac);
    if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
    if (nd < 1) { snprintf(err, (size_t)errcap, "loadPacJNI loaded no device"); return -1; }
  }
  typedef jlongArray (*Fn)(JNIEnv*, jobject, jintArray, jfloat, jbyteArray, jlong, jdoubleArray, jint, jintArray, jbyteArray, jintArray,
                           jlongArray, jintArray, jintArray, jlongArray, jlongArray, jlongArray, jbyteArray);
  Fn fn = (Fn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_mateSWFlatJNI", err, (size_t)errcap);
  if (!fn) return -1;
  const size_t ends = 2 * (size_t)group_size;
  std::vector<uint8_t> seqs;
  int64_t n_regs = 0, rows = 0;
  for (size_t e2 = 0; e2 < ends; ++e2) {
    seqs.insert(seqs.end(), seq_pool + seq_off[e2], seq_pool + seq_off[e2] + seq_len[e2]);
    n_regs += reg_cnt[e2]; rows += ref_cnt[e2];
  }
  std::vector<int64_t> rl((size_t)(3 * n_regs)), lens;
  std::vector<int32_t> ri((size_t)(10 * n_regs));
  for (int64_t j = 0; j < n_regs; ++j) {
    const FlatReg& r = regs[j];
    rl[(size_t)(3 * j)] = r.rb; rl[(size_t)(3 * j + 1)] = r.re; rl[(size_t)(3 * j + 2)] = (int64_t)r.hash;
    int32_t* v = ri.data() + 10 * j;
    v[0] = r.qb; v[1] = r.qe; v[2] = r.score; v[3] = r.truesc; v[4] = r.sub; v[5] = r.csub; v[6] = r.sub_n; v[7] = r.w; v[8] = r.seedcov;
    v[9] = r.secondary;
  }
  std::vector<uint8_t> windows;
  if (!pac) {
    lens.assign(ref_len, ref_len + 4 * rows);
    for (int64_t x = 0; x < 4 * rows; ++x) {
      if (lens[(size_t)x] < 0) lens[(size_t)x] = 0;
      if (lens[(size_t)x] > 0) windows.insert(windows.end(), ref_pool + ref_off[x], ref_pool + ref_off[x] + lens[(size_t)x]);
    }
  }
  jlongArray r = fn(&e.env, J(self), (jintArray)J(int_array(opt_ints, 16)), mask_level_redun,
                    (jbyteArray)J(byte_array(reinterpret_cast<const uint8_t*>(mat), 25)), (jlong)l_pac, (jdoubleArray)J(double_array(pes, 20)),
                    group_size, (jintArray)J(int_array(seq_len, ends)), (jbyteArray)J(byte_array(seqs.data(), seqs.size())),
                    (jintArray)J(int_array(reg_cnt, ends)), (jlongArray)J(long_array(rl.data(), rl.size())),
                    (jintArray)J(int_array(ri.data(), ri.size())), (jintArray)J(int_array(ref_cnt, ends)),
                    (jlongArray)J(long_array(ref_rb, (size_t)(4 * rows))), (jlongArray)J(long_array(ref_re, (size_t)(4 * rows))),
                    pac ? nullptr : (jlongArray)J(long_array(lens.data(), lens.size())),
                    pac ? nullptr : (jbyteArray)J(byte_array(windows.data(), windows.size())));
  if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
  if (!r) { snprintf(err, (size_t)errcap, "null result"); return -1; }
  const std::vector<int64_t>& la = O(r)->la;
  if (la.size() < ends || (la.size() - ends) % 8) { snprintf(err, (size_t)errcap, "malformed result array"); return -1; }
  *out_total = (int64_t)((la.size() - ends) / 8);
  if (*out_total > out_cap) { snprintf(err, (size_t)errcap, "out_cap too small"); return -1; }
  int64_t sum = 0;
  for (size_t e2 = 0; e2 < ends; ++e2) { out_cnt[e2] = (int32_t)la[e2]; sum += la[e2]; }
  if (sum != *out_total) { snprintf(err, (size_t)errcap, "counts do not add up to the regions returned"); return -1; }
  for (int64_t j = 0; j < *out_total; ++j) {
    const int64_t* o = la.data() + ends + 8 * j;
    FlatReg& q = out_regs[j];
    q.rb = o[0]; q.re = o[1]; q.hash = (uint64_t)o[2];
    q.qb = (int32_t)(uint32_t)o[3]; q.qe = (int32_t)(o[3] >> 32); q.score = (int32_t)(uint32_t)o[4]; q.truesc = (int32_t)(o[4] >> 32);
    q.sub = (int32_t)(uint32_t)o[5]; q.csub = (int32_t)(o[5] >> 32); q.sub_n = (int32_t)(uint32_t)o[6]; q.w = (int32_t)(o[6] >> 32);
    q.seedcov = (int32_t)(uint32_t)o[7]; q.secondary = (int32_t)(o[7] >> 32);
  }
  return 0;
}

// loadPacJNI, then chainToAlnJNI with primitive arrays (reads back to back).  out receives the returned
// long[] (n counts, then 8 longs per region); *out_n its length.
int fake_jvm_chain2aln(const char* lib, int partition, const uint8_t* pac, int64_t l_pac, const int32_t opt_ints[10],
                       const int8_t mat[25], int n_reads, const int32_t* read_len, const uint8_t* reads, int64_t reads_bytes,
                       const int32_t* chain_cnt, int64_t n_chains, const int32_t* seed_cnt, int64_t n_seeds, const int64_t* seed_rbeg,
                       const int32_t* seed_qbeg, const int32_t* seed_len, int64_t* out, int64_t out_cap, int64_t* out_n, char* err,
                       int errcap) {
  Jvm vm;
  g_vm = &vm;
  vm.partition = partition;
  Env e;
  typedef jint (*LoadFn)(JNIEnv*, jobject, jbyteArray, jlong);
  LoadFn load = (LoadFn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_loadPacJNI", err, (size_t)errcap);
  if (!load) return -1;
  load(&e.env, J(vm.alloc("cs/ucla/edu/bwaspark/jni/MateSWJNI")), (jbyteArray)J(byte_array(pac, (size_t)((l_pac + 3) / 4))), (jlong)l_pac);
  if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
  typedef jlongArray (*Fn)(JNIEnv*, jobject, jintArray, jbyteArray, jintArray, jbyteArray, jintArray, jintArray, jlongArray, jintArray, jintArray);
  Fn fn = (Fn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_SWExtendFPGAJNI_chainToAlnJNI", err, (size_t)errcap);
  if (!fn) return -1;
  FObj* self = vm.alloc("cs/ucla/edu/bwaspark/jni/SWExtendFPGAJNI");
  jlongArray r = fn(&e.env, J(self), (jintArray)J(int_array(opt_ints, 10)), (jbyteArray)J(byte_array(reinterpret_cast<const uint8_t*>(mat), 25)),
                    (jintArray)J(int_array(read_len, (size_t)n_reads)), (jbyteArray)J(byte_array(reads, (size_t)reads_bytes)),
                    (jintArray)J(int_array(chain_cnt, (size_t)n_reads)), (jintArray)J(int_array(seed_cnt, (size_t)n_chains)),
                    (jlongArray)J(long_array(seed_rbeg, (size_t)n_seeds)), (jintArray)J(int_array(seed_qbeg, (size_t)n_seeds)),
                    (jintArray)J(int_array(seed_len, (size_t)n_seeds)));
  if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
  if (!r) { snprintf(err, (size_t)errcap, "null result"); return -1; }
  *out_n = (int64_t)O(r)->la.size();
  if (*out_n > out_cap) { snprintf(err, (size_t)errcap, "out_cap too small"); return -1; }
  memcpy(out, O(r)->la.data(), 8 * (size_t)*out_n);
  return 0;
}

// loadPacJNI + loadBnsJNI, then samPeTailJNI with primitive arrays.  out_text receives the returned
// byte[], out_off the offsets the native filled in.
static int sam_pe_tail_impl(const char* lib, int partition, const uint8_t* pac, int64_t l_pac, int n_seqs, const int64_t* ann_off,
                         const int32_t* ann_len, const uint8_t* ann_names, int64_t ann_names_bytes, const int32_t opt_ints[12],
                         const double reals[22], const int8_t mat[25], int64_t id0, int n2, const int32_t* read_len, const uint8_t* reads,
                         const uint8_t* quals, int64_t reads_bytes, const int32_t* name_len, const uint8_t* names, int64_t names_bytes,
                         const int32_t* reg_cnt, const int64_t* reg_longs, const int32_t* reg_ints, int64_t n_regs, uint8_t* out_text,
                         int64_t out_cap, int64_t* out_bytes, int64_t* out_off, char* err, int errcap, int async_copies) {
  Jvm vm;
  g_vm = &vm;
  vm.partition = partition;
  Env e;
  FObj* self = vm.alloc("cs/ucla/edu/bwaspark/jni/MateSWJNI");
  typedef jint (*LoadFn)(JNIEnv*, jobject, jbyteArray, jlong);
  LoadFn load = (LoadFn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_loadPacJNI", err, (size_t)errcap);
  if (!load) return -1;
  load(&e.env, J(self), (jbyteArray)J(byte_array(pac, (size_t)((l_pac + 3) / 4))), (jlong)l_pac);
  if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
  typedef jint (*BnsFn)(JNIEnv*, jobject, jlongArray, jintArray, jbyteArray);
  BnsFn bns = (BnsFn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_loadBnsJNI", err, (size_t)errcap);
  if (!bns) return -1;
  bns(&e.env, J(self), (jlongArray)J(long_array(ann_off, (size_t)n_seqs)), (jintArray)J(int_array(ann_len, (size_t)n_seqs)),
      (jbyteArray)J(byte_array(ann_names, (size_t)ann_names_bytes)));
  if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
  typedef jbyteArray (*Fn)(JNIEnv*, jobject, jintArray, jdoubleArray, jbyteArray, jlong, jintArray, jbyteArray, jbyteArray, jintArray,
                           jbyteArray, jintArray, jlongArray, jintArray, jlongArray);
  Fn fn = (Fn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailJNI", err, (size_t)errcap);
  if (!fn) return -1;
  std::vector<int64_t> zero((size_t)n2 + 1, 0);
  FObj* off = long_array(zero.data(), zero.size());
  if (async_copies > 0) {
    // the same group `async_copies` times through samPeTailSubmitJNI (fresh Java arrays per call, dropped right after it, as a task
    // thread's would be), collected in REVERSE order; every copy must come back with the same text and offsets
    typedef jlong (*SubFn)(JNIEnv*, jobject, jintArray, jdoubleArray, jbyteArray, jlong, jintArray, jbyteArray, jbyteArray, jintArray,
                           jbyteArray, jintArray, jlongArray, jintArray);
    typedef jbyteArray (*ColFn)(JNIEnv*, jobject, jlong, jlongArray);
    SubFn sub = (SubFn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailSubmitJNI", err, (size_t)errcap);
    ColFn col = (ColFn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailCollectJNI", err, (size_t)errcap);
    if (!sub || !col) return -1;
    std::vector<jlong> handles;
    for (int c = 0; c < async_copies; ++c) {
      const jlong h = sub(&e.env, J(self), (jintArray)J(int_array(opt_ints, 12)), (jdoubleArray)J(double_array(reals, 22)),
                          (jbyteArray)J(byte_array(reinterpret_cast<const uint8_t*>(mat), 25)), (jlong)id0, (jintArray)J(int_array(read_len, (size_t)n2)),
                          (jbyteArray)J(byte_array(reads, (size_t)reads_bytes)), quals ? (jbyteArray)J(byte_array(quals, (size_t)reads_bytes)) : nullptr,
                          (jintArray)J(int_array(name_len, (size_t)(n2 / 2))), (jbyteArray)J(byte_array(names, (size_t)names_bytes)),
                          (jintArray)J(int_array(reg_cnt, (size_t)n2)), (jlongArray)J(long_array(reg_longs, (size_t)(2 * n_regs))),
                          (jintArray)J(int_array(reg_ints, (size_t)(10 * n_regs))));
      if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
      if (h == 0) { snprintf(err, (size_t)errcap, "null handle"); return -1; }
      handles.push_back(h);
    }
    *out_bytes = -1;
    for (int c = async_copies - 1; c >= 0; --c) {
      FObj* offc = long_array(zero.data(), zero.size());
      jbyteArray r = col(&e.env, J(self), handles[(size_t)c], (jlongArray)J(offc));
      if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
      if (!r) { snprintf(err, (size_t)errcap, "null result"); return -1; }
      const int64_t nb = (int64_t)O(r)->bytes.size();
      if (nb > out_cap) { snprintf(err, (size_t)errcap, "out_cap too small"); return -1; }
      if (*out_bytes < 0) {
        *out_bytes = nb;
        memcpy(out_text, O(r)->bytes.data(), (size_t)nb);
        memcpy(out_off, offc->la.data(), 8 * ((size_t)n2 + 1));
      } else if (nb != *out_bytes || memcmp(out_text, O(r)->bytes.data(), (size_t)nb) != 0 || memcmp(out_off, offc->la.data(), 8 * ((size_t)n2 + 1)) != 0) {
        snprintf(err, (size_t)errcap, "copy %d came back with a different text", c);
        return -1;
      }
    }
    // a handle is good for one collect
    (void)col(&e.env, J(self), handles[0], (jlongArray)J(off));
    if (!vm.pending) { snprintf(err, (size_t)errcap, "a second collect of a handle did not throw"); return -1; }
    vm.pending = false;
    // a handle that will never be collected (a failed task): samPeTailCancelJNI waits for its group and drops it -- 1 the first time,
    // 0 for a handle that is gone (cancelled or collected), and a collect of it throws like any unknown handle
    typedef jint (*CanFn)(JNIEnv*, jobject, jlong);
    CanFn can = (CanFn)load_symbol(lib, "Java_cs_ucla_edu_bwaspark_jni_MateSWJNI_samPeTailCancelJNI", err, (size_t)errcap);
    if (!can) return -1;
    const jlong hx = sub(&e.env, J(self), (jintArray)J(int_array(opt_ints, 12)), (jdoubleArray)J(double_array(reals, 22)),
                         (jbyteArray)J(byte_array(reinterpret_cast<const uint8_t*>(mat), 25)), (jlong)id0, (jintArray)J(int_array(read_len, (size_t)n2)),
                         (jbyteArray)J(byte_array(reads, (size_t)reads_bytes)), quals ? (jbyteArray)J(byte_array(quals, (size_t)reads_bytes)) : nullptr,
                         (jintArray)J(int_array(name_len, (size_t)(n2 / 2))), (jbyteArray)J(byte_array(names, (size_t)names_bytes)),
                         (jintArray)J(int_array(reg_cnt, (size_t)n2)), (jlongArray)J(long_array(reg_longs, (size_t)(2 * n_regs))),
                         (jintArray)J(int_array(reg_ints, (size_t)(10 * n_regs))));
    if (vm.pending || hx == 0) { snprintf(err, (size_t)errcap, "submit before cancel: %s", vm.pending_msg.c_str()); return 1; }
    if (can(&e.env, J(self), hx) != 1 || vm.pending) { snprintf(err, (size_t)errcap, "cancel of a live handle did not return 1"); return -1; }
    if (can(&e.env, J(self), hx) != 0 || can(&e.env, J(self), handles[0]) != 0 || vm.pending) { snprintf(err, (size_t)errcap, "cancel of a gone handle did not return 0"); return -1; }
    (void)col(&e.env, J(self), hx, (jlongArray)J(off));
    if (!vm.pending) { snprintf(err, (size_t)errcap, "a collect of a cancelled handle did not throw"); return -1; }
    vm.pending = false;
    return 0;
  }
  jbyteArray r = fn(&e.env, J(self), (jintArray)J(int_array(opt_ints, 12)), (jdoubleArray)J(double_array(reals, 22)),
                    (jbyteArray)J(byte_array(reinterpret_cast<const uint8_t*>(mat), 25)), (jlong)id0, (jintArray)J(int_array(read_len, (size_t)n2)),
                    (jbyteArray)J(byte_array(reads, (size_t)reads_bytes)), quals ? (jbyteArray)J(byte_array(quals, (size_t)reads_bytes)) : nullptr,
                    (jintArray)J(int_array(name_len, (size_t)(n2 / 2))), (jbyteArray)J(byte_array(names, (size_t)names_bytes)),
                    (jintArray)J(int_array(reg_cnt, (size_t)n2)), (jlongArray)J(long_array(reg_longs, (size_t)(2 * n_regs))),
                    (jintArray)J(int_array(reg_ints, (size_t)(10 * n_regs))), (jlongArray)J(off));
  if (vm.pending) { snprintf(err, (size_t)errcap, "%s", vm.pending_msg.c_str()); return 1; }
  if (!r) { snprintf(err, (size_t)errcap, "null result"); return -1; }
  *out_bytes = (int64_t)O(r)->bytes.size();
  if (*out_bytes > out_cap) { snprintf(err, (size_t)errcap, "out_cap too small"); return -1; }
  memcpy(out_text, O(r)->bytes.data(), (size_t)*out_bytes);
  memcpy(out_off, off->la.data(), 8 * ((size_t)n2 + 1));
  return 0;
}

#define TAIL_ARGS_DECL const char* lib, int partition, const uint8_t* pac, int64_t l_pac, int n_seqs, const int64_t* ann_off,                      \
                         const int32_t* ann_len, const uint8_t* ann_names, int64_t ann_names_bytes, const int32_t opt_ints[12],                 \
                         const double reals[22], const int8_t mat[25], int64_t id0, int n2, const int32_t* read_len, const uint8_t* reads,      \
                         const uint8_t* quals, int64_t reads_bytes, const int32_t* name_len, const uint8_t* names, int64_t names_bytes,        \
                         const int32_t* reg_cnt, const int64_t* reg_longs, const int32_t* reg_ints, int64_t n_regs, uint8_t* out_text,          \
                         int64_t out_cap, int64_t* out_bytes, int64_t* out_off, char* err, int errcap
#define TAIL_ARGS lib, partition, pac, l_pac, n_seqs, ann_off, ann_len, ann_names, ann_names_bytes, opt_ints, reals, mat, id0, n2, read_len, reads, quals, \
                  reads_bytes, name_len, names, names_bytes, reg_cnt, reg_longs, reg_ints, n_regs, out_text, out_cap, out_bytes, out_off, err, errcap
int fake_jvm_sam_pe_tail(TAIL_ARGS_DECL) { return sam_pe_tail_impl(TAIL_ARGS, 0); }
// ... and through samPeTailSubmitJNI / samPeTailCollectJNI: `copies` groups in flight from this one thread
int fake_jvm_sam_pe_tail_async(TAIL_ARGS_DECL, int copies) { return sam_pe_tail_impl(TAIL_ARGS, copies); }

// name lookups (FindClass / GetFieldID / GetMethodID) the shim has made in this process so far, and live global references
long fake_jvm_lookups(void) { return g_world.lookups.load(); }
long fake_jvm_global_refs(void) { return g_world.global_refs.load(); }

// n task threads of one executor, thread t reporting Spark partition partitions[t], each pushing the same wire batch through
// swExtendFPGAJNI `calls` times.  outs: n x ret_task_num int16; info: n x 4 int64 = {partition seen, BPSW_DEVICES entry,
// HIP device, context handle} from bpsw_jni_thread_info on that thread.  Returns 0, or the first non-zero code of a thread.
int fake_jvm_extend_threads(const char* lib, int n, const int* partitions, const uint8_t* wire, int wire_bytes, int ret_task_num,
                            int calls, int16_t* outs, int64_t* info, char* err, int errcap) {
  std::vector<std::thread> th;
  std::vector<int> rc((size_t)n, 0);
  std::vector<std::string> errs((size_t)n);
  typedef uint64_t (*InfoFn)(int32_t*);
  char e0[256] = {0};
  InfoFn info_fn = (InfoFn)load_symbol(lib, "bpsw_jni_thread_info", e0, sizeof e0);
  if (!info_fn) { snprintf(err, (size_t)errcap, "%s", e0); return -1; }
  for (int t = 0; t < n; ++t)
    th.emplace_back([&, t] {
      char e[512] = {0};
      for (int c = 0; c < calls && rc[(size_t)t] == 0; ++c)
        rc[(size_t)t] = fake_jvm_extend(lib, partitions[t], wire, wire_bytes, ret_task_num, outs + (size_t)t * (size_t)ret_task_num, e, sizeof e);
      int32_t three[3] = {-2, -2, -2};
      const uint64_t ctx = info_fn(three);
      info[4 * t] = three[0]; info[4 * t + 1] = three[1]; info[4 * t + 2] = three[2]; info[4 * t + 3] = (int64_t)ctx;
      errs[(size_t)t] = e;
    });
  for (auto& x : th) x.join();
  for (int t = 0; t < n; ++t)
    if (rc[(size_t)t] != 0) { snprintf(err, (size_t)errcap, "thread %d: %s", t, errs[(size_t)t].c_str()); return rc[(size_t)t]; }
  return 0;
}

}  // extern "C"
