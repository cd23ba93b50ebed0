// fake_device.cpp -- TEST INFRASTRUCTURE (never shipped, never linked into libbPSW_hip.so): what the HOST layer of the library needs of
// a GPU, on a box without one, so that csrc/bpsw_runtime.cpp, bpsw_sw_runtime.cpp, bpsw_ring.cpp, bpsw_rescue.cpp, bpsw_pack.cpp,
// bpsw_tail.cpp, bpsw_tail_pool.cpp and bpsw_jni.cpp -- compiled UNCHANGED by g++ -- can run under -fsanitize=address,undefined / thread
// (round-5 review, item 6: ~7 k lines of threaded host C++ had only ever been checked by result equality).
//   * the HIP runtime entry points those files use, over ordinary memory: every "asynchronous" operation completes at once on the calling
//     thread, events are timestamps, streams carry no state -- except the stream of a submission ring, whose resident kernel is a
//     group of C++ threads (tests/ring_host/fake_ring_device.h) that stay alive until the epoch closes, as the real one does;
//   * the kernels behind the launch_* entry points, played by the ORACLE (oracle/bpsw_oracle.c): a rescue job is orc_sw_align2, an
//     extension batch orc_wire_extend.  The results are therefore the oracle's by construction: what a run of the parity tests against
//     this build checks is not the kernels (the GPU suite does that) but everything around them -- the JNI marshalling, the rescue
//     planner's speculation and replay with its hand-placed prefetches, the packers, the staging arithmetic, the rings' host half, the
//     tail pool's hand-offs -- for out-of-bounds accesses, use after free, undefined behaviour and data races.
//   * memRegToAln jobs are orc_reg2aln, a chain batch orc_chain2aln_batch, a reference fetch orc_bns_get_seq: worker2's tail (plan, emit,
//     the tail pool's threads) and the round-loop entry run too.  Coordinate batches are decoded here (the flanks looked up in the 2-bit
//     reference) and run through orc_extension.  What is not played -- the sift kernel's classification (side_how), the device-resident entries
//     -- is left out of the sanitizer run (tests/test_host_sanitizers.py names the tests).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "bpsw_internal.h"
#include "../ring_host/fake_ring_device.h"

extern "C" {
#include "bpsw_oracle.h"
}

using namespace bpsw;

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ------------------------------------------------------------------------------------------------ streams, events
struct FakeKernel {
  std::thread main;
  std::atomic<bool> finished{false};
  std::vector<struct ihipEvent_t*> end_events;
  std::mutex mu;
};
struct ihipStream_t {
  std::mutex mu;
  FakeKernel* k = nullptr;  // a resident ring kernel, if this is a ring's stream
};
struct ihipEvent_t {
  std::atomic<double> t_ms{0.};
  std::atomic<bool> ready{true};
};
static void stamp(hipEvent_t e) { if (e) { e->t_ms.store(now_ms()); e->ready.store(true, std::memory_order_release); } }
static void stream_drain(ihipStream_t* s) {  // (caller holds s->mu)
  if (!s->k) return;
  if (s->k->main.joinable()) s->k->main.join();
  delete s->k;
  s->k = nullptr;
}

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDeviceFlags(unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }  // (ring epochs are closed by ring_pause before anybody calls this)
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
  memset(p, 0, sizeof *p);
  snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
  p->multiProcessorCount = 256;
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 100000; return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipErrorNotSupported ? "not played by the fake device (tests/host_san)" : "fake device error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t s) {
  if (s) { std::lock_guard<std::mutex> lk(s->mu); stream_drain(s); }  // stream order behind a resident kernel
  memset(p, v, n);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new ihipStream_t; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new ihipStream_t; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) {
  if (s) { { std::lock_guard<std::mutex> lk(s->mu); stream_drain(s); } delete s; }
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  if (s) { std::lock_guard<std::mutex> lk(s->mu); stream_drain(s); }
  return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s) {
  if (!s) return hipSuccess;
  std::lock_guard<std::mutex> lk(s->mu);
  return (!s->k || s->k->finished.load(std::memory_order_acquire)) ? hipSuccess : hipErrorNotReady;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new ihipEvent_t; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new ihipEvent_t; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { return e->ready.load(std::memory_order_acquire) ? hipSuccess : hipErrorNotReady; }
hipError_t hipEventSynchronize(hipEvent_t e) {
  while (!e->ready.load(std::memory_order_acquire)) sched_yield();
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  if (!a->ready.load(std::memory_order_acquire) || !b->ready.load(std::memory_order_acquire)) return hipErrorNotReady;
  *ms = (float)(b->t_ms.load() - a->t_ms.load());
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  if (s) {
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->k) {
      std::lock_guard<std::mutex> lk2(s->k->mu);
      if (!s->k->finished.load(std::memory_order_acquire)) {
        e->ready.store(false, std::memory_order_relaxed);
        s->k->end_events.push_back(e);
        return hipSuccess;
      }
    }
  }
  stamp(e);
  return hipSuccess;
}
}

// ------------------------------------------------------------------------------------------------ the kernels, played by the oracle
static void unpack_mat(const MatRows& m, int8_t out[25]) {
  for (int r = 0; r < 5; ++r)
    for (int c = 0; c < 5; ++c) out[5 * r + c] = (int8_t)((m.row[r] >> (8 * c)) & 0xff);
}
static uint8_t ref_base(const uint8_t* pac, long long l_pac, long long p) {  // the doubled reference (util/BNTSeqUtil.scala:37-79)
  const auto at = [&](long long x) { return (uint8_t)((pac[x >> 2] >> ((~x & 3) << 1)) & 3); };
  return p < l_pac ? at(p) : (uint8_t)(3 - at(2 * l_pac - 1 - p));
}
static void play_sw_job(const uint8_t* q_pool, const uint8_t* t_pool, const uint8_t* pac, long long l_pac, long long q_off, long long t_off, int q_len,
                        int t_len, int q_rev, const int8_t mat[25], int a, int b, int o_del, int e_del, int o_ins, int e_ins, int xtra, int32_t out[7]) {
  std::vector<uint8_t> q((size_t)(q_len > 0 ? q_len : 1)), t((size_t)(t_len > 0 ? t_len : 1));
  for (int i = 0; i < q_len; ++i) {
    const uint8_t c = q_pool[q_off + (q_rev ? q_len - 1 - i : i)];
    q[(size_t)i] = q_rev ? (c < 4 ? (uint8_t)(3 - c) : (uint8_t)4) : c;  // MemSamPe.scala:1175-1184
  }
  for (int i = 0; i < t_len; ++i) t[(size_t)i] = t_pool ? t_pool[t_off + i] : ref_base(pac, l_pac, t_off + i);
  int64_t cells = 0;
  int32_t o[7];
  orc_sw_align2(q_len, q.data(), t_len, t.data(), 5, mat, a, b, o_del, e_del, o_ins, e_ins, xtra, o, &cells);
  for (int k = 0; k < 7; ++k) __atomic_store_n(out + k, o[k], __ATOMIC_RELAXED);  // (a system-scope store on the device)
}

namespace bpsw {

int sw_ring_class(const SwScoring&, int max_qlen, int max_tlen, int* bias_out) {
  if (bias_out) *bias_out = 0;
  if (max_tlen > 4000) return 0;  // (the resident kernel's key rows are fixed for the epoch: longer windows take a launch)
  return max_qlen <= 171 ? 3 : max_qlen <= 256 ? 5 : 0;
}
bool sw_quad_enabled() { return true; }
int sw_resident_waves(int num_cu) { return 4 * num_cu; }
size_t sw_scratch_bytes_per_wave(int max_tlen) { return 16 * (size_t)(max_tlen + 64); }
int global_resident_waves(int, int) { return 1024; }
int reg2aln_resident_waves(int, int, int, int) { return 1024; }
size_t reg2aln_lds_per_wave(int, int, int) { return 4096; }
int chain2aln_resident_waves(int) { return 1024; }

void launch_sw_prepass(const SwJobsDev& j, size_t q_pool_bytes, size_t t_pool_bytes, SwPrepass* pre, hipStream_t) {
  for (int i = 0; i < j.n; ++i) {
    const int ql = j.q_len[i], tl = j.t_len[i];
    const long long qo = j.q_off[i], to = j.t_off[i];
    const bool t_ok = j.t_pool ? (unsigned long long)(to + tl) <= t_pool_bytes : (to + tl <= (j.l_pac << 1) && (to >= j.l_pac || to + tl <= j.l_pac));
    if (ql < 1 || tl < 0 || qo < 0 || to < 0 || (unsigned long long)(qo + ql) > q_pool_bytes || !t_ok) pre->error = 1;
    if (ql > pre->max_qlen) pre->max_qlen = ql;
    if (tl > pre->max_tlen) pre->max_tlen = tl;
  }
}
hipError_t launch_sw_kernel(const SwJobsDev& j, const SwScoring& sc, int max_qlen, int max_tlen, int32_t* out, uint32_t*, int, hipStream_t,
                            const SwPrepass* pre, KernelEvents kev) {
  stamp(kev.start);
  if (!(pre && (pre->error || pre->max_qlen > max_qlen || pre->max_tlen > max_tlen))) {
    int8_t mat[25];
    unpack_mat(sc.mat, mat);
    for (int i = 0; i < j.n; ++i)
      play_sw_job(j.q_pool, j.t_pool, j.pac, j.l_pac, j.q_off[i], j.t_off[i], j.q_len[i], j.t_len[i], j.q_rev[i], mat, sc.a, sc.b, sc.o_del, sc.e_del,
                  sc.o_ins, sc.e_ins, sc.xtra, out + 7 * (size_t)i);
  }
  stamp(kev.stop);
  return hipSuccess;
}

// a resident kernel: poller + workers until the epoch closes (tests/ring_host/fake_ring_device.h)
template <class UnitFn>
static hipError_t launch_resident(const RingArgs& A, hipStream_t s, UnitFn fn) {
  std::lock_guard<std::mutex> lk(s->mu);
  stream_drain(s);
  FakeKernel* k = new FakeKernel;
  s->k = k;
  static const int n_workers = getenv("FAKE_RING_WORKERS") ? atoi(getenv("FAKE_RING_WORKERS")) : 4;
  k->main = std::thread([A, k, fn]() {
    std::vector<std::thread> ts;
    ts.emplace_back(fake_ring::play_poller, A);
    for (int i = 0; i < n_workers; ++i) ts.emplace_back([A, i, fn]() { fake_ring::play_worker(A, i, fn); });
    for (auto& t : ts) t.join();
    std::lock_guard<std::mutex> lk2(k->mu);
    for (ihipEvent_t* e : k->end_events) stamp(e);
    k->finished.store(true, std::memory_order_release);
  });
  return hipSuccess;
}
hipError_t launch_swp_resident(int, const RingArgs& A, int, hipStream_t s) {
  return launch_resident(A, s, [](const uint32_t* word, uint32_t unit) {  // bpsw_swalign.hip: swp_resident_kernel's reading of SwRingPayload
    SwRingPayload pl;
    memcpy(&pl, word + sizeof(RingDescHead) / 4, sizeof pl);
    int8_t mat[25];
    MatRows m;
    for (int r = 0; r < 5; ++r) m.row[r] = pl.mat_row[r];
    unpack_mat(m, mat);
    const uint32_t* packed = (const uint32_t*)(uintptr_t)pl.packed;
    // Fault injection for the integrity tripwire's own test (tests/test_host_sanitizers.py): FAKE_DEVICE_SW_FAULT="drop:n" -- the n-th
    // job pair this process's resident kernels take never writes its first record; "late:n" -- it writes it 2 ms after the pair was
    // counted as done (a completion word that overtook a result, as a violated PCIe ordering would look to the host)
    static const char* fault = getenv("FAKE_DEVICE_SW_FAULT");
    static std::atomic<long> units_seen{0};
    const long nth = units_seen.fetch_add(1) + 1;
    const bool hit = fault && atol(strchr(fault, ':') ? strchr(fault, ':') + 1 : "0") == nth;
    for (int job = 2 * (int)unit; job < 2 * (int)unit + 2 && job < pl.n_jobs; ++job) {
      const uint32_t* rec = packed + 8 * (size_t)job;
      long long qo, to;
      memcpy(&qo, rec, 8); memcpy(&to, rec + 2, 8);
      int32_t* dst = (int32_t*)(uintptr_t)pl.out + 7 * (size_t)job;
      int32_t held[7];
      const bool faulty = hit && job == 2 * (int)unit;
      play_sw_job((const uint8_t*)(uintptr_t)pl.q_pool, (const uint8_t*)(uintptr_t)pl.t_pool, (const uint8_t*)(uintptr_t)pl.pac, pl.l_pac, qo, to, (int)rec[4],
                  (int)rec[5], (int)rec[6], mat, pl.a, pl.b, pl.o_del, pl.e_del, pl.o_ins, pl.e_ins, pl.xtra, faulty ? held : dst);
      if (faulty && strncmp(fault, "late", 4) == 0) {
        std::thread([dst, held]() mutable {
          timespec ts = {0, 2000000};
          nanosleep(&ts, nullptr);
          for (int k = 0; k < 7; ++k) __atomic_store_n(dst + k, held[k], __ATOMIC_RELAXED);
        }).detach();
      }
    }
  });
}

// a coordinate batch (wire format 2, include/bpsw.h): 40-byte records, the query flanks as nibbles, the target flanks looked up in the
// 2-bit reference around the seed -- one orc_extension per task
static bool play_ext_coords(const uint8_t* w, int n, const int8_t mat[25], const ExtScoring& sc, int16_t* tmp) {
  if (!sc.pac || sc.l_pac <= 0) return false;
  const auto g16 = [&](size_t at) { return (int)(int16_t)(w[at] | (w[at + 1] << 8)); };
  const auto g32 = [&](size_t at) { uint32_t v; memcpy(&v, w + at, 4); return v; };
  std::vector<uint8_t> seq;
  for (int i = 0; i < n; ++i) {
    const size_t at = 32 + 40 * (size_t)i;
    orc_ext_param_t p;
    memset(&p, 0, sizeof p);
    p.o_del = (int8_t)w[0]; p.e_del = (int8_t)w[1]; p.o_ins = (int8_t)w[2]; p.e_ins = (int8_t)w[3];
    p.pen_clip5 = (int8_t)w[4]; p.pen_clip3 = (int8_t)w[5]; p.w = (int8_t)w[6];
    p.left_qlen = g16(at); p.left_rlen = g16(at + 2); p.right_qlen = g16(at + 4); p.right_rlen = g16(at + 6);
    const size_t pos = (size_t)g32(at + 8) * 4;
    p.reg_score = g16(at + 12); p.q_beg = g16(at + 14); p.h0 = g16(at + 16);
    const int seed_len = g16(at + 18);
    p.idx = (int32_t)g32(at + 28);
    long long rb;
    memcpy(&rb, w + at + 32, 8);
    p.zdrop = sc.zdrop; p.mat = mat;
    const int nq = p.left_qlen + p.right_qlen;
    seq.assign((size_t)(nq + p.left_rlen + p.right_rlen + 8), 0);
    for (int j = 0; j < nq; ++j) seq[(size_t)j] = (uint8_t)((g32(pos + 4 * (size_t)(j >> 3)) >> (28 - 4 * (j & 7))) & 0xF);
    uint8_t* lr = seq.data() + nq;
    uint8_t* rr = lr + p.left_rlen;
    for (int j = 0; j < p.left_rlen; ++j) lr[j] = ref_base(sc.pac, sc.l_pac, rb - 1 - j);
    for (int j = 0; j < p.right_rlen; ++j) rr[j] = ref_base(sc.pac, sc.l_pac, rb + seed_len + j);
    p.left_qs = seq.data(); p.right_qs = seq.data() + p.left_qlen; p.left_rs = lr; p.right_rs = rr;
    orc_ext_ret_t r;
    int64_t cells = 0;
    orc_extension(&p, sc.zdrop_mode, &r, &cells);
    int16_t* o = tmp + 10 * (size_t)i;  // MemChainToAlignBatched.scala:181-188
    o[0] = (int16_t)(r.idx & 0xffff); o[1] = (int16_t)((r.idx >> 16) & 0xffff);
    o[2] = (int16_t)r.q_beg; o[3] = (int16_t)r.q_end; o[4] = (int16_t)r.r_beg; o[5] = (int16_t)r.r_end;
    o[6] = (int16_t)r.score; o[7] = (int16_t)r.true_score; o[8] = (int16_t)r.width; o[9] = 0;
  }
  return true;
}
// ---- extension: a byte batch (wire format 1) through orc_wire_extend, a coordinate batch through play_ext_coords; every task, whatever
// list the launch was given
static hipError_t play_ext(const uint32_t* wire, int16_t* out, const ExtScoring& sc, int only_from, int only_to) {
  const uint8_t* w = (const uint8_t*)wire;
  int32_t n;
  memcpy(&n, w + 8, 4);
  int8_t mat[25];
  unpack_mat(sc.mat, mat);
  std::vector<int16_t> tmp(10 * (size_t)(n > 0 ? n : 1));
  int64_t cells = 0;
  if (w[7] == BPSW_WIRE_COORDS) {
    if (!play_ext_coords(w, n, mat, sc, tmp.data())) return hipErrorInvalidValue;
  } else if (orc_wire_extend(w, (size_t)1 << 40, mat, sc.zdrop, sc.zdrop_mode, tmp.data(), &cells) != n) return hipErrorInvalidValue;
  for (int t = only_from; t < only_to && t < n; ++t)
    for (int k = 0; k < 10; ++k) __atomic_store_n(out + (size_t)sc.out_stride * (size_t)t + k, tmp[10 * (size_t)t + k], __ATOMIC_RELAXED);
  return hipSuccess;
}
void launch_ext_prepass(const uint32_t*, size_t, int, ExtPrepass*, hipStream_t) {}  // (the asynchronous device entries are not played)
hipError_t launch_ext_kernel(const uint32_t* wire, int, int16_t* out, const ExtScoring& sc, int, int, int, int*, const int*, hipStream_t, const ExtPrepass*,
                             bool, KernelEvents kev, bool, int*, int, const uint8_t*, const uint4*, int* defer_post, const int*) {
  stamp(kev.start);
  const hipError_t e = play_ext(wire, out, sc, 0, 1 << 30);
  if (defer_post) *defer_post = 0;  // nothing deferred: the call launches no full kernel behind this one
  stamp(kev.stop);
  return e;
}
hipError_t launch_ext_sift_kernel(const uint32_t*, int, int16_t*, const ExtScoring&, int, int, uint8_t*, uint4*, hipStream_t, KernelEvents kev, const ExtPrepass*,
                                  int*, int*, int) {
  stamp(kev.start); stamp(kev.stop);  // (launch_ext_kernel above computes every task: the sift kernel has nothing to hand over)
  return hipSuccess;
}
// (the oracle computes a whole batch per call: the first unit of a descriptor to arrive does, the others copy their slice)
struct ExtBatchPlay {
  std::once_flag once;
  std::vector<int16_t> out;  // 10 per task; empty: the oracle refused the batch
  int units_left = 0;
};
static std::mutex g_ext_play_mu;
static std::map<std::pair<uint64_t, uint32_t>, std::shared_ptr<ExtBatchPlay>> g_ext_play;  // (completion record address, completion value)
hipError_t launch_ext_resident(const RingArgs& A, int, hipStream_t s) {
  return launch_resident(A, s, [](const uint32_t* word, uint32_t unit) {  // bpsw_extend.hip: ext_resident_kernel's reading of ExtRingPayload
    RingDescHead head;
    ExtRingPayload pl;
    memcpy(&head, word, sizeof head);
    memcpy(&pl, word + sizeof(RingDescHead) / 4, sizeof pl);
    std::shared_ptr<ExtBatchPlay> e;
    const auto key = std::make_pair(head.done_ptr, head.done_value);
    {
      std::lock_guard<std::mutex> lk(g_ext_play_mu);
      auto& slot = g_ext_play[key];
      if (!slot) { slot = std::make_shared<ExtBatchPlay>(); slot->units_left = (int)head.n_units; }
      e = slot;
      if (--slot->units_left == 0) g_ext_play.erase(key);
    }
    std::call_once(e->once, [&]() {
      int8_t mat[25];
      MatRows m;
      for (int r = 0; r < 5; ++r) m.row[r] = pl.mat_row[r];
      unpack_mat(m, mat);
      std::vector<int16_t> tmp(10 * (size_t)(pl.n_tasks > 0 ? pl.n_tasks : 1));
      int64_t cells = 0;
      if (pl.coord) {
        ExtScoring sc;
        memset(&sc, 0, sizeof sc);
        sc.zdrop = pl.zdrop; sc.zdrop_mode = pl.zdrop_mode; sc.pac = (const uint8_t*)(uintptr_t)pl.pac; sc.l_pac = pl.l_pac;
        if (play_ext_coords((const uint8_t*)(uintptr_t)pl.wire, pl.n_tasks, mat, sc, tmp.data())) e->out.swap(tmp);
      } else if (orc_wire_extend((const uint8_t*)(uintptr_t)pl.wire, (size_t)1 << 40, mat, pl.zdrop, pl.zdrop_mode, tmp.data(), &cells) == pl.n_tasks) e->out.swap(tmp);
    });
    if (e->out.empty()) return;
    static const char* xfault = getenv("FAKE_DEVICE_EXT_FAULT");  // "drop:n": the n-th extension-ring unit of the process leaves its records unwritten
    static std::atomic<long> xunits{0};
    if (xfault && atol(strchr(xfault, ':') ? strchr(xfault, ':') + 1 : "0") == xunits.fetch_add(1) + 1) return;
    int16_t* out = (int16_t*)(uintptr_t)pl.out;
    for (int t = (int)unit * pl.per_unit; t < ((int)unit + 1) * pl.per_unit && t < pl.n_tasks; ++t)
      for (int k = 0; k < 10; ++k) __atomic_store_n(out + (size_t)pl.out_stride * (size_t)t + k, e->out[10 * (size_t)t + k], __ATOMIC_RELAXED);
  });
}

void launch_ref_fetch(const uint8_t* pac, long long l_pac, int n, const long long* beg, const long long* end, uint8_t* out_pool, size_t out_pool_bytes, const long long* out_off,
                      long long* out_len, int* d_error, hipStream_t) {  // bnsGetSeq per window (util/BNTSeqUtil.scala:37-79)
  for (int i = 0; i < n; ++i) {
    const long long room = i + 1 < n ? out_off[i + 1] - out_off[i] : (long long)out_pool_bytes - out_off[i];
    const long long got = orc_bns_get_seq(l_pac, pac, beg[i], end[i], out_pool + out_off[i], room);
    if (got < 0) { if (d_error) *d_error = 1; out_len[i] = 0; } else out_len[i] = got;
  }
}
void launch_global_prepass(const GlobalJobsDev&, size_t, size_t, GlobalPrepass*, hipStream_t) {}
// SWGlobal + CIGAR per job through the oracle (bpsw_global.hip: score, operation count -- the true one also when it outgrows max_cigar --, operations)
hipError_t launch_global_kernel(const GlobalJobsDev& J, const SwScoring& sc, int, size_t, int32_t* score, int32_t* ncigar, uint32_t* cigar, uint8_t*, int, hipStream_t) {
  int8_t mat[25];
  unpack_mat(sc.mat, mat);
  std::vector<uint32_t> cg(1 << 14);
  for (int j = 0; j < J.n; ++j) {
    int nc = 0;
    score[j] = orc_sw_global(J.q_len[j], J.q_pool + J.q_off[j], J.t_len[j], J.t_pool + J.t_off[j], 5, mat, sc.o_del, sc.e_del, sc.o_ins, sc.e_ins, J.w[j], &nc, cg.data(),
                             (int)cg.size());
    ncigar[j] = nc;
    if (nc <= J.max_cigar) memcpy(cigar + (size_t)j * (size_t)J.max_cigar, cg.data(), 4 * (size_t)(nc > 0 ? nc : 0));
  }
  return hipSuccess;
}
// memRegToAln per job through the oracle (bpsw_reg2aln.hip: what the kernel leaves of a mem_aln_t, its CIGAR and its MD; counts are the
// true ones also when they outgrow the job's room -- the host resubmits with more)
hipError_t launch_reg2aln_kernel(const Reg2AlnDev& J, const SwScoring& sc, int, int, int, size_t, Reg2AlnOut* out, uint32_t* out_cigar, uint8_t* out_md, uint8_t*, int,
                                 hipStream_t) {
  orc_opt_t o;
  orc_opt_default(&o);
  o.a = J.a; o.b = sc.b; o.o_del = sc.o_del; o.e_del = sc.e_del; o.o_ins = sc.o_ins; o.e_ins = sc.e_ins; o.w = J.opt_w;
  unpack_mat(sc.mat, o.mat);
  orc_tail_opt_t t;
  orc_tail_opt_default(&t);
  std::vector<uint32_t> cig(1 << 14);
  std::vector<char> md(1 << 17);
  for (int j = 0; j < J.n; ++j) {
    orc_alnreg_t ar;
    static_assert(sizeof(orc_alnreg_t) == sizeof(bpsw_alnreg_t), "region record layouts");
    memcpy(&ar, J.regs + j, sizeof ar);
    orc_aln_t a;
    orc_reg2aln(&o, &t, J.n_seqs, (const int64_t*)J.ann_off, J.ann_len, J.l_pac, J.pac, J.read_len[j], J.read_pool + J.read_off[j], &ar, J.flavour, &a, cig.data(),
                (int)cig.size(), md.data(), (int)md.size());
    Reg2AlnOut k;
    memset(&k, 0, sizeof k);
    k.pos = a.pos; k.rid = a.rid; k.is_rev = a.is_rev; k.NM = a.NM; k.n_cigar = a.n_cigar; k.md_len = a.md_len; k.status = a.status;
    if (a.status == 1) { k.pos = -1; k.rid = -1; k.is_rev = 0; k.NM = 0; k.n_cigar = 0; k.md_len = 0; }  // BPSW_ALN_XREF: as the kernel leaves it
    out[j] = k;
    if (k.n_cigar <= J.max_cigar) memcpy(out_cigar + (size_t)j * (size_t)J.max_cigar, cig.data(), 4 * (size_t)(k.n_cigar > 0 ? k.n_cigar : 0));
    memcpy(out_md + (size_t)j * (size_t)J.max_md, md.data(), (size_t)(k.md_len < J.max_md ? (k.md_len > 0 ? k.md_len : 0) : J.max_md));
  }
  return hipSuccess;
}
// memChainToAlnBatched through the oracle: regions of read r in creation order from slot reg_base[r] on
hipError_t launch_chain2aln_kernel(const ChainBatchDev& B, const ChainParams& P, bpsw_alnreg_t* out_regs, int32_t* out_cnt, int32_t*, int, int, int32_t*, hipStream_t) {
  orc_opt_t o;
  orc_opt_default(&o);
  o.a = P.a; o.o_del = P.o_del; o.e_del = P.e_del; o.o_ins = P.o_ins; o.e_ins = P.e_ins; o.pen_clip5 = P.pen_clip5; o.pen_clip3 = P.pen_clip3;
  o.w = P.w; o.zdrop = P.zdrop;
  unpack_mat(P.mat, o.mat);
  long long seeds = 0, chains = 0;
  for (int r = 0; r < B.n_reads; ++r) chains += B.chain_cnt[r];
  for (long long c = 0; c < chains; ++c) seeds += B.seed_cnt[c];
  std::vector<orc_alnreg_t> regs((size_t)seeds + 16);
  std::vector<int32_t> cnt((size_t)(B.n_reads > 0 ? B.n_reads : 1));
  int64_t n_ext = 0, cells = 0;
  const int64_t tot = orc_chain2aln_batch(&o, P.zmode, B.l_pac, B.pac, B.n_reads, B.read_len, (const int64_t*)B.read_off, B.read_pool, B.chain_cnt, B.seed_cnt,
                                          (const int64_t*)B.seed_rbeg, B.seed_qbeg, B.seed_len, cnt.data(), regs.data(), (int64_t)regs.size(), &n_ext, &cells);
  if (tot < 0) return hipErrorInvalidValue;
  size_t at = 0;
  for (int r = 0; r < B.n_reads; ++r) {
    out_cnt[r] = cnt[(size_t)r];
    memcpy(out_regs + B.reg_base[r], regs.data() + at, sizeof(orc_alnreg_t) * (size_t)cnt[(size_t)r]);
    at += (size_t)cnt[(size_t)r];
  }
  return hipSuccess;
}

}  // namespace bpsw
