// bpsw_runtime.cpp -- context, arenas and the C ABI of libbPSW_hip.so (include/bpsw.h).
//
// Replaces the reference's accelerator host side: the SysV-shm + TCP hop of
// src/main/jni_fpga/sw_extend_fpga.c:116-193 and the OpenCL daemon src/main/alphadata/shm_host.c
// become an in-process HIP stream with persistent device arenas and pinned staging buffers.
#include <stdlib.h>
#include <string.h>

#include <sched.h>
#include <sys/prctl.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>

#include "bpsw_internal.h"

namespace bpsw {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
static int hip_fail(hipError_t e, const char* what) {
  return fail(BPSW_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);   \
  } while (0)

// hipFree / hipHostFree wait for everything on the device -- which, with a resident kernel of the submission ring that other threads keep
// feeding, is for as long as they keep coming: the open epochs are closed first (a few hundred microseconds) and the rings held until
// the memory is gone.
struct RingPauseForFree {
  int dev = 0;
  RingPauseForFree() { if (hipGetDevice(&dev) != hipSuccess) dev = 0; ring_pause(dev); }
  ~RingPauseForFree() { ring_resume(dev); }
};

hipError_t DeviceBuffer::reserve(size_t bytes) {
  if (bytes <= cap) return hipSuccess;
  size_t want = cap ? cap : 1 << 20;
  while (want < bytes) want <<= 1;
  if (ptr) { RingPauseForFree paused; (void)hipFree(ptr); }
  ptr = nullptr;
  cap = 0;
  hipError_t e = hipMalloc(&ptr, want);
  if (e == hipSuccess) cap = want;
  return e;
}
void DeviceBuffer::release() {
  if (ptr) { RingPauseForFree paused; (void)hipFree(ptr); }
  ptr = nullptr;
  cap = 0;
}
hipError_t PinnedBuffer::reserve(size_t bytes) {
  if (bytes <= cap) return hipSuccess;
  size_t want = cap ? cap : 1 << 20;
  while (want < bytes) want <<= 1;
  if (ptr) { RingPauseForFree paused; (void)hipHostFree(ptr); }
  ptr = nullptr;
  cap = 0;
  // (coarse-grained pinned memory, hipHostMallocNonCoherent, was tried for the zero-copy reads of the SW kernel: no difference)
  hipError_t e = hipHostMalloc(&ptr, want, hipHostMallocDefault);
  if (e == hipSuccess) cap = want;
  return e;
}
void PinnedBuffer::release() {
  if (ptr) { RingPauseForFree paused; (void)hipHostFree(ptr); }
  ptr = nullptr;
  cap = 0;
}

MatRows pack_mat(const int8_t mat[25]) {
  MatRows m;
  for (int k = 0; k < 5; ++k) {
    unsigned long long r = 0;
    for (int c = 0; c < 5; ++c) r |= (unsigned long long)(uint8_t)mat[k * 5 + c] << (8 * c);
    m.row[k] = r;
  }
  return m;
}

// The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), round robin, and streams that
// share a queue are serialised.  The device phases of the blocking entry points run on a pool of 20 streams (StreamLease), so
// the executor should be started with GPU_MAX_HW_QUEUES=20 in its environment (bench.py sets it before the
// runtime initialises).  The library does not set it itself: setenv from a library constructor inside a multi-threaded JVM
// races with every getenv of the process.  24 or more queues oversubscribe the hardware and throughput collapses.

bool tail_bound_enabled() {
  static const bool off = getenv("BPSW_EXT_TAIL") && atoi(getenv("BPSW_EXT_TAIL")) == 0;
  return !off;
}

bool certify_enabled() {
  static const bool off = getenv("BPSW_EXT_CERT") && atoi(getenv("BPSW_EXT_CERT")) == 0;
  return !off;
}
// 0: no certificate; 1: single gap (deficit below two gap opens); 2: also a deficit of two gap opens (+1), for matrices with
// match score 1 and every other entry <= -1 (flank_closed_form, "Two gap opens"); BPSW_EXT_CERT2=0 keeps level 1;
// 3: also flank_start_gap_form; BPSW_EXT_GAP1=0 keeps level 2
int certify_level(const int8_t mat[25]) {
  if (!certify_enabled()) return 0;
  static const bool off2 = getenv("BPSW_EXT_CERT2") && atoi(getenv("BPSW_EXT_CERT2")) == 0;
  if (off2 || exact_match_score(mat) != 1) return 1;
  for (int r = 0; r < 5; ++r)
    for (int c = 0; c < 5; ++c)
      if (!(r == c && r < 4) && mat[r * 5 + c] > -1) return 1;
  static const bool off3 = getenv("BPSW_EXT_GAP1") && atoi(getenv("BPSW_EXT_GAP1")) == 0;
  return off3 ? 2 : 3;  // 3: also the one-base gap at the start of a flank (flank_start_gap_form)
}

int exact_match_score(const int8_t mat[25]) {
  static const bool off = getenv("BPSW_EXT_EXACT") && atoi(getenv("BPSW_EXT_EXACT")) == 0;  // A/B switch for measurements
  if (off) return 0;
  const int a = mat[0];
  if (a <= 0) return 0;
  for (int r = 0; r < 5; ++r)
    for (int c = 0; c < 5; ++c) {
      const bool diag = r == c && r < 4;
      if (diag ? mat[r * 5 + c] != a : mat[r * 5 + c] >= a) return 0;
    }
  return a;
}

// The exact shortcuts of the extension (bpsw_extend_core.h) as the context's mask allows them: bit 0 closed form for near-exact
// flanks, 1 single-gap certificate, 2 two gap opens, 3 one-base gap at the start of a flank, 4 tail-row bound; the environment
// switches (BPSW_EXT_EXACT / CERT / CERT2 / GAP1 / TAIL = 0) and the scoring matrix can only take shortcuts away.
int sift_uniform_dm(const int8_t mat[25], int exact_a) {
  if (exact_a <= 0) return 0;
  const int mm = mat[1];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (i != j && mat[5 * i + j] != mm) return 0;
  return exact_a - mm > 0 ? exact_a - mm : 0;
}

void apply_shortcuts(int mask, const int8_t mat[25], int* exact_a, int* certify, int* tail_bound) {
  *exact_a = (mask & 1) ? exact_match_score(mat) : 0;
  int lvl = *exact_a > 0 ? certify_level(mat) : 0;
  const int allowed = !(mask & 2) ? 0 : (!(mask & 4) ? 1 : (!(mask & 8) ? 2 : 3));
  *certify = lvl < allowed ? lvl : allowed;
  *tail_bound = ((mask & 16) && tail_bound_enabled()) ? 1 : 0;
}

static void default_mat(int8_t mat[25], int a, int b) {  // bwaFillScmat, datatype/MemOptType.scala:58-73
  int k = 0;
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 4; ++j) mat[k++] = (int8_t)(i == j ? a : -b);
    mat[k++] = -1;
  }
  for (int j = 0; j < 5; ++j) mat[k++] = -1;
}

// BPSW_DEVICES="0,2,3" restricts and orders the devices contexts are spread over.
static std::vector<int> allowed_devices() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return {};
  std::vector<int> all;
  const char* env = getenv("BPSW_DEVICES");
  if (env && *env) {
    const char* p = env;
    while (*p) {
      char* q = nullptr;
      long v = strtol(p, &q, 10);
      if (q == p) break;
      if (v >= 0 && v < n) all.push_back((int)v);
      p = (*q == ',') ? q + 1 : q;
      if (*q != ',' && *q != 0) break;
    }
  }
  if (all.empty())
    for (int i = 0; i < n; ++i) all.push_back(i);
  return all;
}

bool spin_wait() {
  static const bool on = getenv("BPSW_SPIN_WAIT") && atoi(getenv("BPSW_SPIN_WAIT")) != 0;
  return on;
}

// Which small transfers bypass the copy engines: the kernels read / write pinned host memory directly over PCIe.  With sixteen
// streams busy the SDMA engines were the bottleneck of the host-buffer path -- 640 copies per million pairs, most of them a few
// hundred KB, each paying the engine's fixed cost and queueing behind the others (H2D 0.08 ms alone, 0.3-0.4 ms under load).
// bit 0: extension results, bit 1: SW results, bit 2: SW job table + sequences.  BPSW_ZEROCOPY=0 restores the copies.
int zerocopy_mask() {
  static const int m = getenv("BPSW_ZEROCOPY") ? atoi(getenv("BPSW_ZEROCOPY")) : 7;
  return m;
}

// Waiting for the device phase of a blocking call.  The runtime's own waits (hipStreamSynchronize, hipEventSynchronize, even
// with hipDeviceScheduleBlockingSync / hipEventBlockingSync) keep the calling thread on a CPU for most of a sub-millisecond
// wait (measured: 16 threads that wait two thirds of their time keep 14 CPUs busy), and an executor's task threads share a
// CPU quota.  So: sleep through most of the expected duration (a running average of this context's previous waits of the
// same kind), then poll the event, first yielding and -- if the device is late -- with short sleeps.  BPSW_SPIN_WAIT=1
// restores hipEventSynchronize.
// BPSW_WAIT_MODE (default 0; measured on the bench with 32 and 40 threads on 16 CPUs of quota: mode 1 is 1-2 % slower and uses the same CPU): how a caller passes the time until the device is done.  0: round 4's -- sleep through 70 % of the
// expected duration (less the default 50 us timer slack), then poll with sched_yield.  1: the waiting thread's timer slack is set to
// 1 us (prctl, per thread, once), it sleeps through 85 % of the expected duration and then polls with 8 us sleeps: with more task
// threads than the executor's CPU quota every yield-spin is CPU taken from a thread that has bytes to stage.
int wait_mode() {
  static const int m = getenv("BPSW_WAIT_MODE") ? atoi(getenv("BPSW_WAIT_MODE")) : 0;
  return m;
}
void wait_thread_setup() {
  static thread_local bool done = false;
  if (done) return;
  done = true;
  if (wait_mode() == 1) (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
}
// The running estimate of a kind of wait, from which the next nap is taken (70 % of it): an average of the waits' lengths in which ONE
// wait cannot raise it beyond 3 x + 0.2 ms.  Without that bound (rounds 4-5) a single wait that a descheduled thread stretches to 48 ms
// would make the estimate 12 ms, and since a wait that ends with its nap is as long as the nap, the estimate then falls by only 7.5 %
// per call: calls whose kernels take 0.04 ms would sleep 8.4, 7.8, 7.2 ... ms (tests/test_host_logic.py plays it through).  With the
// bound the same outlier costs 0.08 ms of naps.
// (An estimator that also SHRINKS when a wait ends with its nap was tried: under the bench, where the executor's CPU quota is the
// limit, it is bistable -- 2.49 or 1.92 x 10^8 reads/s from run to run: shorter naps are more polling, more polling is CPU the other
// threads do not get.  The averaged lengths err on the side of sleeping, which is the right side there.)
double wait_est_update(double est, double took_ms, int polls, bool napped) {
  (void)polls; (void)napped;
  if (est <= 0.) return took_ms;
  return 0.75 * est + 0.25 * std::min(took_ms, 3.0 * est + 0.2);
}
// (would wait_nap sleep for this estimate?  the same thresholds as below)
bool wait_naps(double est_ms) { return wait_mode() == 1 ? est_ms * 850.0 - 5.0 > 5.0 : (est_ms > 0.15 && est_ms * 700.0 - 60.0 > 20.0); }

void wait_nap(double est_ms) {
  if (wait_mode() == 1) {
    wait_thread_setup();
    const double nap_us = est_ms * 850.0 - 5.0;
    if (nap_us > 5.0) { timespec ts = {0, (long)(nap_us * 1000.0)}; nanosleep(&ts, nullptr); }
    return;
  }
  if (est_ms > 0.15) {
    const double nap_us = est_ms * 700.0 - 60.0;  // 70 % of the estimate, less the kernel's default timer slack
    if (nap_us > 20.0) { timespec ts = {0, (long)(nap_us * 1000.0)}; nanosleep(&ts, nullptr); }
  }
}
void wait_poll_pause(int polls, double waited_ms, double est_ms) {
  if (wait_mode() == 1) {
    timespec ts = {0, waited_ms < est_ms * 2.0 + 0.1 ? 8000 : 20000};
    nanosleep(&ts, nullptr);
    return;
  }
  if (polls < 64 && waited_ms < est_ms * 1.3 + 0.05) sched_yield();
  else { timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
}

hipError_t wait_event(bpsw_ctx* c, hipEvent_t ev, int kind) {
  if (spin_wait()) return hipEventSynchronize(ev);
  double& est = c->wait_est_ms[kind & 3];
  const double t0 = wall_ms();
  const bool napped = wait_naps(est);
  wait_nap(est);
  hipError_t e;
  int polls = 0;
  while ((e = hipEventQuery(ev)) == hipErrorNotReady) wait_poll_pause(++polls, wall_ms() - t0, est);
  est = wait_est_update(est, wall_ms() - t0, polls, napped);
  return e;
}

// the clock of the per-phase statistics (bpsw_stats_t: *_ms): wall time, or -- BPSW_STATS_CLOCK=cpu, a diagnostic -- the calling thread's
// CPU time, which says what a phase costs the executor's CPU quota rather than how long it lasts
double stat_ms() {
  static const bool cpu = getenv("BPSW_STATS_CLOCK") && getenv("BPSW_STATS_CLOCK")[0] == 'c';
  if (!cpu) return wall_ms();
  timespec ts;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

double wall_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

// ---- the per-device stream pool behind StreamLease (bpsw_internal.h) -------------------------------------------------
namespace {
struct StreamPool {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<hipStream_t> streams;
  std::vector<int> pending;  // calls that have taken the stream and not finished waiting for their work
};
StreamPool& stream_pool(int device) {
  static StreamPool table[64];
  return table[device >= 0 && device < 64 ? device : 0];
}
int stream_pool_cap() {
  static const int cap = [] {
    const char* e = getenv("BPSW_STREAM_POOL");
    int v = e ? atoi(e) : 20;
    v = v < 0 ? 0 : (v > 64 ? 64 : v);
    // The runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and serialises the streams that
    // share one: more pooled streams than queues buys nothing, so the pool is clamped to the queue count -- and an executor that was
    // started without the variable is told once, because it runs at a fraction of the rate it could.
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    const int queues = q && atoi(q) > 0 ? atoi(q) : 4;
    // (the library reads the ENVIRONMENT, not what the runtime picked up: the variable must be exported before the process starts --
    // a JVM or a Python that sets it after HIP has initialised gets a pool larger than its queues.  BPSW_STREAM_POOL_FORCE=1 keeps the
    // requested size whatever the variable says.)
    const char* force = getenv("BPSW_STREAM_POOL_FORCE");
    if (v > queues && !(force && atoi(force) != 0)) {
      fprintf(stderr, "bPSW: stream pool clamped from %d (%s) to %d: %s, so only that many calls run on the GPU at a time; start the "
                      "executor with GPU_MAX_HW_QUEUES=20 (two executors per GPU: 10 each), or set BPSW_STREAM_POOL_FORCE=1 to keep %d\n",
              v, e ? "BPSW_STREAM_POOL" : "the default", queues,
              q ? "GPU_MAX_HW_QUEUES gives this process fewer hardware queues" : "GPU_MAX_HW_QUEUES is not set and the HIP runtime's default is 4 hardware queues", v);
      v = queues;
    }
    return v;
  }();
  return cap;
}
// BPSW_STREAM_SHARE (default 1; 2 and 3 measured no better: 124-128 against 130 M reads/s, the device is the limit): how many calls may have work on one pooled stream at a time.  With 1 a stream is handed over
// only after its holder has seen its work complete, and sits idle for the hand-over (wake-up of the waiter, its enqueue); with 2
// the next call's copies and kernel are already queued behind the running one, in stream order, so the stream never drains
// while callers are waiting.  The number of busy streams -- what the hardware queues limit -- is the same.
int stream_share() {
  static const int n = [] {
    const char* e = getenv("BPSW_STREAM_SHARE");
    const int v = e ? atoi(e) : 1;
    return v < 1 ? 1 : (v > 8 ? 8 : v);
  }();
  return n;
}
}  // namespace

// when the last extension call of any context of the device began or ended (wall_ms): the ring sizes an epoch's worker grid by whether
// the rescue path has the device to itself (bpsw_ring.cpp)
static std::atomic<long long> g_ext_last_us[64];
void ext_call_mark(int device) { g_ext_last_us[device >= 0 && device < 64 ? device : 0].store((long long)(wall_ms() * 1e3), std::memory_order_relaxed); }
double ext_call_age_ms(int device) {
  const long long t = g_ext_last_us[device >= 0 && device < 64 ? device : 0].load(std::memory_order_relaxed);
  return t == 0 ? 1e12 : wall_ms() - (double)t * 1e-3;
}

CopyLane& copy_lane(int device) {
  static CopyLane table[64];
  return table[device >= 0 && device < 64 ? device : 0];
}

StreamLease::StreamLease(bpsw_ctx* c) : device(c->device), s(c->stream), pooled(false), wait_ms(0.) {
  const int cap = stream_pool_cap();
  if (cap == 0) return;
  StreamPool& P = stream_pool(device);
  const double t0 = wall_ms();
  std::unique_lock<std::mutex> lk(P.mu);
  for (;;) {
    int best = -1;
    for (size_t k = 0; k < P.streams.size(); ++k)
      if (best < 0 || P.pending[k] < P.pending[(size_t)best]) best = (int)k;
    if (best >= 0 && P.pending[(size_t)best] == 0) { slot = best; break; }  // an idle stream
    if ((int)P.streams.size() < cap) {
      hipStream_t ns = nullptr;
      if (hipStreamCreateWithFlags(&ns, hipStreamNonBlocking) == hipSuccess) {
        P.streams.push_back(ns);
        P.pending.push_back(0);
        slot = (int)P.streams.size() - 1;
        break;
      }
      if (best < 0) { wait_ms = wall_ms() - t0; return; }  // no pooled stream at all: the context's own stream for this call
    }
    if (best >= 0 && P.pending[(size_t)best] < stream_share()) { slot = best; break; }  // queue behind the least loaded stream
    P.cv.wait(lk);
  }
  ++P.pending[(size_t)slot];
  s = P.streams[(size_t)slot];
  pooled = true;
  wait_ms = wall_ms() - t0;
}
StreamLease::~StreamLease() {
  if (!pooled) return;
  StreamPool& P = stream_pool(device);
  {
    std::lock_guard<std::mutex> lk(P.mu);
    --P.pending[(size_t)slot];
  }
  P.cv.notify_one();
}

}  // namespace bpsw

using namespace bpsw;

extern "C" {

const char* bpsw_last_error(void) { return g_err.c_str(); }
// (diagnostics, not part of include/bpsw.h: the wait estimator's update rule for tests/test_host_logic.py -- no device needed)
double bpsw_diag_wait_est_update(double est, double took_ms, int polls, int napped) { return bpsw::wait_est_update(est, took_ms, polls, napped != 0); }
int bpsw_diag_wait_naps(double est_ms) { return bpsw::wait_naps(est_ms) ? 1 : 0; }

const char* bpsw_version(void) {
  return "bPSW-hip 0.5 (gfx950)";  // 0.5 = round 5: bpsw_stats_t grew (sw_ring_calls, ext_ring_calls): rebuild callers against include/bpsw.h
}

int bpsw_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int bpsw_device_slots(void) { return (int)allowed_devices().size(); }
int bpsw_device_for_partition(int partition) {
  if (partition < 0) return -1;
  const std::vector<int> devs = allowed_devices();
  return devs.empty() ? -1 : devs[(size_t)partition % devs.size()];
}

void bpsw_opt_default(bpsw_opt_t* o) {  // datatype/MemOptType.scala:28-73
  memset(o, 0, sizeof *o);
  o->a = 1; o->b = 4; o->o_del = 6; o->e_del = 1; o->o_ins = 6; o->e_ins = 1;
  o->pen_unpaired = 17; o->pen_clip5 = 5; o->pen_clip3 = 5; o->w = 100; o->zdrop = 100;
  o->T = 30; o->flag = 0; o->min_seed_len = 19; o->max_ins = 10000; o->max_matesw = 100;
  o->mask_level_redun = 0.95f;
  default_mat(o->mat, o->a, o->b);
}

int bpsw_create(int device, bpsw_ctx_t** out) {
  if (!out) return fail(BPSW_ERR_ARG, "bpsw_create: null out");
  *out = nullptr;
  std::vector<int> devs = allowed_devices();
  if (devs.empty()) return fail(BPSW_ERR_DEVICE, "bpsw_create: no HIP device visible (there is no CPU fallback)");
  if (device < 0) {
    static std::atomic<unsigned> rr{0};
    device = devs[rr.fetch_add(1) % devs.size()];
  } else {
    int n = bpsw_device_count();
    if (device >= n) return fail(BPSW_ERR_ARG, "bpsw_create: device index out of range");
  }
  HIP_TRY(hipSetDevice(device));
  if (!spin_wait()) {
    // Waiting host threads sleep until the device interrupts instead of spinning on the completion signal (the runtime's
    // default when CPUs outnumber contexts): an executor's task threads share a CPU quota, and on this path two thirds of
    // a call is waiting for the device.  Process-wide for this device; an error (flags already fixed by the host program)
    // only means the host program's choice stands.
    (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    (void)hipGetLastError();
  }
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(BPSW_ERR_DEVICE, std::string("bpsw_create: kernels are built for gfx950, device is ") + prop.gcnArchName);
  bpsw_ctx* c = new bpsw_ctx();
  c->device = device;
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  memset(&c->stats, 0, sizeof c->stats);
  default_mat(c->ext_mat, 1, 4);
  c->ext_sc.mat = pack_mat(c->ext_mat);
  c->ext_sc.zdrop = 100;
  c->ext_sc.zdrop_mode = BPSW_ZDROP_SCALA;
  c->ext_sc.mat_max = 1;
  c->ext_sc.side_how = nullptr;
  c->ext_sc.out_stride = 10;
  c->ext_sc.pac = nullptr; c->ext_sc.l_pac = 0;
  apply_shortcuts(c->shortcut_mask, c->ext_mat, &c->ext_sc.exact_a, &c->ext_sc.certify, &c->ext_sc.tail_bound);
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  // the events the blocking entry points wait on put the calling thread to sleep (interrupt-driven) instead of spinning:
  // an executor's task threads share a CPU quota, and a spinning waiter takes it from the threads doing host work
  // (BPSW_SPIN_WAIT=1 restores the runtime's default busy wait for A/B runs)
  for (int i = 0; e == hipSuccess && i < 8; ++i) e = hipEventCreateWithFlags(&c->ev[i], spin_wait() ? hipEventDefault : hipEventBlockingSync);
  if (e == hipSuccess) e = c->d_pre.reserve(512);
  // scan records and the self-resetting queue heads of ext_kernel.  On the context's own stream and WAITED FOR: rounds 1-5 had a plain
  // hipMemset here, which for device memory is a fill kernel on the legacy stream that the host does not wait for -- and this library's
  // streams are non-blocking, so nothing ordered it before the context's first kernel.  When the legacy stream's hardware queue is
  // busy (eight threads creating contexts beside running calls) the fill can land in the middle of the first extension kernel, clear
  // its queue heads and `done` count, so that the kernel's last wave never puts them back and a later call of the context starts from
  // stale heads: records missing from its result.  (The reading of one failure of test_concurrent_contexts_give_identical_results in
  // round 6's GPU runs -- found in the code, not reproduced on demand)
  if (e == hipSuccess) e = hipMemsetAsync(c->d_pre.ptr, 0, 512, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = c->h_pre.reserve(512);
  if (e == hipSuccess) memset(c->h_pre.ptr, 0, 512);  // (+448: the completion record of this context's ring submissions)
  if (e != hipSuccess) {
    bpsw_destroy(c);
    return hip_fail(e, "bpsw_create");
  }
  *out = c;
  return BPSW_OK;
}

void bpsw_destroy(bpsw_ctx_t* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->pend_ext.active && c->pend_ext.s) (void)hipStreamSynchronize(c->pend_ext.s);  // a caller-provided stream may still run
  if (c->pend_sw.active && c->pend_sw.s) (void)hipStreamSynchronize(c->pend_sw.s);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->ring_abandoned) {
    // a ring batch of this context ran into the watchdog (or lost records): its descriptor still names these blocks, and a unit that
    // finishes late would write into whatever they have become -- they stay allocated for the life of the process (a few MB, once)
    for (int i = 0; i < 8; ++i)
      if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    rescue_scratch_free(c->rescue_scratch);
    c->ev[0] = nullptr;  // (the buffers' owners are leaked with the context object itself)
    return;
  }
  // ONE pause of the device's rings around all the releases (each release pauses them for itself otherwise: a dozen close / synchronise /
  // relaunch rounds that every ring user of the device sits through whenever a task thread or a tail-pool worker goes away; the pause nests)
  RingPauseForFree paused_for_all;
  c->d_wire.release(); c->d_out.release(); c->d_pre.release();
  c->d_sw_in.release(); c->d_sw_out.release(); c->d_sw_scratch.release(); c->d_gl_z.release(); c->d_ext_lists.release(); c->d_sift.release();
  c->h_stage_in.release(); c->h_stage_out.release(); c->h_pre.release();
  rescue_scratch_free(c->rescue_scratch);
  for (int i = 0; i < 8; ++i)
    if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int bpsw_device_of(const bpsw_ctx_t* c) { return c ? c->device : -1; }

int bpsw_set_ext_shortcuts(bpsw_ctx_t* c, int mask) {
  if (!c) return fail(BPSW_ERR_ARG, "null context");
  std::lock_guard<std::mutex> g(c->mu);
  c->shortcut_mask = mask < 0 ? 63 : (mask & 63);
  apply_shortcuts(c->shortcut_mask, c->ext_mat, &c->ext_sc.exact_a, &c->ext_sc.certify, &c->ext_sc.tail_bound);
  return BPSW_OK;
}

int bpsw_set_ext_scoring(bpsw_ctx_t* c, const int8_t mat[25], int zdrop, int zdrop_mode) {
  if (!c) return fail(BPSW_ERR_ARG, "null context");
  if (zdrop_mode != BPSW_ZDROP_SCALA && zdrop_mode != BPSW_ZDROP_BWA) return fail(BPSW_ERR_ARG, "bad zdrop_mode");
  std::lock_guard<std::mutex> g(c->mu);
  if (mat) memcpy(c->ext_mat, mat, 25);
  c->ext_sc.mat = pack_mat(c->ext_mat);
  c->ext_sc.mat_max = c->ext_mat[0];
  for (int k = 1; k < 25; ++k) c->ext_sc.mat_max = c->ext_mat[k] > c->ext_sc.mat_max ? c->ext_mat[k] : c->ext_sc.mat_max;
  c->ext_sc.zdrop = zdrop;
  c->ext_sc.zdrop_mode = zdrop_mode;
  apply_shortcuts(c->shortcut_mask, c->ext_mat, &c->ext_sc.exact_a, &c->ext_sc.certify, &c->ext_sc.tail_bound);
  return BPSW_OK;
}



// ------------------------------------------------------------------------------------- boundary 2
static inline int rd16(const uint8_t* b, size_t at) { return (int16_t)(b[at] | (b[at + 1] << 8)); }
static inline int rd32(const uint8_t* b, size_t at) {
  return (int32_t)((uint32_t)b[at] | ((uint32_t)b[at + 1] << 8) | ((uint32_t)b[at + 2] << 16) | ((uint32_t)b[at + 3] << 24));
}

// Host-side twin of ext_prepass_kernel: validates the table, returns the LDS capacities.
// Validates a wire batch on the host.  *coord: the batch is a coordinate batch (wire format 2: byte 7 of the header is 2, 40-byte
// records, query flanks only); l_pac is the length of the loaded reference (0: none), needed to check its coordinates.
static int scan_wire(const uint8_t* wire, size_t bytes, long long l_pac, int* n_out, int* maxq, int* maxr, bool* coord,
                     std::vector<int>* long_tasks, std::vector<int>* mid_tasks, int* maxr_short, int* n_mid_out) {
  if (!wire || bytes < 32 || (bytes & 3)) return fail(BPSW_ERR_ARG, "extend: wire batch shorter than its header or not word sized");
  const int n = rd32(wire, 8);
  const int fmt = wire[7];
  if (fmt != 0 && fmt != BPSW_WIRE_COORDS) return fail(BPSW_ERR_ARG, "extend: unknown wire format (header byte 7)");
  const bool co = fmt == BPSW_WIRE_COORDS;
  const size_t rec_bytes = co ? 40 : 32;
  if (n < 0 || 32 + rec_bytes * (size_t)n > bytes) return fail(BPSW_ERR_ARG, "extend: task table exceeds the buffer");
  if ((int8_t)wire[0] < 0 || (int8_t)wire[1] < 1 || (int8_t)wire[2] < 0 || (int8_t)wire[3] < 1)
    return fail(BPSW_ERR_ARG, "extend: gap opens must be >= 0 and gap extensions >= 1 (a zero extension divides by zero in SWUtil.scala:110-115)");
  if ((int8_t)wire[6] < 0) return fail(BPSW_ERR_ARG, "extend: negative band width (w travels as a signed byte: at most 127)");
  if (co && l_pac <= 0) return fail(BPSW_ERR_ARG, "extend: a coordinate batch needs the reference on the device (bpsw_ref_load)");
  const int wband = (int8_t)wire[6];
  int mq = 0, mr = 0, mrs = 0;
  const size_t words = bytes >> 2;
  // tasks for the full kernel (a query flank above 255 bases; every task when the gap costs rule out the register sweeps):
  // the short kernel serves the others (bpsw_extend.hip, ext_kernel<.., SHORT>); "mid" tasks (a flank of 128-255 bases) run
  // on its sliding window and may be deferred to the full kernel from the device
  const bool all_long = (int8_t)wire[2] + (int8_t)wire[3] <= 0;
  long_tasks->clear();
  mid_tasks->clear();
  int mid = 0;
  for (int t = 0; t < n; ++t) {
    const size_t at = 32 + rec_bytes * (size_t)t;
    const int lq = rd16(wire, at), lr = rd16(wire, at + 2), rq = rd16(wire, at + 4), rr = rd16(wire, at + 6);
    if (lq < 0 || lr < 0 || rq < 0 || rr < 0) return fail(BPSW_ERR_ARG, "extend: negative sequence length in task table");
    const long long pos = rd32(wire, at + 8);
    const long long w = co ? ((long long)lq + rq + 7) / 8 : ((long long)lq + lr + rq + rr + 7) / 8;
    if (pos < 8 + (long long)(rec_bytes / 4) * n || (unsigned long long)(pos + w) > words)
      return fail(BPSW_ERR_ARG, "extend: task sequence offset outside the buffer");
    if (lq > mq) mq = lq;
    if (rq > mq) mq = rq;
    const bool is_long = all_long || lq > 255 || rq > 255;
    const bool is_mid = !is_long && (lq > 127 || rq > 127);
    if (is_mid) { ++mid; mid_tasks->push_back(t); }
    if (is_long) long_tasks->push_back(t);
    int task_mr = 0;
    if (co) {
      // the flanks [rb - lr, rb) and [rb + len, rb + len + rr) must lie on one strand of the doubled reference, as the windows
      // of getMaxSpan do (MemChainToAlignBatched.scala:654-677)
      const int len = rd16(wire, at + 18);
      long long rb;
      memcpy(&rb, wire + at + 32, 8);
      // rb comes straight from the caller's bytes: range-check it before any arithmetic (lr, rr, len are int16, so lo / hi cannot
      // overflow afterwards; an rb near INT64_MAX used to wrap hi negative and pass every test below)
      if (rb < 0 || rb > (l_pac << 1)) return fail(BPSW_ERR_ARG, "extend: task window outside the reference or bridging its two strands");
      const long long lo = rb - lr, hi = rb + len + rr;
      if (len < 0 || lo < 0 || hi > (l_pac << 1) || (lo < l_pac && hi > l_pac))
        return fail(BPSW_ERR_ARG, "extend: task window outside the reference or bridging its two strands");
      const int sl = std::min(lr, lq + 2 * wband + 2), sr = std::min(rr, rq + 2 * wband + 2);  // what the kernel stages (ext_kernel<true>)
      task_mr = std::max(sl, sr);
    } else {
      task_mr = std::max(lr, rr);
    }
    if (task_mr > mr) mr = task_mr;
    if (!is_long && task_mr > mrs) mrs = task_mr;
  }
  if (mq > BPSW_EXT_MAX_QLEN || mr > BPSW_EXT_MAX_RLEN) return fail(BPSW_ERR_LIMIT, "extend: sequence longer than the kernel limit");
  *n_out = n; *maxq = mq; *maxr = mr; *coord = co; *maxr_short = mrs; *n_mid_out = mid;
  return BPSW_OK;
}

static int extend_batch_impl(bpsw_ctx_t* c, const uint8_t* wire, size_t wire_bytes, int16_t* out, size_t out_len, uint8_t* side_how,
                             const int16_t** out_view = nullptr);

int bpsw_extend_batch(bpsw_ctx_t* c, const uint8_t* wire, size_t wire_bytes, int16_t* out, size_t out_len) {
  return extend_batch_impl(c, wire, wire_bytes, out, out_len, nullptr);
}

int bpsw_extend_stage(bpsw_ctx_t* c, size_t bytes, uint8_t** buf) {
  if (!c || !buf) return fail(BPSW_ERR_ARG, "extend_stage: null argument");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  { int prc = finish_pending(c); if (prc != BPSW_OK) return prc; }
  // room for the batch and for the task list the launch plan may stage behind it (at most one int per 32-byte record)
  HIP_TRY(c->h_stage_in.reserve(bytes + bytes / 8 + 4096));
  c->staged_bytes = bytes;
  *buf = (uint8_t*)c->h_stage_in.ptr;
  return BPSW_OK;
}

int bpsw_extend_commit(bpsw_ctx_t* c, size_t wire_bytes, const int16_t** out, size_t* out_len) {
  if (!c || !out) return fail(BPSW_ERR_ARG, "extend_commit: null argument");
  const uint8_t* staged = nullptr;
  {  // the staging block and what bpsw_extend_stage was asked for, read under the context's lock; a commit of more bytes than were
     // staged would let the table scan read past the pinned block
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->h_stage_in.ptr || c->staged_bytes == 0) return fail(BPSW_ERR_ARG, "extend_commit: nothing staged (call bpsw_extend_stage first)");
    if (wire_bytes > c->staged_bytes || wire_bytes > c->h_stage_in.cap)
      return fail(BPSW_ERR_ARG, "extend_commit: more bytes than bpsw_extend_stage was asked for");
    staged = (const uint8_t*)c->h_stage_in.ptr;
    c->staged_bytes = 0;  // one commit per stage
  }
  const int rc = extend_batch_impl(c, staged, wire_bytes, nullptr, 0, nullptr, out);
  if (rc == BPSW_OK && out_len) *out_len = *out ? 10 * (size_t)rd32((const uint8_t*)c->h_stage_in.ptr, 8) : 0;
  return rc;
}

int bpsw_extend_batch_classify(bpsw_ctx_t* c, const uint8_t* wire, size_t wire_bytes, int16_t* out, size_t out_len, uint8_t* side_how) {
  if (!side_how) return fail(BPSW_ERR_ARG, "extend_classify: null side_how");
  return extend_batch_impl(c, wire, wire_bytes, out, out_len, side_how);
}

// wire == the context's pinned staging block (bpsw_extend_stage): the batch is already where the copy engine reads it, nothing is
// copied in.  out_view: the caller takes the results where the kernel wrote them (the pinned result block), nothing is copied out.
static int extend_batch_impl(bpsw_ctx_t* c, const uint8_t* wire, size_t wire_bytes, int16_t* out, size_t out_len, uint8_t* side_how,
                             const int16_t** out_view) {
  if (!c) return fail(BPSW_ERR_ARG, "null context");
  int n = 0, mq = 0, mr = 0;
  bool coord = false;
  std::lock_guard<std::mutex> g(c->mu);
  if (c->ring_abandoned) return fail(BPSW_ERR_DEVICE, "this context gave up a ring batch (watchdog / integrity): create a new one");
  HIP_TRY(hipSetDevice(c->device));
  const uint8_t* d_pac = nullptr;
  long long l_pac = 0;
  RefHold ref_hold;  // a coordinate batch reads the loaded reference: it stays put until the kernel has been waited for
  if (wire && wire_bytes >= 32 && wire[7] == BPSW_WIRE_COORDS) ref_hold = ref_snapshot(c, &d_pac, &l_pac);
  std::vector<int>& long_tasks = c->ext_long_tasks;
  int mr_short = 0;
  int n_mid = 0;
  std::vector<int>& mid_tasks = c->ext_mid_tasks;
  int rc = scan_wire(wire, wire_bytes, l_pac, &n, &mq, &mr, &coord, &long_tasks, &mid_tasks, &mr_short, &n_mid);
  if (rc != BPSW_OK) return rc;
  // "mid" tasks (a query flank of 128-255 bases) run on the short kernel like the others; only they can meet a band wider than its
  // 128-column window, in which case the kernel defers them to the full kernel on the device (bpsw_extend.hip)
  (void)mid_tasks;
  const bool any_mid = n_mid > 0;
  // Launch plan: the short kernel over the whole batch (it skips the long tasks: a flank above 255 bases) and the full kernel over
  // the list of long ones plus what the short kernel defers -- or the full kernel alone when most tasks are long or the split is
  // switched off (BPSW_EXT_SPLIT=0)
  static const bool split_on = !(getenv("BPSW_EXT_SPLIT") && atoi(getenv("BPSW_EXT_SPLIT")) == 0);
  const int n_long = (int)long_tasks.size();
  const bool use_short = split_on && 2 * (size_t)n_long <= (size_t)n;
  // Flanks of 128-255 bases (2x250 bp reads) used to meet bands wider than the short kernel's 128-column window: round 3 deferred such a
  // task to the full kernel, round 4 swept the side again with the slot sweep in a build of its own (ext_kernel<., 2>, "inline wide").
  // Since round 5 the adaptive sweep has a four-columns-per-lane phase (bpsw_extend_rows.h, rows_cpp4) and no band of a flank up to 255
  // bases leaves it: nothing is deferred on account of a wide band, one build serves every batch, and the full kernel is for what the
  // host lists (flanks above 255 bases).  (Round 4's second build went with the first profiles that showed it idle.)
  const bool expect_full = !use_short || n_long > 0;
  (void)any_mid;
  // behind the wire bytes: the full kernel's list as [count, task indices...]; the host stages its own entries and the count, the
  // short kernel appends (room for every task)
  const size_t list_off = (wire_bytes + 15) & ~(size_t)15;
  // The deferring short kernel ALWAYS has a list to defer to (round 4 gave it none when no task could be expected to defer, and
  // trapped if one did): a batch without mid tasks posts an empty list like any other, and the full kernel is launched behind it
  // only when it is not (lazy_full below) -- or unconditionally where the late launch is off (classify entry, BPSW_EXT_LAZY_FULL=0).
  const bool with_list = use_short;
  const size_t stage_bytes = with_list ? list_off + 4 * (1 + (size_t)n_long) : wire_bytes;
  const size_t dev_bytes = with_list ? list_off + 4 * (1 + (size_t)n) : wire_bytes;
  if (!out_view && (!out || out_len < 10 * (size_t)n)) return fail(BPSW_ERR_CAPACITY, "extend: result buffer smaller than 10*n int16");
  if (out_view) *out_view = nullptr;
  if (n == 0) return BPSW_OK;
  { int prc = finish_pending(c); if (prc != BPSW_OK) return prc; }
  const size_t out_bytes = 20 * (size_t)n;
  // The full kernel behind the short one, when the host has listed nothing for it (no flank above 255 bases) and few tasks could
  // end up on its list at all (2x150 bp reads: the flanks of 128-131 bases): launched only when the short kernel did defer
  // something -- its last wave posts the length of the list behind the results, the call looks after its wait and, if need be,
  // launches the full kernel and waits again.  An empty launch held the stream for 0.1-0.18 ms of a 1.6 ms call
  // (profiles/r03_trace_overlap.txt).  BPSW_EXT_LAZY_FULL=0: always launch it.
  static const bool lazy_on = !(getenv("BPSW_EXT_LAZY_FULL") && atoi(getenv("BPSW_EXT_LAZY_FULL")) == 0);
  const bool lazy_full = lazy_on && !side_how && use_short && with_list && n_long == 0;
  const bool use_full = expect_full || (with_list && !lazy_full);
  const size_t out_post_bytes = out_bytes + (lazy_full ? 4 : 0);
  // (32-byte result slots in the pinned buffer were tried: the 16 + 4 byte stores of a record then cost two write sectors each,
  // more fabric writes than back-to-back 20-byte records that merge in L2, and the host-side gather cost more than the memcpy)
  bool zc_slots = false;
  HIP_TRY(c->d_wire.reserve(dev_bytes));
  HIP_TRY(c->d_out.reserve(out_post_bytes));
  // The sift kernel in front of the short kernel (bpsw_extend_sift.hip: it examines the tasks whose flanks have at most 127 bases):
  // batches whose matrix has one mismatch score (both wire formats) and of whose tasks at most one in sixteen has a longer flank
  // (2x150 bp reads: the flanks of 128-131 bases; a batch of 2x250 bp reads would pay the launch for nothing).
  // BPSW_EXT_SIFT=0 switches it off (A/B runs).
  static const bool sift_on = !(getenv("BPSW_EXT_SIFT") && atoi(getenv("BPSW_EXT_SIFT")) == 0);
  // a lone small call is latency: the extra launch costs it 60-70 us and saves nothing it would notice (tests/small_call_table.py:
  // 61 tasks 0.108 -> 0.171 ms, 4 088 tasks 0.267 -> 0.298, 32 768 tasks 0.76 either way); BPSW_EXT_SIFT_MIN moves the threshold
  static const int sift_min = getenv("BPSW_EXT_SIFT_MIN") ? atoi(getenv("BPSW_EXT_SIFT_MIN")) : 8192;
  const int sift_dm = sift_uniform_dm(c->ext_mat, c->ext_sc.exact_a);
  const bool use_sift = sift_on && n >= sift_min && (c->shortcut_mask & 32) && use_short && 16 * (size_t)n_mid <= (size_t)n && sift_dm > 0;
  const size_t sift_rec_off = ((size_t)n + 15) & ~(size_t)15;
  // the sift kernel lists what it leaves to the short kernel, which takes its tickets from the list (bpsw_extend.hip): the tasks
  // whose unresolved flanks add up to heavy_min bases or more (and the ones the sift does not examine: a flank of 128 bases or
  // more) first.  BPSW_EXT_TODO=0: no list, the short kernel walks the batch; BPSW_EXT_HEAVY_MIN=0: one class.
  static const bool todo_on = !(getenv("BPSW_EXT_TODO") && atoi(getenv("BPSW_EXT_TODO")) == 0);
  static const int heavy_min = getenv("BPSW_EXT_HEAVY_MIN") ? atoi(getenv("BPSW_EXT_HEAVY_MIN")) : 96;
  const bool use_todo = use_sift && todo_on;
  const size_t todo_off = sift_rec_off + 32 * (size_t)n;
  if (use_sift) HIP_TRY(c->d_sift.reserve(todo_off + (use_todo ? 4 * (size_t)n : 0)));
  const bool staged = wire == (const uint8_t*)c->h_stage_in.ptr;
  if (staged && stage_bytes > c->h_stage_in.cap) return fail(BPSW_ERR_ARG, "extend_commit: the staged batch is larger than what bpsw_extend_stage was asked for");
  HIP_TRY(c->h_stage_in.reserve(stage_bytes));
  HIP_TRY(c->h_stage_out.reserve(zc_slots ? 32 * (size_t)n : out_post_bytes));
  if (lazy_full) *(volatile int*)((char*)c->h_stage_out.ptr + out_bytes) = 0;
  const double t_in = stat_ms();
  ext_call_mark(c->device);
  struct ExtMark { int d; ~ExtMark() { ext_call_mark(d); } } ext_mark{c->device};
  if (!staged) memcpy(c->h_stage_in.ptr, wire, wire_bytes);
  if (with_list) {  // rides on the same copy
    int* hl = (int*)((char*)c->h_stage_in.ptr + list_off);
    hl[0] = n_long;
    if (n_long) memcpy(hl + 1, long_tasks.data(), 4 * (size_t)n_long);
  }
  const double t_staged = stat_ms();
  double t_dev0, t_dev1, copy_first_ms = 0.;
  bool kernel_was_last = false, relaunched = false;
  // The submission ring (bpsw_ring.h, class RING_CLASS_EXT): a SMALL batch -- no task for the full kernel, too few for the sift kernel to
  // pay, every flank within the resident kernel's fixed LDS geometry -- becomes one descriptor of the device's resident extension kernel:
  // one bulk copy on the copy lane, no stream, no launch, no event; its tasks are taken by the same wave population as every other
  // thread's, and the call waits on a completion record in its own pinned block.  These are the calls a launch costs most: the
  // reference sends everything below -FPGASWExtThreshold tasks... above it, batches of a few dozen tasks in the later rounds of
  // memChainToAlnBatched (worker1/MemChainToAlignBatched.scala:471-615).  BPSW_EXT_RING=0 (or BPSW_RING=0): a launch per call, as before.
  bool via_ring = false;
  float ring_kernel_ms = 0.f;
  {
    static const bool ext_ring_on = !(getenv("BPSW_EXT_RING") && atoi(getenv("BPSW_EXT_RING")) == 0);
    // (up to 256 tasks.  A lone call of 63 / 126 / 253 tasks takes 0.084 / 0.086 / 0.09-0.15 ms through the ring against 0.108 / 0.105 /
    // 0.10-0.12 with a launch, one of 1 019 tasks 0.22 against 0.12 -- the epoch's grid is one wave per SIMD, a launch of its own fills
    // the device --; sixteen callers make 105 k / 68 k / 46 k calls/s against 33 k / 30 k / 26 k, four callers 46 k / 37 k / 22 k against
    // 32 k / 30 k / 22 k: tests/small_call_table.py, profiles/r05_small_calls*.txt)
    static const int ext_ring_max = getenv("BPSW_EXT_RING_MAX_TASKS") ? atoi(getenv("BPSW_EXT_RING_MAX_TASKS")) : 256;
    const bool eligible = ring_enabled() && ext_ring_on && use_short && n_long == 0 && !use_sift && !side_how && !zc_slots &&
                          (zerocopy_mask() & 1) != 0 && mq <= 255 && mr_short <= EXT_RING_RCAP && n <= ext_ring_max;
    if (eligible && ring_usable(c->device, RING_CLASS_EXT)) {
      t_dev0 = stat_ms();
      // Where the workers read the batch from: up to BPSW_EXT_RING_ZC_BYTES (64 KB: some 500 tasks of 2x150 bp reads) straight from the
      // pinned staging block -- a task's record and flanks are a handful of reads, the tasks are spread over a thousand waves, and a
      // copy's latency would be most of such a call (253 tasks: 0.091 ms zero-copy, 0.127 with a copy on the context's stream; sixteen
      // callers' copies on the device's ONE copy lane were the bound of the first version, 50 k calls/s) --; above, one bulk copy into
      // the context's device buffer on the context's own stream, waited for before the descriptor is published.
      static const size_t ring_zc_bytes = getenv("BPSW_EXT_RING_ZC_BYTES") ? (size_t)atoll(getenv("BPSW_EXT_RING_ZC_BYTES")) : 65536;
      const void* ring_wire = c->h_stage_in.ptr;
      if (wire_bytes > ring_zc_bytes) {
        HIP_TRY(hipMemcpyAsync(c->d_wire.ptr, c->h_stage_in.ptr, wire_bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        HIP_TRY(wait_event(c, c->ev[0], 2));
        ring_wire = c->d_wire.ptr;
        copy_first_ms = stat_ms() - t_dev0;
      }
      RingDesc desc;
      memset(&desc, 0, sizeof desc);
      RingDescHead head;
      memset(&head, 0, sizeof head);
      RingDone* done = (RingDone*)((char*)c->h_pre.ptr + 448);
      if (++c->ring_seq == 0) ++c->ring_seq;
      const int per_unit = n <= 2048 ? 1 : 4;  // tasks per ticket: one atomic per task is fine for a few hundred tasks
      head.n_units = (uint32_t)((n + per_unit - 1) / per_unit);
      head.done_value = c->ring_seq;
      head.done_ptr = (uint64_t)(uintptr_t)done;
      ExtRingPayload pl;
      memset(&pl, 0, sizeof pl);
      pl.wire = (uint64_t)(uintptr_t)ring_wire; pl.out = (uint64_t)(uintptr_t)c->h_stage_out.ptr;
      pl.n_tasks = n; pl.per_unit = per_unit; pl.out_stride = c->ext_sc.out_stride;
      pl.zdrop = c->ext_sc.zdrop; pl.zdrop_mode = c->ext_sc.zdrop_mode; pl.mat_max = c->ext_sc.mat_max; pl.exact_a = c->ext_sc.exact_a;
      pl.tail_bound = c->ext_sc.tail_bound; pl.certify = c->ext_sc.certify;
      pl.coord = coord ? 1 : 0; pl.pac = (uint64_t)(uintptr_t)(coord ? d_pac : nullptr); pl.l_pac = coord ? l_pac : 0;  // (ref_hold keeps the reference put)
      for (int r = 0; r < 5; ++r) pl.mat_row[r] = c->ext_sc.mat.row[r];
      memcpy(desc.w, &head, sizeof head);
      memcpy(desc.w + sizeof(RingDescHead) / 4, &pl, sizeof pl);
      ring_poison((uint32_t*)c->h_stage_out.ptr, 5, (size_t)n, 4);  // (idx and the width word of every record: the tripwire of bpsw_ring.cpp)
      int rc = ring_submit(c->device, RING_CLASS_EXT, c->num_cu, desc);
      if (rc != BPSW_OK && !ring_usable(c->device, RING_CLASS_EXT)) {
        // the epoch could not be started (nothing of this batch has reached a worker): this call and the later ones take launches of their own
        static std::atomic<bool> said{false};
        if (!said.exchange(true)) fprintf(stderr, "bPSW: the extension ring of device %d failed (%s); small extension batches are launched one by one from here on\n", c->device, bpsw_last_error());
        copy_first_ms = 0.;
      } else {
        if (rc != BPSW_OK) return rc;
        rc = ring_wait(c->device, RING_CLASS_EXT, done, c->ring_seq, &c->wait_est_ms[4]);
        if (rc == BPSW_RING_RELAUNCH) { copy_first_ms = 0.; goto ring_left; }  // (another thread's epoch launch failed; nobody will run this batch)
        if (rc != BPSW_OK) { c->ring_abandoned = true; return rc; }
        // every record written?  (also the net under a task the resident kernel could not finish -- it has no defer list: a record left
        // as it was would otherwise pass for a result)
        rc = ring_check((const uint32_t*)c->h_stage_out.ptr, 5, (size_t)n, 4, "extension");
        if (rc != BPSW_OK) {
          // still missing after ring_check's grace period: not a late write but a task the resident kernel left alone.  The launch path
          // below has the full kernel behind it and computes the whole batch again into the same block.
          static std::atomic<bool> said{false};
          if (!said.exchange(true)) fprintf(stderr, "bPSW: an extension batch came back from the ring with unwritten records; it is run again through a launch (%s)\n", bpsw_last_error());
          copy_first_ms = 0.;
          goto ring_left;
        }
        ring_kernel_ms = (float)((double)(done->t_done.load(std::memory_order_relaxed) - done->t_first.load(std::memory_order_relaxed)) / ring_ticks_per_ms(c->device, RING_CLASS_EXT));
        t_dev1 = stat_ms();
        via_ring = true;
        c->stats.ext_ring_calls++;
      }
    }
  ring_left:;
  }
  if (!via_ring) {
    // The copy of the wire batch.  While rescue batches are in flight on this device (bpsw_sw_runtime.cpp counts them) it is made
    // and waited for BEFORE the call takes a pooled stream, on the device's copy lane (one such copy at a time): the stream is then
    // held for the kernels only, and one bulk copy crosses PCIe at a time instead of several beside the rescue path's transfers.  configs[2]: the device phase of an extension call 1.07 -> 0.93 ms, the step +2.7 % (2.03 -> 2.09 x
    // 10^8 reads/s).  With no rescue launch about (an extension-only stream of batches moves 55 GB/s: the link's rate) the copy stays
    // on the call's own stream, where several are in flight: one at a time is 43 GB/s (3.2 instead of 4.0 x 10^8 reads/s on
    // configs[1]).  BPSW_EXT_H2D_FIRST=0 / 1: never / always.
    static const int h2d_first = getenv("BPSW_EXT_H2D_FIRST") ? atoi(getenv("BPSW_EXT_H2D_FIRST")) : -1;
    const bool copy_first = h2d_first == 1 || (h2d_first < 0 && sw_launches_in_flight(c->device) > 0);
    bool copy_on_lane = false;
    if (copy_first) {
      // one bulk copy at a time per device, on a stream of the library's own (round 4 used a blocking hipMemcpy on the legacy default
      // stream for the same effect: that synchronises with every BLOCKING stream of the host process -- torch's, another native
      // library's -- and busy-waits; here the waiting thread sleeps, and nobody else's stream is involved)
      const double t_c0 = stat_ms();
      CopyLane& L = copy_lane(c->device);
      // BPSW_EXT_COPY_WAIT: 0 (default) the runtime's own wait for the lane's stream, under the lane's lock -- what a blocking hipMemcpy
      // does, without the legacy stream; 1 the lock covers the enqueue only and every caller sleeps / polls on an event of its own
      // (wait_event): a fifth less CPU per extension call, but the wake-up comes late -- 1.2-1.4 ms per call's device phase instead of
      // 0.9, the bench step 5-10 % slower with 32, 40 or 48 threads; 2 nobody on the host waits: the call's stream does
      // (hipStreamWaitEvent) -- the step 2.0 instead of 2.45 x 10^8 reads/s: a dependency between two hardware queues costs the device
      // more than the spin costs the host.
      static const int copy_wait = getenv("BPSW_EXT_COPY_WAIT") ? atoi(getenv("BPSW_EXT_COPY_WAIT")) : 0;
      if (copy_wait == 0) {
        std::lock_guard<std::mutex> lk(L.mu);
        if (!L.s) HIP_TRY(hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking));
        HIP_TRY(hipMemcpyAsync(c->d_wire.ptr, c->h_stage_in.ptr, stage_bytes, hipMemcpyHostToDevice, L.s));
        HIP_TRY(hipStreamSynchronize(L.s));
      } else if (copy_wait == 2) {
        // the lane orders the copies (one at a time, in its stream's order); nobody on the host waits for one: the call's own stream
        // does (hipStreamWaitEvent below), and the kernels are queued behind it at once
        std::lock_guard<std::mutex> lk(L.mu);
        if (!L.s) HIP_TRY(hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking));
        HIP_TRY(hipMemcpyAsync(c->d_wire.ptr, c->h_stage_in.ptr, stage_bytes, hipMemcpyHostToDevice, L.s));
        HIP_TRY(hipEventRecord(c->ev[6], L.s));
        copy_on_lane = true;
      } else {
        {
          std::lock_guard<std::mutex> lk(L.mu);
          if (!L.s) HIP_TRY(hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking));
          HIP_TRY(hipMemcpyAsync(c->d_wire.ptr, c->h_stage_in.ptr, stage_bytes, hipMemcpyHostToDevice, L.s));
          HIP_TRY(hipEventRecord(c->ev[3], L.s));
        }
        HIP_TRY(wait_event(c, c->ev[3], 3));
      }
      copy_first_ms = stat_ms() - t_c0;  // (booked as the call's H2D time below)
    }
    StreamLease lease(c);  // a device stream for the device phase only (bpsw_internal.h)
    hipStream_t s = lease.s;
    t_dev0 = stat_ms();
    HIP_TRY(hipEventRecord(c->ev[0], s));
    if (copy_on_lane) HIP_TRY(hipStreamWaitEvent(s, c->ev[6], 0));
    if (!copy_first) HIP_TRY(hipMemcpyAsync(c->d_wire.ptr, c->h_stage_in.ptr, stage_bytes, hipMemcpyHostToDevice, s));
    bool on_dispatch = false;  // ev[1] / ev[2] ride on the kernel's own dispatch (KernelEvents, bpsw_internal.h)
    // results: written by the kernel straight into the pinned staging buffer (20 B per task, posted PCIe writes), or into
    // device memory and copied back
    const bool zc_out = (zerocopy_mask() & 1) != 0;
    int16_t* k_out = zc_out ? (int16_t*)c->h_stage_out.ptr : (int16_t*)c->d_out.ptr;
    ExtScoring c_sc_full = c->ext_sc;  // what a late launch of the full kernel gets (lazy_full)
    {
      ExtScoring sc = c->ext_sc;
      if (zc_slots) sc.out_stride = 16;
      if (coord) { sc.pac = d_pac; sc.l_pac = l_pac; }
      c_sc_full = sc;
      if (side_how) {  // diagnostics: the kernel notes per side whether a shortcut or the DP produced the result
        HIP_TRY(c->d_ext_lists.reserve(2 * (size_t)n + 16));
        HIP_TRY(hipMemsetAsync(c->d_ext_lists.ptr, 0, 2 * (size_t)n, s));
        sc.side_how = (uint8_t*)c->d_ext_lists.ptr;
      }
      on_dispatch = true;
      int* d_queue = (int*)((char*)c->d_pre.ptr + 128);
      int* d_list = with_list ? (int*)((char*)c->d_wire.ptr + list_off) : nullptr;
      if (use_short) {
        KernelEvents kev;
        kev.start = c->ev[1]; kev.stop = (use_full && !lazy_full) ? nullptr : c->ev[2];
        uint8_t* d_sflag = use_sift ? (uint8_t*)c->d_sift.ptr : nullptr;
        uint4* d_srecs = use_sift ? (uint4*)((char*)c->d_sift.ptr + sift_rec_off) : nullptr;
        int* d_todo = use_todo ? (int*)((char*)c->d_sift.ptr + todo_off) : nullptr;
        // (should anything fail between the sift kernel, which fills the to-do counters d_queue[2..3], and the short kernel, whose last
        // wave puts them back to zero, they are cleared here: the next call's sift kernel adds to what it finds)
        struct TodoGuard {
          int* q; hipStream_t s; bool armed;
          ~TodoGuard() { if (armed) { (void)hipMemsetAsync(q, 0, 4 * sizeof(int), s); (void)hipStreamSynchronize(s); } }
        } todo_guard{d_queue, s, false};
        if (use_sift) {  // the kernel time of the call starts with it
          KernelEvents sev;
          sev.start = kev.start; kev.start = nullptr;
          todo_guard.armed = use_todo;
          HIP_TRY(launch_ext_sift_kernel((const uint32_t*)c->d_wire.ptr, n, k_out, sc, sift_dm, 127, d_sflag, d_srecs, s, sev, nullptr,
                                         use_todo ? d_queue + 2 : nullptr, d_todo, heavy_min));
        }
        HIP_TRY(launch_ext_kernel((const uint32_t*)c->d_wire.ptr, n, k_out, sc, std::min(mq, 255), mr_short, c->num_cu, d_queue, nullptr, s,
                                  nullptr, false, kev, true, d_list, 255, d_sflag, d_srecs, lazy_full ? (int*)(k_out + 10 * (size_t)n) : nullptr, d_todo));
        todo_guard.armed = false;
      }
      if (use_full && !lazy_full) {
        KernelEvents kev;
        kev.start = use_short ? nullptr : c->ev[1]; kev.stop = c->ev[2];
        // behind the short kernel: the list it completed on the device (what it may have deferred is unknown to the host
        // -- a subset of the mid tasks, normally a small one: a quarter of them sizes the grid; an empty list costs a launch that
        // returns at once)
        // (a batch with few mid tasks -- 2x150 bp reads: the flanks of 128-131 bases -- defers next to nothing: one workgroup, whose
        // four persistent waves take whatever the list holds, gets its wave slots sooner than a grid sized for a quarter of them, and
        // the call holds its stream for that long)
        // (this is the launch that is NOT late -- the classify entry, BPSW_EXT_LAZY_FULL=0, or tasks the host listed itself: a quarter of
        // the mid tasks sizes it, at least one workgroup; the late launch below is sized by the posted length of the list)
        const bool may_defer = use_short;
        const int grid_tasks = !use_short ? n : n_long + (may_defer ? 4 : 0);  // (a wide band defers nothing since round 5: room for the unexpected only)
        HIP_TRY(launch_ext_kernel((const uint32_t*)c->d_wire.ptr, grid_tasks, k_out, sc, mq, mr, c->num_cu, d_queue, nullptr, s,
                                  nullptr, false, kev, false, use_short ? d_list : nullptr));
      }
      if (side_how) HIP_TRY(hipMemcpyAsync(side_how, c->d_ext_lists.ptr, 2 * (size_t)n, hipMemcpyDeviceToHost, s));
    }
    if (!on_dispatch) HIP_TRY(hipEventRecord(c->ev[2], s));
    const bool kernel_is_last = on_dispatch && zc_out && !side_how;  // then the call waits for the kernel's own stop event
    kernel_was_last = kernel_is_last;
    if (!zc_out) HIP_TRY(hipMemcpyAsync(c->h_stage_out.ptr, c->d_out.ptr, out_post_bytes, hipMemcpyDeviceToHost, s));
    if (!kernel_is_last) HIP_TRY(hipEventRecord(c->ev[3], s));
    HIP_TRY(wait_event(c, kernel_is_last ? c->ev[2] : c->ev[3], 0));  // the last operation of the call on this stream
    if (lazy_full && *(volatile int*)((char*)c->h_stage_out.ptr + out_bytes) > 0) {  // the short kernel left a list: the full kernel, now
      // (events of its own: the call's kernel time is the first launches' span plus this one's, not the host's round trip in between)
      KernelEvents kev;
      kev.start = c->ev[4]; kev.stop = c->ev[5];
      relaunched = true;
      const int listed = *(volatile int*)((char*)c->h_stage_out.ptr + out_bytes);
      HIP_TRY(launch_ext_kernel((const uint32_t*)c->d_wire.ptr, listed, k_out, c_sc_full, mq, mr, c->num_cu, (int*)((char*)c->d_pre.ptr + 128), nullptr, s,
                                nullptr, false, kev, false, (int*)((char*)c->d_wire.ptr + list_off)));
      if (!zc_out) HIP_TRY(hipMemcpyAsync(c->h_stage_out.ptr, c->d_out.ptr, out_bytes, hipMemcpyDeviceToHost, s));
      if (!kernel_is_last) HIP_TRY(hipEventRecord(c->ev[3], s));
      HIP_TRY(wait_event(c, kernel_is_last ? c->ev[5] : c->ev[3], 0));
      c->stats.ext_full_relaunches++;
    }
    t_dev1 = stat_ms();
    c->stats.ext_wait_ms += lease.wait_ms;
  }
  if (zc_slots) {  // gather the 20-byte records out of their 32-byte slots
    const uint8_t* src = (const uint8_t*)c->h_stage_out.ptr;
    uint8_t* dst = (uint8_t*)out;
    for (int t = 0; t < n; ++t) {
      memcpy(dst + 20 * (size_t)t, src + 32 * (size_t)t, 16);
      memcpy(dst + 20 * (size_t)t + 16, src + 32 * (size_t)t + 16, 4);
    }
  } else if (out_view) {
    *out_view = (const int16_t*)c->h_stage_out.ptr;  // valid until the next call on this context
  } else {
    memcpy(out, c->h_stage_out.ptr, out_bytes);
  }
  const double t_out = stat_ms();
  float a = 0, b = 0, d = 0;
  if (via_ring) {
    b = ring_kernel_ms;  // first task taken -> last task finished, on the device's clock
    t_dev0 += copy_first_ms;  // (booked once below, with the copy)
  } else {
    (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&b, c->ev[1], c->ev[2]);
    if (relaunched) { float b2 = 0; (void)hipEventElapsedTime(&b2, c->ev[4], c->ev[5]); b += b2; }
    if (!kernel_was_last) (void)hipEventElapsedTime(&d, relaunched ? c->ev[5] : c->ev[2], c->ev[3]);
  }
  c->stats.ext_calls++; c->stats.ext_tasks += (uint64_t)n; c->stats.ext_wire_bytes += wire_bytes;
  c->stats.ext_h2d_ms += a + copy_first_ms; c->stats.ext_kernel_ms += b; c->stats.ext_d2h_ms += d;
  c->stats.ext_host_in_ms += t_staged - t_in; c->stats.ext_dev_ms += t_dev1 - t_dev0 + copy_first_ms; c->stats.ext_host_out_ms += t_out - t_dev1;
  c->last_ext_ms = b;
  c->have_ext_ev = false;
  return BPSW_OK;
}

// Fixed LDS geometry of the asynchronous launch: covers every register-path task (sides <= 255 bases) and the longest
// reference flank the kernels accept; 29.8 KB per workgroup, i.e. 5 workgroups per CU -- what the VGPR budget allows anyway.
static const int ASYNC_QCAP = 256, ASYNC_RCAP = 4096;

// the synchronous geometry-dependent launch (batches that outgrow the async geometry; the experimental kernels)
static int ext_device_sync_launch(bpsw_ctx_t* c, const void* d_wire, size_t wire_bytes, int n_tasks, void* d_out, hipStream_t s,
                                  const ExtPrepass* h_pre, const int* h_counts) {
  HIP_TRY(hipEventRecord(c->ev[4], s));
  (void)wire_bytes; (void)h_counts;
  HIP_TRY(launch_ext_kernel((const uint32_t*)d_wire, n_tasks, (int16_t*)d_out, c->ext_sc, h_pre->max_qlen, h_pre->max_rlen, c->num_cu,
                            (int*)((char*)c->d_pre.ptr + 128), nullptr, s));
  HIP_TRY(hipEventRecord(c->ev[5], s));
  c->have_ext_ev = true;
  return BPSW_OK;
}

extern "C++" {
namespace bpsw {
// Resolves an asynchronous extension launch of this context: waits for it, reads the table scan, reports a malformed
// batch, and re-launches a batch that outgrew the speculative geometry.  Caller holds c->mu and has set the device.
int finish_pending_ext(bpsw_ctx_t* c) {
  if (!c->pend_ext.active) return BPSW_OK;
  bpsw_ctx::PendingExt p = c->pend_ext;
  c->pend_ext.active = false;
  HIP_TRY(hipStreamSynchronize(p.s));
  const ExtPrepass* h_pre = (const ExtPrepass*)c->h_pre.ptr;
  if (h_pre->error) { c->have_ext_ev = false; return fail(BPSW_ERR_ARG, "extend_device: malformed wire batch (code " + std::to_string(h_pre->error) + ")"); }
  if (h_pre->max_qlen > BPSW_EXT_MAX_QLEN || h_pre->max_rlen > BPSW_EXT_MAX_RLEN) {
    c->have_ext_ev = false;
    return fail(BPSW_ERR_LIMIT, "extend_device: sequence longer than the kernel limit");
  }
  c->ext_geom_q = std::max(c->ext_geom_q, h_pre->max_qlen);
  c->ext_geom_r = std::max(c->ext_geom_r, h_pre->max_rlen);
  if (h_pre->max_qlen > p.qcap || h_pre->max_rlen > p.rcap) {  // the kernel left the batch untouched: launch it for real
    int counts[3] = {0, 0, p.n_tasks};
    int rc = ext_device_sync_launch(c, p.d_wire, p.wire_bytes, p.n_tasks, p.d_out, p.s, h_pre, counts);
    if (rc != BPSW_OK) return rc;
    HIP_TRY(hipStreamSynchronize(p.s));
  }
  return BPSW_OK;
}
}  // namespace bpsw
}  // extern "C++"

int bpsw_extend_batch_device(bpsw_ctx_t* c, const void* d_wire, size_t wire_bytes, int n_tasks, void* d_out,
                             void* hip_stream) {
  if (!c) return fail(BPSW_ERR_ARG, "null context");
  if (!d_wire || !d_out || n_tasks < 0 || wire_bytes < 32 + 32 * (size_t)n_tasks || (wire_bytes & 3) ||
      ((uintptr_t)d_wire & 15) || ((uintptr_t)d_out & 3))
    return fail(BPSW_ERR_ARG, "extend_device: bad buffer arguments");
  if (n_tasks == 0) return BPSW_OK;
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  { int rc = finish_pending(c); if (rc != BPSW_OK) return rc; }  // the scan buffers of this context are about to be reused
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
  HIP_TRY(c->d_ext_lists.reserve(4 * ((size_t)n_tasks + 1) + 16));  // the full kernel's device-side task list (nothing pending: safe to grow)
  ExtPrepass* d_pre = (ExtPrepass*)c->d_pre.ptr;
  ExtPrepass* h_pre = (ExtPrepass*)c->h_pre.ptr;
  // one fill for the scan record (+0), the bin counts (+64) and the kernel's queue head (+128): every small fill is a kernel
  // of its own on the stream, and the one between the scan and the main launch used to delay the latter
  static_assert(sizeof(ExtPrepass) <= 64, "scan buffer layout");
  HIP_TRY(hipMemsetAsync(d_pre, 0, 128 + 16, s));
  launch_ext_prepass((const uint32_t*)d_wire, wire_bytes >> 2, n_tasks, d_pre, s);
  HIP_TRY(hipGetLastError());
  c->stats.ext_calls++; c->stats.ext_tasks += (uint64_t)n_tasks; c->stats.ext_wire_bytes += wire_bytes;
  {
    // Asynchronous: scan, main launch (sized for ASYNC_QCAP / ASYNC_RCAP, checking the scan on the device) and the scan's
    // read-back are enqueued back to back; nothing waits.  Errors surface at the next call on this context or at
    // bpsw_last_kernel_ms, which is also where a batch that outgrew the geometry is re-launched.
    // LDS geometry: the first call on a context is sized for ASYNC_QCAP / ASYNC_RCAP; later ones for the longest sides the
    // verified calls have shown, plus headroom (a third of the LDS for 2x150 bp reads, so that the launches of several
    // contexts and the rescue kernel are resident together).  Same caps as launch_ext_kernel's rounding.
    int qcap = ASYNC_QCAP, rcap = ASYNC_RCAP;
    if (c->ext_geom_q > 0) {
      qcap = std::min(ASYNC_QCAP, (c->ext_geom_q + 32 + 31) & ~31);
      rcap = std::min(ASYNC_RCAP, (c->ext_geom_r + 128 + 63) & ~63);
    }
    HIP_TRY(hipEventRecord(c->ev[4], s));
    static const bool split_on = !(getenv("BPSW_EXT_SPLIT") && atoi(getenv("BPSW_EXT_SPLIT")) == 0);
    int* d_queue = (int*)((char*)c->d_pre.ptr + 128);
    if (split_on) {
      // the short kernel over every task, deferring on the device what it cannot take (nobody has seen the records
      // here), and the full kernel behind it for that list; both check the scan and leave a bad batch untouched
      int* d_list = (int*)c->d_ext_lists.ptr;
      HIP_TRY(hipMemsetAsync(d_list, 0, sizeof(int), s));
      // the sift kernel in front, as on the host-buffer path (extend_batch_impl): large batches of a context whose verified calls
      // have shown reads of the 2x150 bp kind (nobody has seen this batch's records: a batch of longer flanks pays a launch that
      // examines nothing)
      static const bool sift_on = !(getenv("BPSW_EXT_SIFT") && atoi(getenv("BPSW_EXT_SIFT")) == 0);
      static const int sift_min = getenv("BPSW_EXT_SIFT_MIN") ? atoi(getenv("BPSW_EXT_SIFT_MIN")) : 8192;
      const int sift_dm = sift_uniform_dm(c->ext_mat, c->ext_sc.exact_a);
      const bool use_sift = sift_on && n_tasks >= sift_min && (c->shortcut_mask & 32) && sift_dm > 0 && c->ext_geom_q > 0 && c->ext_geom_q <= 140;
      const size_t sift_rec_off = ((size_t)n_tasks + 15) & ~(size_t)15;
      uint8_t* d_sflag = nullptr;
      uint4* d_srecs = nullptr;
      if (use_sift) {
        HIP_TRY(c->d_sift.reserve(sift_rec_off + 32 * (size_t)n_tasks));
        d_sflag = (uint8_t*)c->d_sift.ptr;
        d_srecs = (uint4*)((char*)c->d_sift.ptr + sift_rec_off);
        HIP_TRY(launch_ext_sift_kernel((const uint32_t*)d_wire, n_tasks, (int16_t*)d_out, c->ext_sc, sift_dm, 127, d_sflag, d_srecs, s, KernelEvents(), d_pre));
      }
      HIP_TRY(launch_ext_kernel((const uint32_t*)d_wire, n_tasks, (int16_t*)d_out, c->ext_sc, qcap, rcap, c->num_cu, d_queue, nullptr, s,
                                d_pre, true, KernelEvents(), true, d_list, -255, d_sflag, d_srecs));
      // (sized like a normal launch: how many tasks the first one defers is unknown here -- all of them when the gap costs rule the
      // register sweeps out --, an empty list returns at once and the persistent queue lets surplus waves leave immediately)
      HIP_TRY(launch_ext_kernel((const uint32_t*)d_wire, n_tasks, (int16_t*)d_out, c->ext_sc, qcap, rcap, c->num_cu,
                                d_queue, nullptr, s, d_pre, true, KernelEvents(), false, d_list));
    } else {
      HIP_TRY(launch_ext_kernel((const uint32_t*)d_wire, n_tasks, (int16_t*)d_out, c->ext_sc, qcap, rcap, c->num_cu, d_queue, nullptr, s,
                                d_pre, true));
    }
    c->pend_ext.qcap = qcap; c->pend_ext.rcap = rcap;
    HIP_TRY(hipEventRecord(c->ev[5], s));
    HIP_TRY(hipMemcpyAsync(h_pre, d_pre, 128, hipMemcpyDeviceToHost, s));
    c->have_ext_ev = true;
    c->pend_ext.active = true; c->pend_ext.d_wire = d_wire; c->pend_ext.wire_bytes = wire_bytes; c->pend_ext.n_tasks = n_tasks;
    c->pend_ext.d_out = d_out; c->pend_ext.s = s;
    return BPSW_OK;
  }
}

int bpsw_get_stats(bpsw_ctx_t* c, bpsw_stats_t* out) {
  if (!c || !out) return fail(BPSW_ERR_ARG, "null argument");
  std::lock_guard<std::mutex> g(c->mu);
  *out = c->stats;
  return BPSW_OK;
}
int bpsw_reset_stats(bpsw_ctx_t* c) {
  if (!c) return fail(BPSW_ERR_ARG, "null context");
  std::lock_guard<std::mutex> g(c->mu);
  memset(&c->stats, 0, sizeof c->stats);
  return BPSW_OK;
}

}  // extern "C"
