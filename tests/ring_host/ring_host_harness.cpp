// ring_host_harness.cpp -- the HOST half of the submission ring (cloud-scale-bwamem_amd/csrc/bpsw_ring.cpp, compiled unchanged) under
// -fsanitize=thread / address,undefined on a box without a GPU.  Test infrastructure: nothing here is linked into the product.
//
// What plays the device.  The few HIP entry points bpsw_ring.cpp uses are defined here over ordinary memory, and an epoch's
// "resident kernel" is a group of C++ threads -- one poller, a few workers -- that follow bpsw_ring_dev.h step by step
// (ring_poller, ring_next_unit, ring_unit_done; the line references below are to that file).  Where the device code leans on the
// hardware for an ordering, the emulation states it as the C++ ordering it amounts to -- that mapping IS the visibility argument of
// DESIGN.md 4.2a, written as code a thread sanitizer checks:
//     results: system-scope stores + s_waitcnt vmcnt(0) before the `done` add     ->  the add is a release (fetch_add acq_rel)
//     completion word stored behind its own s_waitcnt vmcnt(0)                    ->  a release store
//     h_tail acquire load / descriptor words                                      ->  acquire load
//     the closing handshake's sequentially consistent stores and loads            ->  seq_cst, as on the host side
// A "unit" computes a checksum the caller can recompute, into the caller's result block, which the caller poisoned before it
// published the descriptor (ring_poison / ring_check: the product's own tripwire runs here too).
//
// What the run checks: every call of every thread returns, with its own results; nothing is lost across idle closes, used-up rings
// (capacity 64), pauses (bpsw_ref_load's ring_pause / ring_resume) and both ring classes at once; and the sanitizer saw no race,
// no lock inversion, no out-of-bounds access in the host code.
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "bpsw_internal.h"

using namespace bpsw;

// ------------------------------------------------------------------------------------------------ what bpsw_runtime.cpp provides
namespace bpsw {
static thread_local std::string t_err;
int fail(int code, const std::string& msg) { t_err = msg; return code; }
double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
double ext_call_age_ms(int) { return 1e9; }
double wait_est_update(double est, double took_ms, int, bool) { return est <= 0. ? took_ms : 0.75 * est + 0.25 * took_ms; }
bool wait_naps(double) { return false; }
void wait_nap(double) {}
void wait_poll_pause(int polls, double, double) {
  if (polls < 64) sched_yield();
  else { timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
}
bool spin_wait() { return false; }
}  // namespace bpsw
extern "C" const char* bpsw_last_error(void) { return bpsw::t_err.c_str(); }

#include "fake_ring_device.h"
using fake_ring::play_poller;
using fake_ring::play_worker;
using fake_ring::dev_clock;

// payload of the harness's descriptors (words 8..): where the unit checksums go
struct TestPayload {
  uint64_t out;   // uint32_t per unit
  uint32_t salt, pad;
};
static uint32_t unit_value(uint32_t salt, uint32_t done_value, uint32_t unit) { return (salt * 2654435761u) ^ (done_value * 40503u) ^ (unit * 0x9e3779b9u) ^ 0x1u; }
static void test_unit(const uint32_t* word, uint32_t unit) {
  RingDescHead head;
  TestPayload pl;
  memcpy(&head, word, sizeof head);
  memcpy(&pl, word + sizeof(RingDescHead) / 4, sizeof pl);
  __atomic_store_n(((uint32_t*)(uintptr_t)pl.out) + unit, unit_value(pl.salt, head.done_value, unit), __ATOMIC_RELAXED);
}

// ------------------------------------------------------------------------------------------------ the HIP entry points bpsw_ring.cpp uses
struct FakeEvent {
  std::atomic<double> t_ms{0.};
  std::atomic<bool> ready{true};
};
struct FakeKernel {
  std::thread main;
  std::atomic<bool> finished{false};
  std::vector<FakeEvent*> end_events;  // recorded behind the kernel: stamped when it ends (touched under the stream's lock and by main after finished)
  std::mutex mu;
};
struct ihipStream_t {
  std::mutex mu;
  FakeKernel* k = nullptr;
};
struct ihipEvent_t : FakeEvent {};
static int g_workers = 6;
static std::atomic<int> g_launches{0};

static void stream_drain(ihipStream_t* s) {  // (caller holds s->mu)
  if (!s->k) return;
  if (s->k->main.joinable()) s->k->main.join();
  delete s->k;
  s->k = nullptr;
}
static hipError_t fake_launch(const RingArgs& A, hipStream_t s) {
  std::lock_guard<std::mutex> lk(s->mu);
  stream_drain(s);  // stream order: behind the previous epoch's kernel
  FakeKernel* k = new FakeKernel;
  s->k = k;
  ++g_launches;
  k->main = std::thread([A, k]() {
    std::vector<std::thread> ts;
    ts.emplace_back(play_poller, A);
    for (int i = 0; i < g_workers; ++i) ts.emplace_back([A, i]() { play_worker(A, i, test_unit); });
    for (auto& t : ts) t.join();
    std::lock_guard<std::mutex> lk(k->mu);
    const double now = wall_ms();
    for (FakeEvent* e : k->end_events) { e->t_ms.store(now); e->ready.store(true, std::memory_order_release); }
    k->finished.store(true, std::memory_order_release);
  });
  return hipSuccess;
}
namespace bpsw {
hipError_t launch_swp_resident(int, const RingArgs& A, int, hipStream_t s) { return fake_launch(A, s); }
hipError_t launch_ext_resident(const RingArgs& A, int, hipStream_t s) { return fake_launch(A, s); }
}  // namespace bpsw

extern "C" {
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 100000; return hipSuccess; }  // kHz: 100 ticks per microsecond
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new ihipEvent_t; return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { return e->ready.load(std::memory_order_acquire) ? hipSuccess : hipErrorNotReady; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  if (!a->ready.load(std::memory_order_acquire) || !b->ready.load(std::memory_order_acquire)) return hipErrorNotReady;
  *ms = (float)(b->t_ms.load() - a->t_ms.load());
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->k) {
    std::lock_guard<std::mutex> lk2(s->k->mu);
    if (!s->k->finished.load(std::memory_order_acquire)) {
      e->ready.store(false, std::memory_order_relaxed);
      s->k->end_events.push_back(e);
      return hipSuccess;
    }
  }
  e->t_ms.store(wall_ms());
  e->ready.store(true, std::memory_order_release);
  return hipSuccess;
}
const char* hipGetErrorString(hipError_t) { return "fake"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t s) {
  std::lock_guard<std::mutex> lk(s->mu);
  stream_drain(s);  // stream order
  memset(p, v, n);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = new ihipStream_t; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned int, int) { *s = new ihipStream_t; return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) {
  std::lock_guard<std::mutex> lk(s->mu);
  return (!s->k || s->k->finished.load(std::memory_order_acquire)) ? hipSuccess : hipErrorNotReady;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  std::lock_guard<std::mutex> lk(s->mu);
  stream_drain(s);
  return hipSuccess;
}
}

// ------------------------------------------------------------------------------------------------ the callers
struct Caller {
  RingDone* done;
  uint32_t* out;
  uint32_t seq = 0;
  double est = 0.;
};
static std::atomic<uint64_t> g_calls{0}, g_units{0}, g_bad{0}, g_fallback{0}, g_relaunched{0};
// what a product caller does when the ring cannot take (or will never run) its batch: a launch of its own -- here: the units computed in place
static void own_launch(Caller& c, uint32_t n_units, uint32_t salt) {
  for (uint32_t u = 0; u < n_units; ++u) c.out[u] = unit_value(salt, c.seq, u);
}

static int one_call(Caller& c, int c_class, uint32_t n_units, uint32_t salt) {
  RingDesc desc;
  memset(&desc, 0, sizeof desc);
  RingDescHead head;
  memset(&head, 0, sizeof head);
  if (++c.seq == 0) ++c.seq;
  head.n_units = n_units; head.done_value = c.seq; head.done_ptr = (uint64_t)(uintptr_t)c.done;
  TestPayload pl;
  pl.out = (uint64_t)(uintptr_t)c.out; pl.salt = salt; pl.pad = 0;
  memcpy(desc.w, &head, sizeof head);
  memcpy(desc.w + sizeof(RingDescHead) / 4, &pl, sizeof pl);
  if (!ring_usable(0, c_class)) {  // a ring that failed earlier: as sw_stage_run / extend_batch_impl, a launch per batch from here on
    own_launch(c, n_units, salt);
    g_fallback.fetch_add(1);
  } else {
    ring_poison(c.out, 1, n_units, 0);
    int rc = ring_submit(0, c_class, 256, desc);
    if (rc != BPSW_OK && !ring_usable(0, c_class)) {  // this call met the failed epoch launch
      own_launch(c, n_units, salt);
      g_fallback.fetch_add(1);
    } else {
      if (rc != BPSW_OK) return rc;
      rc = ring_wait(0, c_class, c.done, c.seq, &c.est);
      if (rc == BPSW_RING_RELAUNCH) {  // another thread's epoch launch failed while this batch waited to be carried over
        own_launch(c, n_units, salt);
        g_relaunched.fetch_add(1);
      } else {
        if (rc != BPSW_OK) return rc;
        rc = ring_check(c.out, 1, n_units, 0, "harness");
        if (rc != BPSW_OK) return rc;
      }
    }
  }
  for (uint32_t u = 0; u < n_units; ++u)
    if (c.out[u] != unit_value(salt, c.seq, u)) g_bad.fetch_add(1);
  g_calls.fetch_add(1); g_units.fetch_add(n_units);
  return BPSW_OK;
}

int main(int argc, char** argv) {
  const int n_threads = argc > 1 ? atoi(argv[1]) : 8, n_calls = argc > 2 ? atoi(argv[2]) : 400;
  g_workers = argc > 3 ? atoi(argv[3]) : 6;
  // a tiny ring that idles out quickly: the closing handshake, the roll-over and the carry-over all happen hundreds of times
  setenv("BPSW_RING_CAPACITY", "64", 0);
  setenv("BPSW_RING_IDLE_US", "300", 0);
  setenv("BPSW_RING_TIMEOUT_MS", "60000", 0);
  std::atomic<bool> stop{false};
  std::atomic<int> failures{0};
  std::vector<std::thread> ts;
  for (int t = 0; t < n_threads; ++t) {
    ts.emplace_back([&, t]() {
      Caller c;
      c.done = (RingDone*)aligned_alloc(64, 64);
      memset((void*)c.done, 0, 64);
      c.out = (uint32_t*)aligned_alloc(64, 4 * 64);
      std::minstd_rand rng(99u + (unsigned)t);
      const int c_class = (t % 3 == 0) ? RING_CLASS_EXT : (t % 3 == 1) ? 3 : 5;
      for (int i = 0; i < n_calls; ++i) {
        const uint32_t n_units = 1u + rng() % 48u;
        const int rc = one_call(c, c_class, n_units, (uint32_t)(t * 100003 + i));
        if (rc != BPSW_OK) { fprintf(stderr, "thread %d call %d: rc %d (%s)\n", t, i, rc, bpsw_last_error()); failures.fetch_add(1); break; }
        const unsigned r = rng() % 16u;
        if (r == 0u) { timespec s = {0, 600000}; nanosleep(&s, nullptr); }  // longer than the idle limit: the epoch closes under this caller
        else if (r < 6u) { timespec s = {0, (long)(200000 + rng() % 200000u)}; nanosleep(&s, nullptr); }  // about the idle limit: publications race the closing handshake
        else if (r < 9u) sched_yield();
      }
      free((void*)c.done); free(c.out);
    });
  }
  std::thread pauser([&]() {  // bpsw_ref_load / bpsw_destroy's buffer releases: close every epoch, hold the rings, let go
    int n = 0;
    while (!stop.load()) {
      timespec s = {0, 3000000};
      nanosleep(&s, nullptr);
      ring_pause(0);
      if ((++n & 3) == 0) ring_pause(0);  // nested, as a reference reload that frees the old reference does
      sched_yield();
      if ((n & 3) == 0) ring_resume(0);
      ring_resume(0);
    }
  });
  for (auto& t : ts) t.join();
  stop.store(true);
  pauser.join();
  uint64_t e = 0, s = 0, carried = 0, checked = 0, faults = 0;
  ring_get_stats(0, &e, &s, &carried);
  ring_integrity_stats(&checked, &faults);
  printf("RINGHOST threads %d calls %llu units %llu epochs %llu submitted %llu carried %llu launches %d wrong %llu integrity_checked %llu integrity_faults %llu failures %d fallback %llu relaunched %llu\n",
         n_threads, (unsigned long long)g_calls.load(), (unsigned long long)g_units.load(), (unsigned long long)e, (unsigned long long)s,
         (unsigned long long)carried, g_launches.load(), (unsigned long long)g_bad.load(), (unsigned long long)checked, (unsigned long long)faults, failures.load(),
         (unsigned long long)g_fallback.load(), (unsigned long long)g_relaunched.load());
  const bool fail_test = getenv("BPSW_RING_TEST_FAIL_LAUNCH") != nullptr || getenv("BPSW_RING_TEST_FAIL_CARRY_LAUNCH") != nullptr;  // (then some calls never went through the ring: `submitted` is smaller)
  const bool ok = failures.load() == 0 && g_bad.load() == 0 && faults == 0 && g_calls.load() == (uint64_t)n_threads * (uint64_t)n_calls &&
                  (fail_test ? s <= g_calls.load() : s == g_calls.load());
  // close what is open so that no thread outlives main
  ring_pause(0);
  ring_resume(0);
  return ok ? 0 : 1;
}
