// bpsw_ring.cpp -- host side of the per-device submission ring (bpsw_ring.h): services, epochs, submission, waiting.
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>

#include "bpsw_internal.h"

namespace bpsw {

namespace {

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

struct RingService {
  std::recursive_mutex mu;  // (recursive: bpsw_ref_load pauses the rings and may free the old reference, which pauses them again)
  int pause_depth = 0;
  int device = 0, c_class = 0;
  bool inited = false, running = false, broken = false;
  bool launch_failed = false;  // broken because an epoch could not be launched: the descriptors at h_desc[0..published) will never be consumed
  uint32_t epoch = 0, published = 0, capacity = 0;
  uint32_t carry_from = 0, carry_n = 0;  // descriptors of a closed epoch the device did not consume
  RingHostCtl* H = nullptr;
  RingDesc* h_desc = nullptr;
  void* h_block = nullptr;
  void* d_block = nullptr;
  RingDevCtl* D = nullptr;
  RingDesc* d_desc = nullptr;
  RingCtr* ctr = nullptr;
  size_t d_zero_bytes = 0;  // control block + counters: cleared at every epoch start
  hipStream_t stream = nullptr;
  int blocks = 0, num_cu = 256;
  unsigned long long ticks_per_us = 100;
  // statistics (under mu)
  uint64_t epochs = 0, submitted = 0, carried = 0;
  // every epoch's launch is bracketed by two events on the ring's stream: the duration of the resident kernel as a kernel trace sees it
  hipEvent_t ev_begin[2] = {nullptr, nullptr}, ev_end[2] = {nullptr, nullptr};
  bool ev_pending[2] = {false, false};
  double epochs_ms = 0.;
  uint64_t epochs_timed = 0;
};

// one service per device and ring class
constexpr int kRingClasses = 3;  // 3: SW jobs with mates up to 171 bases, 5: up to 256, RING_CLASS_EXT: small extension batches
RingService g_rings[64][kRingClasses];
RingService& service(int device, int c_class) { return g_rings[device >= 0 && device < 64 ? device : 0][c_class == 3 ? 0 : c_class == RING_CLASS_EXT ? 2 : 1]; }

int hip_fail_ring(hipError_t e, const char* what) { return fail(BPSW_ERR_DEVICE, std::string("ring: ") + what + ": " + hipGetErrorString(e)); }
#define RING_TRY(expr)                                      \
  do {                                                      \
    hipError_t e_ = (expr);                                 \
    if (e_ != hipSuccess) return hip_fail_ring(e_, #expr);  \
  } while (0)

// (caller holds S.mu and has set the device)
int init_service(RingService& S, int device, int c_class, int num_cu) {
  S.device = device; S.c_class = c_class;
  int cap = env_int("BPSW_RING_CAPACITY", 16384);
  cap = cap < 64 ? 64 : (cap > (1 << 20) ? (1 << 20) : cap);
  S.capacity = (uint32_t)cap;
  const size_t h_bytes = sizeof(RingHostCtl) + sizeof(RingDesc) * (size_t)cap;
  RING_TRY(hipHostMalloc(&S.h_block, h_bytes, hipHostMallocDefault));
  memset(S.h_block, 0, h_bytes);
  S.H = (RingHostCtl*)S.h_block;
  S.h_desc = (RingDesc*)((char*)S.h_block + sizeof(RingHostCtl));
  S.d_zero_bytes = sizeof(RingDevCtl) + sizeof(RingCtr) * (size_t)cap;
  RING_TRY(hipMalloc(&S.d_block, S.d_zero_bytes + sizeof(RingDesc) * (size_t)cap));
  S.D = (RingDevCtl*)S.d_block;
  S.ctr = (RingCtr*)((char*)S.d_block + sizeof(RingDevCtl));
  S.d_desc = (RingDesc*)((char*)S.d_block + S.d_zero_bytes);
  // The epoch's kernel sits on its hardware queue for as long as it lives: a stream that shared the queue would wait behind it for
  // the whole epoch (the runtime maps a process's streams onto GPU_MAX_HW_QUEUES queues PER PRIORITY LEVEL, round robin).  A stream of
  // the highest priority takes its queue from a pool no other stream of this library uses.
  {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; (void)hipGetLastError(); }
    if (greatest != least) RING_TRY(hipStreamCreateWithPriority(&S.stream, hipStreamNonBlocking, greatest));
    else RING_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
  }
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) S.ticks_per_us = (unsigned long long)(khz / 1000 > 0 ? khz / 1000 : 1);
  S.num_cu = num_cu;
  S.inited = true;
  return BPSW_OK;
}

// (caller holds S.mu and has set the device)  Starts the next epoch, carrying over what the closed one left unconsumed.
int start_epoch(RingService& S) {
  if (S.broken) return fail(BPSW_ERR_DEVICE, "ring: the submission ring of this device failed earlier");
  ++S.epoch;
  if (S.epoch >= 0xffffffu) S.epoch = 1;
  if (S.carry_n && S.carry_from) memmove(S.h_desc, S.h_desc + S.carry_from, sizeof(RingDesc) * (size_t)S.carry_n);
  const uint32_t carrying = S.carry_n;
  S.carried += S.carry_n;
  S.published = S.carry_n;
  S.carry_from = S.carry_n = 0;
  // (nobody else looks at the block yet: the epoch's kernel is launched below, and its first loads are ordered behind the launch)
  S.H->close_req.store(0, std::memory_order_relaxed);
  S.H->state.store(ring_state(S.epoch, 0, RING_OPEN), std::memory_order_relaxed);
  S.H->tail.store(S.published, std::memory_order_seq_cst);
  RingArgs A;
  A.H = S.H; A.h_desc = S.h_desc; A.D = S.D; A.d_desc = S.d_desc; A.ctr = S.ctr;
  A.epoch = S.epoch; A.capacity = S.capacity;
  A.sleep_ticks_us = S.ticks_per_us;
  A.idle_ticks = (unsigned long long)env_int("BPSW_RING_IDLE_US", 2000) * S.ticks_per_us;
  A.worker_idle_ticks = (unsigned long long)env_int("BPSW_RING_WORKER_IDLE_US", 50000) * S.ticks_per_us;
  // how long a waiting worker sleeps at most between two looks: 16 rounds, ~55 us, for every class.  (A quarter of that was tried for
  // the extension ring, whose tasks are 20-40 us each: a lone call is no faster -- 0.083-0.091 ms either way --, and sixteen callers of
  // 63 / 126-task calls make 85 k / 61 k calls/s instead of 103 k / 71 k: a thousand waiting waves that look four times as often are
  // in the working waves' way.)
  A.nap_rounds_max = (uint32_t)env_int("BPSW_RING_NAP_ROUNDS", 16);
  if (A.nap_rounds_max < 1) A.nap_rounds_max = 1;
  A.pad = 0;
  {
    // Worker workgroups per CU (four wavefronts each).  The resident grid holds its wave slots for as long as the epoch lives, and more
    // workers are not more throughput: the device is shared with the extension kernels, and what a worker more takes from them costs the
    // step more than a batch's units waiting a little for a free worker.  Measured on the bench (configs[2], 32 threads): 0.5 / 0.75 / 1 /
    // 1.25 / 1.5 / 2 / 3 workgroups per CU = 1.63 / 2.34 / 2.35 / 2.26 / 2.20 / 2.05 / 1.70 x 10^8 reads/s; configs[4] (mates of 250
    // bases, the second class; the step is the extension's): 0.5 / 1 / 2 = 2.28 / 2.12 / 1.96 x 10^7.
    // ... unless the rescue path has the device to itself (no extension call on it for 20 ms: an executor that runs boundary 1 only, or
    // a phase of a job): then the wave slots are nobody else's.  Rescue calls alone, 0.5 / 1 / 2 / 3 / 4 workgroups per CU: mates of 250
    // bases 0.42 / 0.83 / 1.10 / 1.03 / 0.93 x 10^8 reads/s (a launch per batch: 1.18); mates of 150 bases 1 / 2 / 3 / 4 / 5: 3.44 / 2.93 /
    // 2.98 / 2.24 / 1.96 (3.67; host-bound) -- so only the second class grows, to 2.  The choice holds for the epoch (it ends with its
    // 16 384 batches or 2 ms after the last one).  BPSW_RING_WG_PER_CU fixes it.
    const bool alone = ext_call_age_ms(S.device) > 20.0;
    // (the extension ring serves small batches only: one workgroup per CU, idle unless such calls are being made)
    const double dflt = S.c_class == RING_CLASS_EXT ? 1.0 : S.c_class == 3 ? 1.0 : (alone ? 2.0 : 0.5);
    const double per_cu = getenv("BPSW_RING_WG_PER_CU") ? atof(getenv("BPSW_RING_WG_PER_CU")) : dflt;
    const int blocks = (int)(S.num_cu * (per_cu > 0.0 ? per_cu : 1.0));
    S.blocks = blocks < 2 ? 2 : blocks;
  }
  hipError_t e = hipMemsetAsync(S.d_block, 0, S.d_zero_bytes, S.stream);  // behind the previous epoch's kernel, in stream order
  const int slot = (int)(S.epoch & 1u);
  if (e == hipSuccess && !S.ev_begin[slot]) {
    e = hipEventCreate(&S.ev_begin[slot]);
    if (e == hipSuccess) e = hipEventCreate(&S.ev_end[slot]);
  }
  if (e == hipSuccess && S.ev_pending[slot]) {  // the epoch before last: long over (this stream has run a whole epoch since)
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, S.ev_begin[slot], S.ev_end[slot]) == hipSuccess) { S.epochs_ms += ms; ++S.epochs_timed; }
    else (void)hipGetLastError();
    S.ev_pending[slot] = false;
  }
  if (e == hipSuccess) e = hipEventRecord(S.ev_begin[slot], S.stream);
  // (test hook, BPSW_RING_TEST_FAIL_LAUNCH=k: the k-th epoch launch of a ring "fails" -- tests/test_ring_gpu.py checks that its callers
  // go on with a launch per batch)
  static const int fail_at = env_int("BPSW_RING_TEST_FAIL_LAUNCH", 0);
  if (e == hipSuccess && fail_at > 0 && S.epochs + 1 == (uint64_t)fail_at) e = hipErrorLaunchFailure;
  // (... and BPSW_RING_TEST_FAIL_CARRY_LAUNCH=1: the first epoch launch of a ring that CARRIES descriptors over "fails": their waiters --
  // other threads than the one in here -- must end up with launches of their own too: tests/ring_host)
  static const int fail_carry = env_int("BPSW_RING_TEST_FAIL_CARRY_LAUNCH", 0);
  if (e == hipSuccess && fail_carry > 0 && carrying > 0) e = hipErrorLaunchFailure;
  if (e == hipSuccess) e = S.c_class == RING_CLASS_EXT ? launch_ext_resident(A, S.blocks, S.stream) : launch_swp_resident(S.c_class, A, S.blocks, S.stream);
  if (e == hipSuccess) { e = hipEventRecord(S.ev_end[slot], S.stream); S.ev_pending[slot] = e == hipSuccess; }
  if (e != hipSuccess) { S.broken = true; S.launch_failed = true; return hip_fail_ring(e, "epoch launch"); }
  S.running = true;
  ++S.epochs;
  return BPSW_OK;
}

// the device's state word of the current epoch with CLOSING resolved (the device commits or withdraws within microseconds)
uint64_t resolved_state(RingService& S) {
  const double t0 = wall_ms();
  for (;;) {
    const uint64_t s = S.H->state.load(std::memory_order_seq_cst);
    if (ring_state_epoch(s) != S.epoch || ring_state_phase(s) != RING_CLOSING) return s;
    if (wall_ms() - t0 > 200.0) return s;  // a device that never resolves: the caller's watchdog reports it
    sched_yield();
  }
}

// (caller holds S.mu)  Notes a closed epoch: what it consumed, what has to be carried over.
bool ring_debug() {
  static const bool on = env_int("BPSW_RING_DEBUG", 0) != 0;
  return on;
}
void note_closed(RingService& S, uint64_t s) {
  const uint32_t consumed = ring_state_consumed(s);
  if (ring_debug())
  {
    const uint64_t du = S.H->diag_units.load(std::memory_order_relaxed);
    const double nu = du ? (double)du : 1.0;
    static const char* why[5] = {"?", "asked by the host", "ring used up", "idle", "no progress"};
    fprintf(stderr, "bPSW ring[%d/%d]: epoch %u closed (%s): consumed %u of %u published, %u worker waves took a unit, %d workgroups; diag: %llu units, "
            "taken %.1f us after publication on average, %.1f us per unit\n", S.device, S.c_class, S.epoch, why[S.H->close_reason.load(std::memory_order_relaxed) < 5 ? S.H->close_reason.load(std::memory_order_relaxed) : 0], consumed, S.published, (unsigned)S.H->workers_seen.load(std::memory_order_relaxed),
            S.blocks, (unsigned long long)du, (double)S.H->diag_claim_ticks.load(std::memory_order_relaxed) / nu / (double)S.ticks_per_us, (double)S.H->diag_unit_ticks.load(std::memory_order_relaxed) / nu / (double)S.ticks_per_us);
  }
  S.running = false;
  S.carry_from = consumed;
  S.carry_n = S.published > consumed ? S.published - consumed : 0;
}

int wait_closed(RingService& S, double limit_ms) {
  const double t0 = wall_ms();
  for (;;) {
    const uint64_t s = S.H->state.load(std::memory_order_acquire);
    if (ring_state_epoch(s) == S.epoch && ring_state_phase(s) == RING_CLOSED) { note_closed(S, s); return BPSW_OK; }
    if (wall_ms() - t0 > limit_ms) { S.broken = true; return fail(BPSW_ERR_DEVICE, "ring: the resident kernel did not close its epoch"); }
    timespec ts = {0, 20000};
    nanosleep(&ts, nullptr);
  }
}

int timeout_ms() {
  static const int v = env_int("BPSW_RING_TIMEOUT_MS", 20000);
  return v;
}

}  // namespace

bool ring_enabled() {
  static const bool on = env_int("BPSW_RING", 1) != 0;
  return on;
}

// false once the ring of (device, class) has failed (an epoch that could not be launched, a watchdog): its callers go back to a launch per
// batch (sw_stage_run) instead of failing for the rest of the process's life
bool ring_usable(int device, int c_class) {
  RingService& S = service(device, c_class);
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  return !S.broken;
}

// Appends one descriptor to the ring of (device, class); starts an epoch when none is open.  The caller then waits on its completion
// record (ring_wait).  Caller has set the device.
int ring_submit(int device, int c_class, int num_cu, const RingDesc& desc) {
  RingService& S = service(device, c_class);
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  if (!S.inited) { const int rc = init_service(S, device, c_class, num_cu); if (rc != BPSW_OK) return rc; }
  for (;;) {
    if (S.running) {  // an epoch that closed by itself (idle) since the last submission
      const uint64_t s = S.H->state.load(std::memory_order_acquire);
      if (ring_state_epoch(s) == S.epoch && ring_state_phase(s) == RING_CLOSED) note_closed(S, s);
    }
    if (!S.running) { const int rc = start_epoch(S); if (rc != BPSW_OK) return rc; }
    if (S.published < S.capacity) break;
    // the epoch's ring is used up: the poller closes it once it has consumed the last descriptor
    const int rc = wait_closed(S, (double)timeout_ms());
    if (rc != BPSW_OK) return rc;
  }
  S.h_desc[S.published] = desc;
  // the descriptor before the count (release), and W(tail) ; R(state) against the poller's W(state) ; R(tail) -- both sequentially
  // consistent (on x86 the store is an xchg: the full fence round 5 had as a free-standing one), so that at least one side sees the other
  S.H->tail.store(++S.published, std::memory_order_seq_cst);
  ++S.submitted;
  const uint64_t s = resolved_state(S);
  if (ring_state_epoch(s) == S.epoch && ring_state_phase(s) == RING_CLOSED) {
    note_closed(S, s);
    if (S.carry_n) return start_epoch(S);  // this descriptor (at least) was not consumed: it opens the next epoch
  }
  return BPSW_OK;
}

// A waiting caller's slow path: if the epoch closed with descriptors unconsumed and nobody has restarted it, do so; report a device
// error on the ring's stream.  Caller has set the device.
int ring_poke(int device, int c_class) {
  RingService& S = service(device, c_class);
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  if (!S.inited) return BPSW_OK;
  if (S.broken) return fail(BPSW_ERR_DEVICE, "ring: the submission ring of this device failed");
  if (S.running) {
    const uint64_t s = resolved_state(S);
    if (ring_state_epoch(s) == S.epoch && ring_state_phase(s) == RING_CLOSED) note_closed(S, s);
  }
  if (!S.running && S.carry_n) return start_epoch(S);
  const hipError_t q = hipStreamQuery(S.stream);
  if (q != hipSuccess && q != hipErrorNotReady) { S.broken = true; return hip_fail_ring(q, "resident kernel"); }
  return BPSW_OK;
}

// (caller holds nothing)  Is the batch that completes `done` with `value` among the descriptors no kernel will ever consume?  True only for a
// ring whose epoch LAUNCH failed (no resident kernel exists: start_epoch had moved the closed epoch's unconsumed descriptors to the front
// of the ring before the launch that failed, and later submitters were turned away).  The waiters of those descriptors used to get
// BPSW_ERR_DEVICE from ring_poke while only the thread that met the failure fell back to a launch (advisor, round 5).
static bool ring_unconsumed(int device, int c_class, const RingDone* done, uint32_t value) {
  RingService& S = service(device, c_class);
  std::lock_guard<std::recursive_mutex> lk(S.mu);
  if (!S.inited || !S.launch_failed || S.running) return false;
  for (uint32_t d = 0; d < S.published && d < S.capacity; ++d) {
    RingDescHead h;
    memcpy(&h, S.h_desc[d].w, sizeof h);
    if (h.done_ptr == (uint64_t)(uintptr_t)done && h.done_value == value) return true;
  }
  return false;
}

// Waits until the completion record shows `value`.  est_ms: running average of this caller's waits of the kind (updated).
int ring_wait(int device, int c_class, const RingDone* done, uint32_t value, double* est_ms) {
  const double t0 = wall_ms();
  double est = est_ms ? *est_ms : 0.;
  const bool napped = !spin_wait() && wait_naps(est);
  if (!spin_wait()) wait_nap(est);
  int polls = 0;
  double next_poke = 2.0;
  while (done->value.load(std::memory_order_acquire) != value) {
    const double waited = wall_ms() - t0;
    if (spin_wait()) sched_yield();
    else wait_poll_pause(++polls, waited, est);
    if (waited > next_poke) {
      const int rc = ring_poke(device, c_class);
      if (rc != BPSW_OK) {
        if (done->value.load(std::memory_order_acquire) == value) break;  // (it had been consumed before the ring broke, and has finished)
        return ring_unconsumed(device, c_class, done, value) ? BPSW_RING_RELAUNCH : rc;
      }
      // pokes at 2, 4, 8 ... ms, never further apart than the watchdog's limit (round 5 doubled without bound and tested the limit only at
      // a poke: BPSW_RING_TIMEOUT_MS=20000 fired after 32.8 s)
      next_poke = waited * 2.0;
      if (next_poke < (double)timeout_ms() && next_poke * 2.0 > (double)timeout_ms()) next_poke = (double)timeout_ms();
      if (next_poke > waited + (double)timeout_ms()) next_poke = waited + (double)timeout_ms();
      if (waited >= (double)timeout_ms()) {
        RingService& S = service(device, c_class);
        std::lock_guard<std::recursive_mutex> lk(S.mu);
        S.broken = true;
        return fail(BPSW_ERR_DEVICE, "ring: watchdog: a submitted batch was not completed within BPSW_RING_TIMEOUT_MS");
      }
    }
  }
  if (est_ms) *est_ms = wait_est_update(est, wall_ms() - t0, polls, napped);
  return BPSW_OK;
}

// The integrity tripwire.  The results of a batch and its completion word travel as separate posted writes over PCIe from up to a
// thousand wavefronts on eight XCDs (bpsw_ring_dev.h: the argument, and what it assumes of the fabric).  If the completion word ever
// overtook a result the caller would read stale bytes of an earlier call -- a silent wrong alignment.  So the caller poisons every
// record of its result block before it publishes the descriptor (a value no kernel writes) and looks at every record again after the
// completion word: a record that still holds the poison is counted, reported once on stderr, given a moment to arrive, and the call
// fails with BPSW_ERR_DEVICE if it does not.  Host-side only: the kernels do not know.  BPSW_RING_INTEGRITY=0 switches it off.
static std::atomic<uint64_t> g_integrity_checked{0}, g_integrity_faults{0};
bool ring_integrity_on() {
  static const bool on = env_int("BPSW_RING_INTEGRITY", 1) != 0;
  return on;
}
void ring_poison(uint32_t* first_word, size_t stride_words, size_t n_records, size_t second_word_offset) {
  if (!ring_integrity_on()) return;
  for (size_t i = 0; i < n_records; ++i) {
    first_word[i * stride_words] = RING_POISON;
    if (second_word_offset) first_word[i * stride_words + second_word_offset] = RING_POISON;
  }
  // (the block is the caller's own pinned memory and the descriptor that names it is published behind these stores: ring_submit's
  // release store of the tail)
}
int ring_check(const uint32_t* first_word, size_t stride_words, size_t n_records, size_t second_word_offset, const char* what) {
  if (!ring_integrity_on()) return BPSW_OK;
  g_integrity_checked.fetch_add(n_records, std::memory_order_relaxed);
  const auto scan = [&]() {
    size_t bad = 0;
    for (size_t i = 0; i < n_records; ++i) {
      const volatile uint32_t* r = first_word + i * stride_words;
      if (r[0] == RING_POISON || (second_word_offset && r[second_word_offset] == RING_POISON)) ++bad;
    }
    return bad;
  };
  size_t bad = scan();
  if (!bad) return BPSW_OK;
  g_integrity_faults.fetch_add(bad, std::memory_order_relaxed);
  static std::atomic<bool> said{false};
  if (!said.exchange(true))
    fprintf(stderr, "bPSW: RING INTEGRITY: %zu of %zu %s records were not in host memory when their batch's completion word was (reported once; "
            "bpsw_ring_integrity counts them).  Set BPSW_RING=0 and report this.\n", bad, n_records, what);
  const double t0 = wall_ms();
  while (bad && wall_ms() - t0 < 5.0) { sched_yield(); bad = scan(); }
  if (bad) return fail(BPSW_ERR_DEVICE, std::string("ring: integrity: ") + what + " records missing behind the completion word");
  return BPSW_OK;   // late, but whole: the call's results are what the kernel wrote
}
void ring_integrity_stats(uint64_t* checked, uint64_t* faults) {
  if (checked) *checked = g_integrity_checked.load(std::memory_order_relaxed);
  if (faults) *faults = g_integrity_faults.load(std::memory_order_relaxed);
}

double ring_ticks_per_ms(int device, int c_class) { return 1000.0 * (double)service(device, c_class).ticks_per_us; }

// Closes the open epochs of a device and waits for their kernels to end; the rings stay locked until ring_resume, so that a
// device-wide synchronisation in between (bpsw_ref_load / unload) cannot be held up by an epoch other threads keep feeding.
void ring_pause(int device) {
  for (int k = 0; k < kRingClasses; ++k) {
    RingService& S = g_rings[device >= 0 && device < 64 ? device : 0][k];
    S.mu.lock();
    ++S.pause_depth;
    if (!S.inited || !S.running) continue;
    S.H->close_req.store(S.epoch, std::memory_order_seq_cst);
    (void)wait_closed(S, 2000.0);
    (void)hipStreamSynchronize(S.stream);
  }
}
void ring_resume(int device) {
  for (int k = 0; k < kRingClasses; ++k) {
    RingService& S = g_rings[device >= 0 && device < 64 ? device : 0][k];
    if (--S.pause_depth == 0 && S.inited && !S.running && S.carry_n && !S.broken) (void)start_epoch(S);
    S.mu.unlock();
  }
}

void ring_get_stats(int device, uint64_t* epochs, uint64_t* submitted, uint64_t* carried, double* epochs_ms, uint64_t* epochs_timed) {
  uint64_t e = 0, s = 0, c = 0, t = 0;
  double ms_sum = 0.;
  for (int k = 0; k < kRingClasses; ++k) {
    RingService& S = g_rings[device >= 0 && device < 64 ? device : 0][k];
    std::lock_guard<std::recursive_mutex> lk(S.mu);
    // the durations of the epochs that are over (an epoch that is still open, or has not left the device yet, is not counted)
    for (int slot = 0; slot < 2; ++slot) {
      if (!S.ev_pending[slot] || hipEventQuery(S.ev_end[slot]) != hipSuccess) { (void)hipGetLastError(); continue; }
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, S.ev_begin[slot], S.ev_end[slot]) == hipSuccess) { S.epochs_ms += ms; ++S.epochs_timed; }
      else (void)hipGetLastError();
      S.ev_pending[slot] = false;
    }
    e += S.epochs; s += S.submitted; c += S.carried; ms_sum += S.epochs_ms; t += S.epochs_timed;
  }
  if (epochs) *epochs = e;
  if (submitted) *submitted = s;
  if (carried) *carried = c;
  if (epochs_ms) *epochs_ms = ms_sum;
  if (epochs_timed) *epochs_timed = t;
}

// Process exit / library unload: ask every open epoch to close and give it a moment -- plain memory traffic only, the HIP runtime
// may already be shutting down.  (Left alone an epoch closes by itself BPSW_RING_IDLE_US after its last descriptor.)
__attribute__((destructor)) static void ring_shutdown() {
  bool any = false;
  for (auto& dev : g_rings)
    for (RingService& S : dev)
      if (S.inited && S.running && S.H) { S.H->close_req.store(S.epoch, std::memory_order_seq_cst); any = true; }
  if (!any) return;
  const double t0 = wall_ms();
  for (auto& dev : g_rings)
    for (RingService& S : dev) {
      if (!(S.inited && S.running && S.H)) continue;
      while (wall_ms() - t0 < 50.0) {
        const uint64_t s = S.H->state.load(std::memory_order_acquire);
        if (ring_state_epoch(s) == S.epoch && ring_state_phase(s) == RING_CLOSED) break;
        sched_yield();
      }
    }
}

}  // namespace bpsw
