// fake_ring_device.h -- TEST INFRASTRUCTURE: the device side of the submission ring played by C++ threads, shared by the two CPU
// sanitizer harnesses (tests/ring_host: bpsw_ring.cpp alone; tests/host_san: the whole host layer).  A poller and workers that follow
// cloud-scale-bwamem_amd/csrc/bpsw_ring_dev.h step by step (ring_poller, ring_next_unit, ring_unit_done).  Where the device code leans on the
// hardware for an ordering, the emulation states it as the C++ ordering it amounts to -- that mapping IS the visibility argument of
// DESIGN.md 4.2a, written as code a thread sanitizer checks:
//     results: system-scope stores + s_waitcnt vmcnt(0) before the `done` add     ->  the add is a release (fetch_add acq_rel)
//     completion word stored behind its own s_waitcnt vmcnt(0)                    ->  a release store
//     h_tail acquire load / descriptor words                                      ->  acquire load
//     the closing handshake's sequentially consistent stores and loads            ->  seq_cst, as on the host side
#pragma once
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <chrono>
#include <random>

#include "bpsw_ring.h"

namespace fake_ring {
using namespace bpsw;

static unsigned long long dev_clock() {  // 100 MHz, as hipDeviceAttributeWallClockRate reports below
  return (unsigned long long)(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count() * 100.0);
}
template <class T> static T ld(const T* p, int mo = __ATOMIC_RELAXED) { return __atomic_load_n(p, mo); }
template <class T> static void st(T* p, T v, int mo = __ATOMIC_RELAXED) { __atomic_store_n(p, v, mo); }

static void play_poller(const RingArgs A) {  // bpsw_ring_dev.h: ring_poller
  uint32_t consumed = 0;
  unsigned long long last = dev_clock(), last_progress = last;
  uint32_t seen_cur = 0, seen_next = 0;
  for (;;) {
    uint32_t t = A.H->tail.load(std::memory_order_acquire);
    const uint32_t close_req = A.H->close_req.load(std::memory_order_relaxed);
    if (t > A.capacity) t = A.capacity;
    const unsigned long long now = dev_clock();
    // FAKE_POLLER_LAG_US (tests): a poller that is slow to pick new descriptors up -- a close request that arrives meanwhile then finds
    // published descriptors unconsumed, which the next epoch has to carry over (the path a failed carry launch is tested on)
    static const int lag_us = getenv("FAKE_POLLER_LAG_US") ? atoi(getenv("FAKE_POLLER_LAG_US")) : 0;
    bool asked_meanwhile = false;
    if (lag_us > 0 && t > consumed) {
      timespec ts = {0, (long)lag_us * 1000};
      nanosleep(&ts, nullptr);
      const uint32_t cr = A.H->close_req.load(std::memory_order_relaxed);
      asked_meanwhile = cr != 0 && cr == A.epoch;
    }
    if (t > consumed && !asked_meanwhile) {
      for (uint32_t d = consumed; d < t; ++d) {
        for (uint32_t w = 0; w < RING_DESC_WORDS; ++w) st(&A.d_desc[d].w[w], A.h_desc[d].w[w]);  // (ordered behind the acquire of the tail)
        st(&A.ctr[d].n_units, A.h_desc[d].w[0]);
        st((unsigned long long*)&A.ctr[d].t_pub, now);
      }
      st(&A.D->tail, t, __ATOMIC_RELEASE);  // fence(release, agent) ; store
      A.H->heartbeat.store(now, std::memory_order_relaxed);
      consumed = t;
      last = now;
      continue;
    }
    const uint32_t cur = ld(&A.D->cur);
    bool stalled = false;
    if (cur < consumed) {
      last = now;
      const uint32_t nxt = ld(&A.ctr[cur].next);
      if (cur != seen_cur || nxt != seen_next) { seen_cur = cur; seen_next = nxt; last_progress = now; }
      stalled = now - last_progress > 2000000ull * A.sleep_ticks_us;
    } else {
      last_progress = now;
    }
    const bool asked = (close_req != 0 && close_req == A.epoch) || stalled;
    const bool full = consumed >= A.capacity;
    if (asked || full || now - last > A.idle_ticks) {
      A.H->state.store(ring_state(A.epoch, consumed, RING_CLOSING), std::memory_order_seq_cst);
      uint32_t t2 = A.H->tail.load(std::memory_order_seq_cst);
      for (int look = 0; look < 3 && t2 <= consumed; ++look) { sched_yield(); t2 = A.H->tail.load(std::memory_order_seq_cst); }
      if (t2 > A.capacity) t2 = A.capacity;
      if (t2 > consumed && !asked && !full) {
        A.H->state.store(ring_state(A.epoch, consumed, RING_OPEN), std::memory_order_seq_cst);
        last = now;
        continue;
      }
      // (the device writes CLOSED first and the diagnostics behind it; a host thread may read them torn -- they are diagnostics.  Here
      // they go first so that the sanitizer has nothing to say about words nobody relies on)
      A.H->workers_seen.store(ld(&A.D->workers), std::memory_order_relaxed);
      A.H->close_reason.store(stalled ? 4u : (close_req != 0 && close_req == A.epoch) ? 1u : full ? 2u : 3u, std::memory_order_relaxed);
      A.H->state.store(ring_state(A.epoch, consumed, RING_CLOSED), std::memory_order_seq_cst);
      st(&A.D->quit, 1u, __ATOMIC_RELEASE);
      return;
    }
    sched_yield();
  }
}

// unit_fn(words of the descriptor, unit index): the unit's work -- it stores its results into the caller's block (system-scope stores on the device)
template <class UnitFn>
static void play_worker(const RingArgs A, const int id, UnitFn unit_fn) {  // bpsw_ring_dev.h: ring_next_unit / ring_unit_done around a unit of "work"
  uint32_t Wd = 0;
  bool idle = false, counted = false;
  unsigned long long idle_since = 0;
  std::minstd_rand rng(1234u + (unsigned)id);
  for (;;) {
    // ---- ring_next_unit
    uint32_t unit = 0;
    for (;;) {
      const uint32_t tail = ld(&A.D->tail, __ATOMIC_ACQUIRE), cur = ld(&A.D->cur), quit = ld(&A.D->quit, __ATOMIC_ACQUIRE);
      if (cur > Wd) Wd = cur;
      if (Wd >= tail) {
        if (quit != 0u) {
          const uint32_t t2 = ld(&A.D->tail, __ATOMIC_ACQUIRE);
          if (Wd >= t2) return;
          continue;
        }
        const unsigned long long now = dev_clock();
        if (!idle) { idle = true; idle_since = now; }
        else if (now - idle_since > A.worker_idle_ticks) return;
        sched_yield();
        continue;
      }
      idle = false;
      const uint32_t handed = ld(&A.ctr[Wd].next), n_units = ld(&A.ctr[Wd].n_units);
      if (handed >= n_units) { Wd += 1u; continue; }
      const uint32_t k = __atomic_fetch_add(&A.ctr[Wd].next, 1u, __ATOMIC_RELAXED);
      if (k >= n_units) { Wd += 1u; continue; }
      if (k + 1u == n_units) {  // fetch_max(cur, Wd + 1)
        uint32_t c = ld(&A.D->cur);
        while (c < Wd + 1u && !__atomic_compare_exchange_n(&A.D->cur, &c, Wd + 1u, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
      }
      unit = k;
      if (k == 0u) st((unsigned long long*)&A.ctr[Wd].t0, dev_clock());
      if (!counted) { __atomic_fetch_add(&A.D->workers, 1u, __ATOMIC_RELAXED); counted = true; }
      break;
    }
    // ---- the descriptor (mirrored by the poller before its release of D->tail, which this thread has acquired)
    uint32_t word[RING_DESC_WORDS];
    for (uint32_t w = 0; w < RING_DESC_WORDS; ++w) word[w] = ld(&A.d_desc[Wd].w[w]);
    RingDescHead head;
    memcpy(&head, word, sizeof head);
    // ---- the unit
    if ((rng() & 31u) == 0u) sched_yield();  // uneven units
    unit_fn(word, unit);
    // ---- ring_unit_done: s_waitcnt vmcnt(0) ; add -- a release, and the last adder has acquired every earlier one's
    const uint32_t before = __atomic_fetch_add(&A.ctr[Wd].done, 1u, __ATOMIC_ACQ_REL);
    if (before + 1u == head.n_units) {
      RingDone* r = (RingDone*)(uintptr_t)head.done_ptr;
      r->t_first.store(ld((unsigned long long*)&A.ctr[Wd].t0), std::memory_order_relaxed);
      r->t_done.store(dev_clock(), std::memory_order_relaxed);
      r->value.store(head.done_value, std::memory_order_release);  // s_waitcnt vmcnt(0) ; store
    }
  }
}


}  // namespace fake_ring
