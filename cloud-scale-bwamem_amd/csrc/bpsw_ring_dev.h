// bpsw_ring_dev.h -- device side of the submission ring (bpsw_ring.h): the poller wavefront and the workers' claim / completion
// steps, shared by the resident kernels.  Every loop in here ends by itself: the poller after `idle_ticks` without a new descriptor
// (or at once on close_req), a worker on `quit` or after `worker_idle_ticks` without work.
#pragma once
#include <hip/hip_runtime.h>

#include "bpsw_ring.h"

namespace bpsw {

#define RING_SYS __HIP_MEMORY_SCOPE_SYSTEM
#define RING_DEV __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ uint32_t ring_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The poller: ONE wavefront of the epoch's kernel.  Mirrors newly published descriptors from pinned host memory into device
// memory (one 256-byte load per descriptor), publishes the count to the workers, and runs the closing handshake.
__device__ inline void ring_poller(const RingArgs& A, const int lane) {
  uint32_t consumed = 0;
  unsigned long long last = wall_clock64();
  // progress watch: with units outstanding, `cur` or the current descriptor's hand-out count must move; if neither has for two
  // seconds (no worker alive: cannot happen by construction) the epoch is closed anyway -- the kernel must end, the waiting
  // callers' watchdog reports the rest
  unsigned long long last_progress = last;
  uint32_t seen_cur = 0, seen_next = 0;
  uint32_t* h_tail = (uint32_t*)&A.H->tail;
  uint32_t* h_close = (uint32_t*)&A.H->close_req;
  unsigned long long* h_state = (unsigned long long*)&A.H->state;
  const auto state_word = [&](const uint32_t c, const unsigned long long phase) {
    return ((unsigned long long)(A.epoch & 0xffffffu) << 40) | ((unsigned long long)c << 8) | phase;
  };
  for (;;) {
    uint32_t t = ring_uni(__hip_atomic_load(h_tail, __ATOMIC_ACQUIRE, RING_SYS));
    const uint32_t close_req = ring_uni(__hip_atomic_load(h_close, __ATOMIC_RELAXED, RING_SYS));
    if (t > A.capacity) t = A.capacity;
    const unsigned long long now = wall_clock64();
    if (t > consumed) {
      for (uint32_t d = consumed; d < t; ++d) {
        const uint32_t w = __hip_atomic_load((uint32_t*)&A.h_desc[d].w[lane], __ATOMIC_RELAXED, RING_SYS);
        __hip_atomic_store(&A.d_desc[d].w[lane], w, __ATOMIC_RELAXED, RING_DEV);
        if (lane == 0) {
          __hip_atomic_store(&A.ctr[d].n_units, w, __ATOMIC_RELAXED, RING_DEV);  // word 0 = n_units
          __hip_atomic_store((unsigned long long*)&A.ctr[d].t_pub, now, __ATOMIC_RELAXED, RING_DEV);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // every lane's words before the count
      if (lane == 0) {
        __hip_atomic_store(&A.D->tail, t, __ATOMIC_RELEASE, RING_DEV);
        __hip_atomic_store((unsigned long long*)&A.H->heartbeat, now, __ATOMIC_RELAXED, RING_SYS);
      }
      consumed = t;
      last = now;
      continue;
    }
    const uint32_t cur = ring_uni(__hip_atomic_load(&A.D->cur, __ATOMIC_RELAXED, RING_DEV));
    bool stalled = false;
    if (cur < consumed) {  // units still to hand out: not idle
      last = now;
      const uint32_t nxt = ring_uni(__hip_atomic_load(&A.ctr[cur].next, __ATOMIC_RELAXED, RING_DEV));
      if (cur != seen_cur || nxt != seen_next) { seen_cur = cur; seen_next = nxt; last_progress = now; }
      stalled = now - last_progress > 2000000ull * A.sleep_ticks_us;
    } else {
      last_progress = now;
    }
    const bool asked = (close_req != 0 && close_req == A.epoch) || stalled;
    const bool full = consumed >= A.capacity;
    if (asked || full || now - last > A.idle_ticks) {
      // two-phase close (bpsw_ring.h): announce, look at the tail again, then either withdraw or commit
      if (lane == 0) __hip_atomic_store(h_state, state_word(consumed, RING_CLOSING), __ATOMIC_SEQ_CST, RING_SYS);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
      uint32_t t2 = ring_uni(__hip_atomic_load(h_tail, __ATOMIC_SEQ_CST, RING_SYS));
      for (int look = 0; look < 3 && t2 <= consumed; ++look) {  // margin: three more looks about a microsecond apart
        __builtin_amdgcn_s_sleep(32);
        t2 = ring_uni(__hip_atomic_load(h_tail, __ATOMIC_SEQ_CST, RING_SYS));
      }
      if (t2 > A.capacity) t2 = A.capacity;
      if (t2 > consumed && !asked && !full) {  // somebody published meanwhile: stay open and serve it
        if (lane == 0) __hip_atomic_store(h_state, state_word(consumed, RING_OPEN), __ATOMIC_SEQ_CST, RING_SYS);
        last = now;
        continue;
      }
      if (lane == 0) {
        __hip_atomic_store(h_state, state_word(consumed, RING_CLOSED), __ATOMIC_SEQ_CST, RING_SYS);
        __hip_atomic_store((uint32_t*)&A.H->workers_seen, __hip_atomic_load(&A.D->workers, __ATOMIC_RELAXED, RING_DEV), __ATOMIC_RELAXED, RING_SYS);
        __hip_atomic_store((uint32_t*)&A.H->close_reason, stalled ? 4u : (close_req != 0 && close_req == A.epoch) ? 1u : full ? 2u : 3u, __ATOMIC_RELAXED, RING_SYS);
        __hip_atomic_store((unsigned long long*)&A.H->diag_claim_ticks, __hip_atomic_load(&A.D->diag_claim_ticks, __ATOMIC_RELAXED, RING_DEV), __ATOMIC_RELAXED, RING_SYS);
        __hip_atomic_store((unsigned long long*)&A.H->diag_unit_ticks, __hip_atomic_load(&A.D->diag_unit_ticks, __ATOMIC_RELAXED, RING_DEV), __ATOMIC_RELAXED, RING_SYS);
        __hip_atomic_store((unsigned long long*)&A.H->diag_units, __hip_atomic_load(&A.D->diag_units, __ATOMIC_RELAXED, RING_DEV), __ATOMIC_RELAXED, RING_SYS);
        __hip_atomic_store(&A.D->quit, 1u, __ATOMIC_RELEASE, RING_DEV);
      }
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

struct RingWorker {
  unsigned long long diag_t_claim = 0;  // (diagnostics builds) when the current unit was taken
  uint32_t d = 0;  // first descriptor this wave has not seen drained
  bool idle = false, counted = false;
  unsigned long long idle_since = 0;
};

// The next unit of work for this wavefront: descriptor W.d, unit `unit`; `word` = this lane's word of the descriptor (field k
// of it: readlane(word, k)).  false: leave the kernel.
//
// What the waiting costs matters as much as what the work costs: thousands of worker waves are resident and, between batches, most of
// them wait.  The first version polled with acquire loads every 0.6 us and closed every unit with a system-scope release fence; an
// acquire / release at agent or system scope is a cache invalidate (buffer_inv) / an L2 write-back (buffer_wbl2) for the whole XCD, and
// at millions per second they slowed every kernel on the device down twenty-fold (the bench step 0.17 -> 1.3 s).  So: waiting is
// relaxed loads (sc1: served at the device's point of coherence, nothing invalidated) with a back-off to ~7 us; ONE invalidate of the
// CU's vector cache per unit taken (the staging blocks the unit reads are pinned host memory that callers reuse); and a unit's
// results are written through with system-scope stores, so that "visible to the host" is `s_waitcnt vmcnt(0)`, not a write-back.
__device__ inline bool ring_next_unit(const RingArgs& A, const int lane, RingWorker& W, uint32_t& unit, uint32_t& word) {
  int naps = 0;
  for (;;) {
    // tail, cur, quit with one load (lanes 0-2 a word each; a waiting wave costs the device one request per look)
    uint32_t ctl = 0;
    if (lane < 3) ctl = __hip_atomic_load(&A.D->tail + lane, __ATOMIC_RELAXED, RING_DEV);
    const uint32_t tail = (uint32_t)__builtin_amdgcn_readlane((int)ctl, 0);
    const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)ctl, 1);
    const uint32_t quit = (uint32_t)__builtin_amdgcn_readlane((int)ctl, 2);
    if (cur > W.d) W.d = cur;
    if (W.d >= tail) {
      if (quit != 0u) {
        // (the last count is published before quit, but the three words are three loads: look once more)
        const uint32_t t2 = ring_uni(__hip_atomic_load(&A.D->tail, __ATOMIC_RELAXED, RING_DEV));
        if (W.d >= t2) return false;
        continue;
      }
      const unsigned long long now = wall_clock64();
      if (!W.idle) { W.idle = true; W.idle_since = now; }
      else if (now - W.idle_since > A.worker_idle_ticks) return false;
      // Back off: 0.8 us for the first looks, then doubling to ~55 us.  A wave that has just finished a unit looks often; of the
      // waves that have been waiting for long, there are many, they look at different times, and a new batch's units are gone within
      // microseconds all the same -- while two thousand waves looking every microsecond kept the L2 channel of these words busy enough
      // to slow every kernel on the device down several times.
      __builtin_amdgcn_s_sleep(32);
      if (naps >= 6) {
        int rounds = naps >= 12 ? 16 : (1 << ((naps - 6) / 2 + 1)) / 2;  // 1, 1, 2, 2, 4, 4, then 16 x 3.4 us
        if (rounds > (int)A.nap_rounds_max) rounds = (int)A.nap_rounds_max;   // (BPSW_RING_NAP_ROUNDS)
        for (int r = 0; r < rounds; ++r) __builtin_amdgcn_s_sleep(127);
      }
      ++naps;
      continue;
    }
    W.idle = false;
    // Is anything of descriptor W.d left?  One load says so (the hand-out count and the poller's copy of the unit count share eight
    // bytes); only a wave that sees units left takes a ticket.  Without this every waiting wave -- thousands -- greeted every new
    // descriptor with a returning atomic on its counter and another on `cur`, and a batch's own units queued behind them.
    uint32_t cw = 0;
    if (lane < 2) cw = __hip_atomic_load(&A.ctr[W.d].next + lane, __ATOMIC_RELAXED, RING_DEV);
    const uint32_t handed = (uint32_t)__builtin_amdgcn_readlane((int)cw, 0), n_units = (uint32_t)__builtin_amdgcn_readlane((int)cw, 1);
    if (handed >= n_units) { W.d += 1u; continue; }
    uint32_t k = 0;
    if (lane == 0) k = __hip_atomic_fetch_add(&A.ctr[W.d].next, 1u, __ATOMIC_RELAXED, RING_DEV);
    k = ring_uni(k);
    if (k >= n_units) { W.d += 1u; continue; }
    if (k + 1u == n_units && lane == 0) __hip_atomic_fetch_max(&A.D->cur, W.d + 1u, __ATOMIC_RELAXED, RING_DEV);  // one per descriptor
    word = __hip_atomic_load(&A.d_desc[W.d].w[lane], __ATOMIC_RELAXED, RING_DEV);
    unit = k;
    if (lane == 0) {
      if (k == 0u) __hip_atomic_store((unsigned long long*)&A.ctr[W.d].t0, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, RING_DEV);
      if (!W.counted) __hip_atomic_fetch_add(&A.D->workers, 1u, __ATOMIC_RELAXED, RING_DEV);
    }
    W.counted = true;
#ifdef BPSW_RING_DIAG
    W.diag_t_claim = wall_clock64();
    if (lane == 0) {
      const unsigned long long t_pub = __hip_atomic_load((unsigned long long*)&A.ctr[W.d].t_pub, __ATOMIC_RELAXED, RING_DEV);
      __hip_atomic_fetch_add(&A.D->diag_claim_ticks, W.diag_t_claim - t_pub, __ATOMIC_RELAXED, RING_DEV);
    }
#endif
#ifndef BPSW_RING_ACQ
#define BPSW_RING_ACQ 1
#endif
    if (BPSW_RING_ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // nothing this CU's vector cache holds of an earlier batch's staging block
    return true;
  }
}

// After a unit's results have been stored by lane 0 with system-scope (write-through) stores: count it, and -- the last unit of its
// descriptor -- write the caller's completion record.  `word` as ring_next_unit returned it.  Every store of this lane has been
// acknowledged (s_waitcnt vmcnt(0)) before the count moves, the counts are one atomic sequence, so the lane that sees the last count
// writes the record after every unit's results.
__device__ inline void ring_unit_done(const RingArgs& A, const int lane, const RingWorker& W, const uint32_t word) {
  const uint32_t n_units = (uint32_t)__builtin_amdgcn_readlane((int)word, 0);
  const uint32_t done_value = (uint32_t)__builtin_amdgcn_readlane((int)word, 1);
  const unsigned long long done_ptr = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)word, 3) << 32) |
                                      (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)word, 2);
  if (lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef BPSW_RING_DIAG
    __hip_atomic_fetch_add(&A.D->diag_unit_ticks, (unsigned long long)wall_clock64() - W.diag_t_claim, __ATOMIC_RELAXED, RING_DEV);
    __hip_atomic_fetch_add(&A.D->diag_units, 1ull, __ATOMIC_RELAXED, RING_DEV);
#endif
    const uint32_t before = __hip_atomic_fetch_add(&A.ctr[W.d].done, 1u, __ATOMIC_RELAXED, RING_DEV);
    if (before + 1u == n_units) {
      RingDone* r = (RingDone*)done_ptr;
      const unsigned long long t0 = __hip_atomic_load((unsigned long long*)&A.ctr[W.d].t0, __ATOMIC_RELAXED, RING_DEV);
      __hip_atomic_store((unsigned long long*)&r->t_first, t0, __ATOMIC_RELAXED, RING_SYS);
      __hip_atomic_store((unsigned long long*)&r->t_done, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, RING_SYS);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store((uint32_t*)&r->value, done_value, __ATOMIC_RELAXED, RING_SYS);
    }
  }
}

}  // namespace bpsw
