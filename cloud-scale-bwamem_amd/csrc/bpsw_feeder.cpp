// bpsw_feeder.cpp -- the host side of bench.py's timed region: T native threads that call the HOST-BUFFER entry points of
// libbPSW_hip.so (bpsw_extend_batch, bpsw_matesw_group) the way T Spark task threads of one executor call the two JNI
// symbols (MemChainToAlignBatched.scala:175-176, MemSamPe.scala:2091-2092): one context per thread, one blocking call per
// wire batch / pair group.  Part of libbpsw_synth.so (the harness library), NOT of the product: it only knows the C ABI,
// which it receives as function pointers, so that nothing here links against the product or the oracle.
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

extern "C" {

typedef struct {
  int32_t kind;     // 0: extension wire batch, 1: rescue group
  int32_t rc;       // return code of the call (output)
  const void* in;   // kind 0: wire bytes; kind 1: const bpsw_rescue_group_t*
  size_t in_bytes;  // kind 0: wire size
  void* out;        // kind 0: int16[out_cap]; kind 1: int32 out_cnt[2 * group_size]
  void* out2;       // kind 1: bpsw_alnreg_t[out_cap]
  int64_t out_cap;
  int64_t out_total;  // kind 1: regions written (output)
  double ms;          // wall time of the call (output)
  double cpu_ms;      // CPU time the calling thread spent in the call (CLOCK_THREAD_CPUTIME_ID; output)
} bpsw_feed_item_t;

typedef int (*extend_fn)(void* ctx, const uint8_t* wire, size_t bytes, int16_t* out, size_t out_len);
// the two-step form the JNI shim uses (csrc/bpsw_jni.cpp, swExtendFPGAJNI): the wire bytes go straight into the context's pinned staging
// block -- GetByteArrayRegion there, a memcpy here -- and the results are read where the kernel wrote them
typedef int (*stage_fn)(void* ctx, size_t bytes, uint8_t** buf);
typedef int (*commit_fn)(void* ctx, size_t wire_bytes, const int16_t** out, size_t* out_len);
typedef int (*matesw_fn)(void* ctx, const void* opt, const void* group, int mode, int32_t* out_cnt, void* out_regs, int64_t out_cap,
                         int64_t* out_total);

struct bpsw_feeder {
  std::vector<std::thread> threads;
  std::vector<void*> ctxs;
  extend_fn f_ext = nullptr;
  stage_fn f_stage = nullptr;    // both set: extension items go through stage / commit like the shim's calls
  commit_fn f_commit = nullptr;
  matesw_fn f_grp = nullptr;
  const void* opt = nullptr;
  int mode = 0;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  uint64_t epoch = 0;
  int running = 0;
  bool quit = false;
  bpsw_feed_item_t* items = nullptr;
  int n_items = 0;
  long long n_tickets = 0;                       // n_items * repeats: ticket i runs item i % n_items
  std::vector<std::atomic<unsigned char>> busy;  // per item: a call on it is in flight (its output buffers are in use)
  std::atomic<long long> next{0};
  std::atomic<int> first_rc{0};
};

static double now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

static double thread_cpu_ms() {
  timespec ts;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

static void feeder_worker(bpsw_feeder* F, int t) {
  uint64_t seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(F->mu);
      F->cv_go.wait(lk, [&] { return F->quit || F->epoch != seen; });
      if (F->quit) return;
      seen = F->epoch;
    }
    for (;;) {
      const long long ticket = F->next.fetch_add(1, std::memory_order_relaxed);
      if (ticket >= F->n_tickets) break;
      const int i = (int)(ticket % F->n_items);
      bpsw_feed_item_t& it = F->items[i];
      // an item of the next repeat while a slow thread still runs it: wait for that call (its result buffers are the same)
      unsigned char idle = 0;
      while (!F->busy[(size_t)i].compare_exchange_weak(idle, 1, std::memory_order_acquire)) { idle = 0; sched_yield(); }
      const double t0 = now_ms(), c0 = thread_cpu_ms();
      if (it.kind == 0 && F->f_stage && F->f_commit) {
        uint8_t* buf = nullptr;
        const int16_t* res = nullptr;
        size_t res_len = 0;
        int rc = F->f_stage(F->ctxs[(size_t)t], it.in_bytes, &buf);
        if (rc == 0) {
          memcpy(buf, it.in, it.in_bytes);                                   // GetByteArrayRegion
          rc = F->f_commit(F->ctxs[(size_t)t], it.in_bytes, &res, &res_len);
        }
        if (rc == 0 && res_len > (size_t)it.out_cap) rc = -6;
        if (rc == 0 && res_len) memcpy(it.out, res, 2 * res_len);            // SetShortArrayRegion
        it.rc = rc;
      } else if (it.kind == 0) it.rc = F->f_ext(F->ctxs[(size_t)t], (const uint8_t*)it.in, it.in_bytes, (int16_t*)it.out, (size_t)it.out_cap);
      else it.rc = F->f_grp(F->ctxs[(size_t)t], F->opt, it.in, F->mode, (int32_t*)it.out, it.out2, it.out_cap, &it.out_total);
      it.ms = now_ms() - t0;
      it.cpu_ms = thread_cpu_ms() - c0;
      const int rc = it.rc;
      F->busy[(size_t)i].store(0, std::memory_order_release);
      if (rc != 0) {
        int zero = 0;
        F->first_rc.compare_exchange_strong(zero, rc);
      }
    }
    {
      std::lock_guard<std::mutex> lk(F->mu);
      if (--F->running == 0) F->cv_done.notify_all();
    }
  }
}

// ctxs: n_threads bpsw_ctx_t* (one per thread); cpus (optional): the n_cpus CPU ids every thread may run on (the NUMA node
// of the rank's GPU).  Threads are confined to the set, not pinned one per CPU: the box is shared, and the scheduler knows
// which of those CPUs are free.
bpsw_feeder* bpsw_feeder_create(int n_threads, void** ctxs, void* fn_extend, void* fn_matesw, const void* opt, int mode,
                                const int32_t* cpus, int n_cpus) {
  if (n_threads < 1 || !ctxs) return nullptr;
  bpsw_feeder* F = new bpsw_feeder();
  F->ctxs.assign(ctxs, ctxs + n_threads);
  F->f_ext = (extend_fn)fn_extend;
  F->f_grp = (matesw_fn)fn_matesw;
  F->opt = opt;
  F->mode = mode;
  for (int t = 0; t < n_threads; ++t) {
    F->threads.emplace_back(feeder_worker, F, t);
    if (cpus && n_cpus > 0) {
      cpu_set_t set;
      CPU_ZERO(&set);
      for (int k = 0; k < n_cpus; ++k) CPU_SET(cpus[k], &set);
      (void)pthread_setaffinity_np(F->threads.back().native_handle(), sizeof set, &set);
    }
  }
  return F;
}

// Runs every item `repeats` times (dynamic assignment: the next free thread takes the next item; after the last item the first
// one again, with no barrier in between -- task threads of an executor do not wait for each other -- but never two calls on one
// item at a time) and returns when all are done.  Returns the first non-zero return code of a call, or 0.
int bpsw_feeder_run_repeats(bpsw_feeder* F, bpsw_feed_item_t* items, int n_items, int repeats) {
  if (!F || n_items < 0 || repeats < 0) return -1;
  if (n_items == 0 || repeats == 0) return 0;
  {
    std::lock_guard<std::mutex> lk(F->mu);
    F->items = items;
    F->n_items = n_items;
    F->n_tickets = (long long)n_items * repeats;
    if (F->busy.size() < (size_t)n_items) F->busy = std::vector<std::atomic<unsigned char>>((size_t)n_items);
    for (int i = 0; i < n_items; ++i) F->busy[(size_t)i].store(0);
    F->next.store(0);
    F->first_rc.store(0);
    F->running = (int)F->threads.size();
    ++F->epoch;
  }
  F->cv_go.notify_all();
  std::unique_lock<std::mutex> lk(F->mu);
  F->cv_done.wait(lk, [&] { return F->running == 0; });
  return F->first_rc.load();
}

// extension items through bpsw_extend_stage / bpsw_extend_commit from now on (both null: back to bpsw_extend_batch)
void bpsw_feeder_use_stage_commit(bpsw_feeder* F, void* fn_stage, void* fn_commit) {
  if (!F) return;
  F->f_stage = (stage_fn)fn_stage;
  F->f_commit = (commit_fn)fn_commit;
}

int bpsw_feeder_run(bpsw_feeder* F, bpsw_feed_item_t* items, int n_items) { return bpsw_feeder_run_repeats(F, items, n_items, 1); }

void bpsw_feeder_destroy(bpsw_feeder* F) {
  if (!F) return;
  {
    std::lock_guard<std::mutex> lk(F->mu);
    F->quit = true;
  }
  F->cv_go.notify_all();
  for (auto& th : F->threads) th.join();
  delete F;
}

}  // extern "C"
