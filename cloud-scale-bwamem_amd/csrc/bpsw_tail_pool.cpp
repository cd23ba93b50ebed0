// worker2's tail off the calling thread (round 5): a library-owned pool of tail workers per device.  A task thread only ENQUEUES its
// groups of pairs (bpsw_tail_pool_submit: the arguments of bpsw_sam_pe_batch / bpsw_worker2_batch, copied by value; the arrays they
// point to stay the caller's and must not change until the ticket is collected) and collects them later (bpsw_tail_pool_wait), in
// any order.  Each worker owns a context -- a stream, pinned staging, the per-thread scratch of the tail -- and runs the same entry
// point the caller would have: the text is byte for byte what a direct call writes (tests/test_tail_gpu.py).
//
// What it is for: the tail of a group is plan (mark_primary, mem_pair: host) -> reg2aln kernel -> SAM text (host), 1.9 ms for 4 096 pairs
// of which the device has work for 0.46 ms; one calling thread that makes the calls itself is bound by its own host passes
// (4.2 M reads/s), and the reference's worker2 is exactly such a thread (worker2/MemSamPe.scala:1390-1612 runs in the partition's
// task thread, FastMap.scala:266-293).  With the pool the partition's thread enqueues and n workers overlap plan, kernel and text
// of different groups: profiles/r05_tail_pool.json.
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "bpsw_internal.h"

using namespace bpsw;

namespace {

struct TailJob {
  bpsw_opt_t opt;
  bpsw_tail_opt_t topt;
  bpsw_pairs_t g;
  int rescue_mode;  // BPSW_TAIL_POOL_TAIL_ONLY: bpsw_sam_pe_batch; else bpsw_worker2_batch with this rescue mode
  char* out_text; size_t text_cap; int64_t* out_off;
  int32_t* out_reg_cnt; bpsw_alnreg_t* out_regs; int64_t out_regs_cap;
  // results
  int rc = BPSW_OK; size_t needed = 0; int64_t regs_total = 0; std::string err; bool done = false;
};

}  // namespace

struct bpsw_tail_pool {
  int device = 0;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<int64_t> queue;
  std::map<int64_t, TailJob> jobs;  // ticket -> job, until collected
  int64_t next_ticket = 1;
  bool stop = false;
  int started = 0, failed_rc = BPSW_OK;
  std::string failed_err;
  std::vector<std::thread> workers;
};

namespace {

void tail_worker(bpsw_tail_pool* p) {
  bpsw_ctx_t* ctx = nullptr;
  const int rc0 = bpsw_create(p->device, &ctx);
  {
    std::lock_guard<std::mutex> g(p->mu);
    ++p->started;
    if (rc0 != BPSW_OK) { p->failed_rc = rc0; p->failed_err = bpsw_last_error(); }
  }
  p->cv_done.notify_all();
  if (rc0 != BPSW_OK) return;
  for (;;) {
    int64_t ticket;
    TailJob* j;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_work.wait(lk, [&] { return p->stop || !p->queue.empty(); });
      if (p->queue.empty()) break;  // stop, and nothing left to do
      ticket = p->queue.front();
      p->queue.pop_front();
      j = &p->jobs[ticket];  // (std::map: the node stays where it is while other tickets come and go)
    }
    int rc;
    size_t needed = 0;
    int64_t total = 0;
    if (j->rescue_mode == BPSW_TAIL_POOL_TAIL_ONLY)
      rc = bpsw_sam_pe_batch(ctx, &j->opt, &j->topt, &j->g, j->out_text, j->text_cap, j->out_off, &needed, j->out_regs);
    else
      rc = bpsw_worker2_batch(ctx, &j->opt, &j->topt, &j->g, j->rescue_mode, j->out_text, j->text_cap, j->out_off, &needed, j->out_reg_cnt,
                              j->out_regs, j->out_regs_cap, &total);
    {
      std::lock_guard<std::mutex> g(p->mu);
      j->rc = rc; j->needed = needed; j->regs_total = total;
      if (rc != BPSW_OK) j->err = bpsw_last_error();
      j->done = true;
    }
    p->cv_done.notify_all();
  }
  bpsw_destroy(ctx);
}

}  // namespace

extern "C" {

int bpsw_tail_pool_create(int device, int n_workers, bpsw_tail_pool_t** out) {
  if (!out) return fail(BPSW_ERR_ARG, "tail_pool_create: null argument");
  *out = nullptr;
  if (n_workers < 1 || n_workers > 256) return fail(BPSW_ERR_ARG, "tail_pool_create: 1..256 workers");
  bpsw_tail_pool* p = new bpsw_tail_pool();
  p->device = device;
  for (int i = 0; i < n_workers; ++i) p->workers.emplace_back(tail_worker, p);
  {  // every worker has its context (or the pool is refused: no device, no pool -- there is no CPU path behind it)
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->started == n_workers; });
  }
  if (p->failed_rc != BPSW_OK) {
    const int rc = p->failed_rc;
    const std::string err = p->failed_err;
    bpsw_tail_pool_destroy(p);
    return fail(rc, "tail_pool_create: " + err);
  }
  *out = p;
  return BPSW_OK;
}

void bpsw_tail_pool_destroy(bpsw_tail_pool_t* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> g(p->mu);
    p->stop = true;  // queued jobs are still run (their buffers are the callers'); tickets nobody collects are dropped
  }
  p->cv_work.notify_all();
  for (std::thread& t : p->workers) t.join();
  delete p;
}

int bpsw_tail_pool_submit(bpsw_tail_pool_t* p, const bpsw_opt_t* opt, const bpsw_tail_opt_t* topt, const bpsw_pairs_t* g, int rescue_mode,
                          char* out_text, size_t text_cap, int64_t* out_off, int32_t* out_reg_cnt, bpsw_alnreg_t* out_regs,
                          int64_t out_regs_cap, int64_t* ticket) {
  if (!p || !opt || !topt || !g || !out_off || !ticket) return fail(BPSW_ERR_ARG, "tail_pool_submit: null argument");
  if (rescue_mode != BPSW_TAIL_POOL_TAIL_ONLY && rescue_mode != BPSW_RESCUE_C && rescue_mode != BPSW_RESCUE_SCALA)
    return fail(BPSW_ERR_ARG, "tail_pool_submit: rescue_mode is BPSW_RESCUE_C, BPSW_RESCUE_SCALA or BPSW_TAIL_POOL_TAIL_ONLY");
  TailJob j;
  j.opt = *opt; j.topt = *topt; j.g = *g; j.rescue_mode = rescue_mode;
  j.out_text = out_text; j.text_cap = text_cap; j.out_off = out_off;
  j.out_reg_cnt = out_reg_cnt; j.out_regs = out_regs; j.out_regs_cap = out_regs_cap;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->stop) return fail(BPSW_ERR_ARG, "tail_pool_submit: the pool is being destroyed");
    *ticket = p->next_ticket++;
    p->jobs.emplace(*ticket, std::move(j));
    p->queue.push_back(*ticket);
  }
  p->cv_work.notify_one();
  return BPSW_OK;
}

int bpsw_tail_pool_wait(bpsw_tail_pool_t* p, int64_t ticket, size_t* out_needed, int64_t* out_regs_total) {
  if (!p) return fail(BPSW_ERR_ARG, "tail_pool_wait: null pool");
  int rc;
  std::string err;
  {
    std::unique_lock<std::mutex> lk(p->mu);
    auto it = p->jobs.find(ticket);
    if (it == p->jobs.end()) return fail(BPSW_ERR_ARG, "tail_pool_wait: unknown ticket (never issued, or collected already)");
    p->cv_done.wait(lk, [&] { return it->second.done; });
    rc = it->second.rc;
    err = std::move(it->second.err);
    if (out_needed) *out_needed = it->second.needed;
    if (out_regs_total) *out_regs_total = it->second.regs_total;
    p->jobs.erase(it);
  }
  return rc == BPSW_OK ? BPSW_OK : fail(rc, err);  // the worker's error text becomes the collecting thread's bpsw_last_error()
}

int bpsw_tail_pool_workers(const bpsw_tail_pool_t* p) { return p ? (int)p->workers.size() : 0; }

}  // extern "C"
