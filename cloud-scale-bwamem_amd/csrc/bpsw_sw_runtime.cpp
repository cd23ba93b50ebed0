// bpsw_sw_runtime.cpp -- C ABI entry points for the local-SW (mate rescue) jobs.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "bpsw_internal.h"

using namespace bpsw;

namespace {

int hip_fail(hipError_t e, const char* what) {
  return fail(BPSW_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                 \
  do {                                                \
    hipError_t e_ = (expr);                           \
    if (e_ != hipSuccess) return hip_fail(e_, #expr); \
  } while (0)

int make_scoring(const bpsw_opt_t* opt, int xtra, SwScoring* sc) {
  if (!opt) return fail(BPSW_ERR_ARG, "swalign: null options");
  if (opt->a < 1 || opt->o_del < 0 || opt->e_del < 0 || opt->o_ins < 0 || opt->e_ins < 0)
    return fail(BPSW_ERR_ARG, "swalign: scoring must have a >= 1 and non-negative gap penalties");
  sc->mat = pack_mat(opt->mat);
  sc->a = opt->a; sc->b = opt->b;
  sc->o_del = opt->o_del; sc->e_del = opt->e_del; sc->o_ins = opt->o_ins; sc->e_ins = opt->e_ins;
  sc->xtra = xtra;
  return BPSW_OK;
}

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace

namespace bpsw {

DeviceRef& device_ref(int device) {
  static DeviceRef table[64];
  return table[device >= 0 && device < 64 ? device : 0];
}
RefHold ref_snapshot(const bpsw_ctx* c, const uint8_t** pac, long long* l_pac) {
  DeviceRef& r = device_ref(c->device);
  RefHold hold(&r.gate);
  std::lock_guard<std::mutex> g(r.mu);
  *pac = (const uint8_t*)r.buf.ptr;
  *l_pac = r.l_pac;
  return hold;
}

// Layout of a staged SW batch in c->h_stage_in (job table SoA + the two byte pools, every part 16-byte aligned).
int sw_stage_begin(bpsw_ctx* c, int n, size_t q_pool_bytes, size_t t_pool_bytes, SwStage* st) {
  st->n = n;
  st->q_pool_bytes = q_pool_bytes; st->t_pool_bytes = t_pool_bytes;
  st->o_qlen = 0; st->o_tlen = align16(st->o_qlen + 4 * (size_t)n); st->o_qoff = align16(st->o_tlen + 4 * (size_t)n);
  st->o_toff = align16(st->o_qoff + 8 * (size_t)n); st->o_qrev = align16(st->o_toff + 8 * (size_t)n);
  st->o_qpool = align16(st->o_qrev + (size_t)n); st->o_tpool = align16(st->o_qpool + q_pool_bytes);
  st->o_packed = (st->o_tpool + t_pool_bytes + 63) & ~(size_t)63;  // job records for the packed kernel (sw_stage_run)
  st->total = align16(st->o_packed + 32 * (size_t)n);
  HIP_TRY(c->h_stage_in.reserve(st->total));
  HIP_TRY(c->h_stage_out.reserve(28 * (size_t)n));
  st->base = (uint8_t*)c->h_stage_in.ptr;
  return BPSW_OK;
}

// One launch over a batch staged by sw_stage_begin (and filled by the caller): H2D (or the kernel reads the pinned block
// itself), kernel, D2H.  *results = 7 int32 per job in the pinned result buffer, valid until the next call on the context.
static std::atomic<int> g_sw_in_flight[64];
int sw_launches_in_flight(int device) { return g_sw_in_flight[device >= 0 && device < 64 ? device : 0].load(std::memory_order_relaxed); }
void sw_launch_in_flight(int device, int delta) { g_sw_in_flight[device >= 0 && device < 64 ? device : 0].fetch_add(delta, std::memory_order_relaxed); }

int sw_stage_run(bpsw_ctx* c, const bpsw_opt_t* opt, int xtra, const SwStage& st, int mq, int mt, bool pac_mode, const int32_t** results) {
  if (c->ring_abandoned) return fail(BPSW_ERR_DEVICE, "this context gave up a ring batch (watchdog / integrity): create a new one");
  SwScoring sc;
  int rc = make_scoring(opt, xtra, &sc);
  if (rc != BPSW_OK) return rc;
  const int n = st.n;
  const uint8_t* d_pac = nullptr;
  long long l_pac = 0;
  const RefHold ref_hold = ref_snapshot(c, &d_pac, &l_pac);
  if (pac_mode && l_pac <= 0) return fail(BPSW_ERR_ARG, "swalign: t_pool is null and no reference is loaded (bpsw_ref_load)");
  if (mq > BPSW_SW_MAX_QLEN || mt > BPSW_SW_MAX_TLEN) return fail(BPSW_ERR_LIMIT, "swalign: sequence longer than the kernel limit");
  const size_t out_bytes = 28 * (size_t)n;
  // zero-copy (bpsw_runtime.cpp, zerocopy_mask): the kernel reads the staging block and writes its results over PCIe
  const bool zc_in = (zerocopy_mask() & 4) != 0, zc_out = (zerocopy_mask() & 2) != 0;
  if (!zc_in) HIP_TRY(c->d_sw_in.reserve(st.total));
  if (!zc_out) HIP_TRY(c->d_sw_out.reserve(out_bytes));
  const size_t scratch = sw_scratch_bytes_per_wave(mt) * (size_t)sw_resident_waves(c->num_cu);
  HIP_TRY(c->d_sw_scratch.reserve(scratch));
  uint8_t* h = st.base;
  uint8_t* d = zc_in ? h : (uint8_t*)c->d_sw_in.ptr;
  int32_t* k_out = zc_out ? (int32_t*)c->h_stage_out.ptr : (int32_t*)c->d_sw_out.ptr;
  SwJobsDev dev;
  dev.n = n;
  dev.q_len = (const int32_t*)(d + st.o_qlen); dev.t_len = (const int32_t*)(d + st.o_tlen);
  dev.q_off = (const int64_t*)(d + st.o_qoff); dev.t_off = (const int64_t*)(d + st.o_toff);
  dev.q_rev = d + st.o_qrev; dev.q_pool = d + st.o_qpool; dev.t_pool = pac_mode ? nullptr : d + st.o_tpool;
  dev.pac = d_pac; dev.l_pac = l_pac;
  {  // the table once more as 32-byte records (SwJobsDev::packed)
    const int32_t* ql = (const int32_t*)(h + st.o_qlen); const int32_t* tl = (const int32_t*)(h + st.o_tlen);
    const int64_t* qo = (const int64_t*)(h + st.o_qoff); const int64_t* to = (const int64_t*)(h + st.o_toff);
    const uint8_t* qr = h + st.o_qrev;
    uint32_t* rec = (uint32_t*)(h + st.o_packed);
    for (int i = 0; i < n; ++i, rec += 8) {
      memcpy(rec, qo + i, 8); memcpy(rec + 2, to + i, 8);
      rec[4] = (uint32_t)ql[i]; rec[5] = (uint32_t)tl[i]; rec[6] = qr[i]; rec[7] = 0;
    }
    dev.packed = (const uint32_t*)(d + st.o_packed);
  }
  // The submission ring (bpsw_ring.h): the batch becomes one descriptor of the device's resident rescue kernel -- no stream, no launch,
  // no event; the call waits on a completion record in its own pinned block.  Batches the resident kernel cannot take (a scoring the
  // packed kernel does not cover, mates above 256 bases, windows longer than its key rows) and BPSW_RING=0 go through a launch.
  int ring_bias = 0;
  int ring_class = (ring_enabled() && zc_in && zc_out) ? sw_ring_class(sc, mq, mt, &ring_bias) : 0;
  if (ring_class && !ring_usable(c->device, ring_class)) ring_class = 0;  // a ring that failed earlier: a launch per batch, as before round 5
  // A LONE caller with a sizeable batch -- no other SW batch in flight on the device, no extension call on it for 20 ms -- is better off
  // with a launch of its own: the epoch's grid is one or two waves per SIMD, a launch fills the device (256 / 1 024 / 4 096 pairs per
  // call, one calling thread: 0.28 / 0.33 / 0.48 ms through the ring, 0.23 / 0.27 / 0.38 with a launch; below sixteen jobs the ring wins:
  // tests/small_call_table.py).  As soon as callers overlap, or extension launches keep the queues busy, the ring it is.
  // (For a FEW overlapping callers the two are level -- groups of 1 024 / 4 096 pairs from two to six task threads: 7.2-7.5 / 3.7-4.1 k
  // groups/s at two, 13.5-14.7 / 7.4-7.6 k at four either way; from eight on the ring wins, 27 k against 17 k, and 48-53 k against 16 k at
  // sixteen.)  BPSW_RING_LONE_LAUNCH = n: a launch while fewer than n SW batches are in flight (default 1: the lone caller); 0: the ring
  // whenever it can.
  static const int lone_launch = getenv("BPSW_RING_LONE_LAUNCH") ? atoi(getenv("BPSW_RING_LONE_LAUNCH")) : 1;
  if (ring_class && lone_launch > 0 && n >= 16 && sw_launches_in_flight(c->device) < lone_launch && ext_call_age_ms(c->device) > 20.0) ring_class = 0;
  if (ring_class) {
    struct InFlight { int d; explicit InFlight(int dev) : d(dev) { sw_launch_in_flight(d, 1); } ~InFlight() { sw_launch_in_flight(d, -1); } } in_flight(c->device);
    const double t_dev0 = stat_ms();
    // Where the workers read the batch from.  A launch of its own reads it from the pinned block (zero-copy); under the resident kernel a
    // job pair that does so pays about eight dependent round trips over PCIe and takes 160-190 us instead of 145-155 (BPSW_RING_DIAG
    // build, one worker wave per SIMD; 390 against 320 with three), and the bench step is a fifth slower (1.73 against 2.05 x 10^8 reads/s at
    // two worker workgroups per CU).  So a batch is copied into the context's device arena by the copy engine -- one bulk transfer,
    // waited for before the descriptor is published -- and the workers read HBM.  Small batches (BPSW_RING_ZC_BYTES, default 16 KB) stay
    // zero-copy: the copy's latency would be most of their call.
    static const size_t ring_zc_bytes = getenv("BPSW_RING_ZC_BYTES") ? (size_t)atoll(getenv("BPSW_RING_ZC_BYTES")) : 16384;
    if (st.total > ring_zc_bytes) {
      HIP_TRY(c->d_sw_in.reserve(st.total));
      uint8_t* dd = (uint8_t*)c->d_sw_in.ptr;
      HIP_TRY(hipMemcpyAsync(dd, h, st.total, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipEventRecord(c->ev[0], c->stream));
      HIP_TRY(wait_event(c, c->ev[0], 2));
      dev.q_pool = dd + st.o_qpool; dev.t_pool = pac_mode ? nullptr : dd + st.o_tpool; dev.packed = (const uint32_t*)(dd + st.o_packed);
      c->stats.sw_h2d_ms += stat_ms() - t_dev0;  // (wall time of the copy as the caller saw it)
    }
    RingDesc desc;
    memset(&desc, 0, sizeof desc);
    RingDescHead head;
    memset(&head, 0, sizeof head);
    RingDone* done = (RingDone*)((char*)c->h_pre.ptr + 448);
    if (++c->ring_seq == 0) ++c->ring_seq;
    head.n_units = (uint32_t)((n + 1) / 2);
    head.done_value = c->ring_seq;
    head.done_ptr = (uint64_t)(uintptr_t)done;
    SwRingPayload pl;
    memset(&pl, 0, sizeof pl);
    pl.packed = (uint64_t)(uintptr_t)dev.packed; pl.q_pool = (uint64_t)(uintptr_t)dev.q_pool; pl.t_pool = (uint64_t)(uintptr_t)dev.t_pool;
    pl.pac = (uint64_t)(uintptr_t)dev.pac; pl.l_pac = dev.l_pac; pl.out = (uint64_t)(uintptr_t)k_out;
    pl.n_jobs = n; pl.bias = ring_bias;
    for (int r = 0; r < 5; ++r) pl.mat_row[r] = sc.mat.row[r];
    pl.a = sc.a; pl.b = sc.b; pl.o_del = sc.o_del; pl.e_del = sc.e_del; pl.o_ins = sc.o_ins; pl.e_ins = sc.e_ins; pl.xtra = sc.xtra;
    memcpy(desc.w, &head, sizeof head);
    memcpy(desc.w + sizeof(RingDescHead) / 4, &pl, sizeof pl);
    ring_poison((uint32_t*)k_out, 7, (size_t)n, 6);  // (score and qb of every record: the tripwire of bpsw_ring.cpp)
    rc = ring_submit(c->device, ring_class, c->num_cu, desc);
    if (rc != BPSW_OK && !ring_usable(c->device, ring_class)) {
      // the epoch could not be started (nothing of this batch has reached the device): this call and the later ones take a launch of their own
      static std::atomic<bool> said{false};
      if (!said.exchange(true)) fprintf(stderr, "bPSW: the submission ring of device %d failed (%s); SW batches are launched one by one from here on\n", c->device, bpsw_last_error());
      ring_class = 0;
      dev.q_pool = d + st.o_qpool; dev.t_pool = pac_mode ? nullptr : d + st.o_tpool; dev.packed = (const uint32_t*)(d + st.o_packed);
    } else {
      if (rc != BPSW_OK) return rc;
      rc = ring_wait(c->device, ring_class, done, c->ring_seq, &c->wait_est_ms[5]);
      if (rc == BPSW_RING_RELAUNCH) {  // another thread's epoch launch failed while this batch waited to be carried over: nobody will run it
        ring_class = 0;
        dev.q_pool = d + st.o_qpool; dev.t_pool = pac_mode ? nullptr : d + st.o_tpool; dev.packed = (const uint32_t*)(d + st.o_packed);
        goto launch_instead;
      }
      if (rc != BPSW_OK) { c->ring_abandoned = true; return rc; }  // (the descriptor still names this context's pinned blocks: see bpsw_destroy)
      rc = ring_check((const uint32_t*)k_out, 7, (size_t)n, 6, "rescue job");
      if (rc != BPSW_OK) { c->ring_abandoned = true; return rc; }
      const float span_ms = (float)((double)(done->t_done.load(std::memory_order_relaxed) - done->t_first.load(std::memory_order_relaxed)) / ring_ticks_per_ms(c->device, ring_class));
      c->stats.grp_dev_ms += stat_ms() - t_dev0;
      c->stats.sw_calls++; c->stats.sw_jobs += (uint64_t)n; c->stats.sw_ring_calls++;
      c->stats.sw_kernel_ms += span_ms;  // first unit taken -> last unit finished, on the device's clock
      c->last_sw_ms = span_ms;
      c->have_sw_ev = false;
      *results = (const int32_t*)c->h_stage_out.ptr;
      return BPSW_OK;
    }
  launch_instead:;
  }
  {
    struct InFlight { int d; explicit InFlight(int dev) : d(dev) { sw_launch_in_flight(d, 1); } ~InFlight() { sw_launch_in_flight(d, -1); } } in_flight(c->device);
    StreamLease lease(c);  // a device stream for the device phase only (bpsw_internal.h)
    hipStream_t s = lease.s;
    const double t_dev0 = stat_ms();
    // ev[1] / ev[2] ride on the kernel's own dispatch (KernelEvents): its begin and end as a kernel trace sees them, and two marker
    // packets fewer per call; only copies get markers of their own
    if (!zc_in) {
      HIP_TRY(hipEventRecord(c->ev[0], s));
      HIP_TRY(hipMemcpyAsync(d, h, st.total, hipMemcpyHostToDevice, s));
    }
    KernelEvents kev;
    kev.start = c->ev[1]; kev.stop = c->ev[2];
    HIP_TRY(launch_sw_kernel(dev, sc, mq, mt, k_out, (uint32_t*)c->d_sw_scratch.ptr, c->num_cu, s, nullptr, kev));
    if (!zc_out) {
      HIP_TRY(hipMemcpyAsync(c->h_stage_out.ptr, c->d_sw_out.ptr, out_bytes, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipEventRecord(c->ev[3], s));
    }
    HIP_TRY(wait_event(c, zc_out ? c->ev[2] : c->ev[3], 1));  // the last operation of the call on this stream
    c->stats.grp_wait_ms += lease.wait_ms;
    c->stats.grp_dev_ms += stat_ms() - t_dev0;
  }
  float a = 0, b = 0, e = 0;
  if (!zc_in) (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
  (void)hipEventElapsedTime(&b, c->ev[1], c->ev[2]);
  if (!zc_out) (void)hipEventElapsedTime(&e, c->ev[2], c->ev[3]);
  c->stats.sw_calls++; c->stats.sw_jobs += (uint64_t)n;
  c->stats.sw_h2d_ms += a; c->stats.sw_kernel_ms += b; c->stats.sw_d2h_ms += e;
  c->last_sw_ms = b;
  c->have_sw_ev = false;
  *results = (const int32_t*)c->h_stage_out.ptr;
  return BPSW_OK;
}

// Jobs whose arrays live in host memory: validate, stage the job table + pools in one pinned block, run.
int run_sw_jobs_host(bpsw_ctx* c, const bpsw_opt_t* opt, const bpsw_sw_jobs_t* j, int32_t* out) {
  SwScoring sc;
  int rc = make_scoring(opt, j->xtra, &sc);
  if (rc != BPSW_OK) return rc;
  const int n = j->n;
  if (n == 0) return BPSW_OK;
  if (n < 0 || !j->q_len || !j->t_len || !j->q_off || !j->t_off || !j->q_rev || !j->q_pool || !out)
    return fail(BPSW_ERR_ARG, "swalign: null job arrays");
  const bool pac_mode = j->t_pool == nullptr;  // windows named by coordinates
  const uint8_t* d_pac = nullptr;
  long long l_pac = 0;
  const RefHold ref_hold = ref_snapshot(c, &d_pac, &l_pac);
  if (pac_mode && l_pac <= 0) return fail(BPSW_ERR_ARG, "swalign: t_pool is null and no reference is loaded (bpsw_ref_load)");
  const size_t t_pool_bytes = pac_mode ? 0 : j->t_pool_bytes;
  int mq = 0, mt = 0;
  for (int i = 0; i < n; ++i) {  // host twin of sw_prepass_kernel
    const int ql = j->q_len[i], tl = j->t_len[i];
    const long long qo = j->q_off[i], to = j->t_off[i];
    const bool t_ok = pac_mode ? (to + tl <= (l_pac << 1) && (to >= l_pac || to + tl <= l_pac))
                               : (unsigned long long)(to + tl) <= j->t_pool_bytes;
    if (ql < 1 || tl < 0 || qo < 0 || to < 0 || (unsigned long long)(qo + ql) > j->q_pool_bytes || !t_ok)
      return fail(BPSW_ERR_ARG, pac_mode ? "swalign: window outside the loaded reference or bridging its strands"
                                         : "swalign: job sequence outside its pool");
    if (ql > mq) mq = ql;
    if (tl > mt) mt = tl;
  }
  if (mq > BPSW_SW_MAX_QLEN || mt > BPSW_SW_MAX_TLEN) return fail(BPSW_ERR_LIMIT, "swalign: sequence longer than the kernel limit");
  SwStage st;
  rc = sw_stage_begin(c, n, j->q_pool_bytes, t_pool_bytes, &st);
  if (rc != BPSW_OK) return rc;
  uint8_t* h = st.base;
  memcpy(h + st.o_qlen, j->q_len, 4 * (size_t)n); memcpy(h + st.o_tlen, j->t_len, 4 * (size_t)n);
  memcpy(h + st.o_qoff, j->q_off, 8 * (size_t)n); memcpy(h + st.o_toff, j->t_off, 8 * (size_t)n);
  memcpy(h + st.o_qrev, j->q_rev, (size_t)n);
  memcpy(h + st.o_qpool, j->q_pool, j->q_pool_bytes);
  if (!pac_mode) memcpy(h + st.o_tpool, j->t_pool, t_pool_bytes);
  const int32_t* res = nullptr;
  rc = sw_stage_run(c, opt, j->xtra, st, mq, mt, pac_mode, &res);
  if (rc != BPSW_OK) return rc;
  memcpy(out, res, 28 * (size_t)n);
  return BPSW_OK;
}

}  // namespace bpsw

extern "C" {

int bpsw_swalign2_batch(bpsw_ctx_t* c, const bpsw_opt_t* opt, const bpsw_sw_jobs_t* jobs, int32_t* out) {
  if (!c || !jobs) return fail(BPSW_ERR_ARG, "swalign: null argument");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  { const int prc_ = finish_pending(c); if (prc_ != BPSW_OK) return prc_; }
  return run_sw_jobs_host(c, opt, jobs, out);
}

extern "C++" {
namespace bpsw {
// Resolves an asynchronous SW launch of this context (see finish_pending_ext).
int finish_pending_sw(bpsw_ctx* c) {
  if (!c->pend_sw.active) return BPSW_OK;
  bpsw_ctx::PendingSw p = c->pend_sw;
  c->pend_sw.active = false;
  HIP_TRY(hipStreamSynchronize(p.s));
  if (p.verified) return BPSW_OK;
  const SwPrepass* h_pre = (const SwPrepass*)((const char*)c->h_pre.ptr + 256);
  if (h_pre->error) { c->have_sw_ev = false; return fail(BPSW_ERR_ARG, "swalign_device: job sequence outside its pool (or window outside / bridging the reference)"); }
  if (h_pre->max_qlen > BPSW_SW_MAX_QLEN || h_pre->max_tlen > BPSW_SW_MAX_TLEN) {
    c->have_sw_ev = false;
    return fail(BPSW_ERR_LIMIT, "swalign_device: sequence longer than the kernel limit");
  }
  c->sw_geom_qlen = h_pre->max_qlen; c->sw_geom_tlen = h_pre->max_tlen;
  if (h_pre->max_qlen > p.cap_qlen || h_pre->max_tlen > p.cap_tlen) {  // the kernel left the jobs untouched: launch for the real geometry
    SwScoring sc;
    int rc = make_scoring(&p.opt, p.jobs.xtra, &sc);
    if (rc != BPSW_OK) return rc;
    const uint8_t* d_pac = nullptr;
    long long l_pac = 0;
    const RefHold ref_hold = ref_snapshot(c, &d_pac, &l_pac);
    SwJobsDev dev;
    dev.n = p.jobs.n; dev.q_len = p.jobs.q_len; dev.t_len = p.jobs.t_len; dev.q_off = p.jobs.q_off; dev.t_off = p.jobs.t_off;
    dev.q_rev = p.jobs.q_rev; dev.q_pool = p.jobs.q_pool; dev.t_pool = p.jobs.t_pool; dev.pac = d_pac; dev.l_pac = l_pac;
    const size_t scratch = sw_scratch_bytes_per_wave(h_pre->max_tlen) * (size_t)sw_resident_waves(c->num_cu);
    HIP_TRY(c->d_sw_scratch.reserve(scratch));
    HIP_TRY(hipEventRecord(c->ev[6], p.s));
    HIP_TRY(launch_sw_kernel(dev, sc, h_pre->max_qlen, h_pre->max_tlen, (int32_t*)p.d_out, (uint32_t*)c->d_sw_scratch.ptr, c->num_cu, p.s));
    HIP_TRY(hipEventRecord(c->ev[7], p.s));
    HIP_TRY(hipStreamSynchronize(p.s));
    c->have_sw_ev = true;
  }
  return BPSW_OK;
}
}  // namespace bpsw
}  // extern "C++"

int bpsw_swalign2_batch_device(bpsw_ctx_t* c, const bpsw_opt_t* opt, const bpsw_sw_jobs_t* j, void* d_out, void* hip_stream) {
  if (!c || !j || !d_out) return fail(BPSW_ERR_ARG, "swalign_device: null argument");
  SwScoring sc;
  int rc = make_scoring(opt, j->xtra, &sc);
  if (rc != BPSW_OK) return rc;
  if (j->n == 0) return BPSW_OK;
  if (j->n < 0 || !j->q_len || !j->t_len || !j->q_off || !j->t_off || !j->q_rev || !j->q_pool)
    return fail(BPSW_ERR_ARG, "swalign_device: null job arrays");
  std::lock_guard<std::mutex> g(c->mu);
  const uint8_t* d_pac = nullptr;
  long long l_pac = 0;
  const RefHold ref_hold = ref_snapshot(c, &d_pac, &l_pac);
  if (!j->t_pool && l_pac <= 0) return fail(BPSW_ERR_ARG, "swalign_device: t_pool is null and no reference is loaded (bpsw_ref_load)");
  HIP_TRY(hipSetDevice(c->device));
  rc = finish_pending(c);
  if (rc != BPSW_OK) return rc;
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
  SwJobsDev dev;
  dev.n = j->n; dev.q_len = j->q_len; dev.t_len = j->t_len; dev.q_off = j->q_off; dev.t_off = j->t_off;
  dev.q_rev = j->q_rev; dev.q_pool = j->q_pool; dev.t_pool = j->t_pool;
  dev.pac = d_pac; dev.l_pac = l_pac;
  SwPrepass* d_pre = (SwPrepass*)((char*)c->d_pre.ptr + 256);
  SwPrepass* h_pre = (SwPrepass*)((char*)c->h_pre.ptr + 256);
  HIP_TRY(hipMemsetAsync(d_pre, 0, sizeof(SwPrepass), s));
  launch_sw_prepass(dev, j->q_pool_bytes, j->t_pool_bytes, d_pre, s);
  HIP_TRY(hipGetLastError());
  c->stats.sw_calls++; c->stats.sw_jobs += (uint64_t)j->n;
  if (c->sw_geom_qlen > 0) {
    // Asynchronous: the launch is sized for the geometry the previous call on this context was verified to have (the kernel
    // template by mate length, the scratch rows by window length) and checks the scan on the device; the scan is read back
    // at the next call on this context or at bpsw_last_kernel_ms, where a batch that outgrew the guess is launched again.
    const int cap_q = (c->sw_geom_qlen <= 160 && sw_quad_enabled()) ? 160 : 64 * ((c->sw_geom_qlen + 63) / 64);  // quad-job / sw_kernel<C>
    const int cap_t = (int)(c->d_sw_scratch.cap / ((size_t)sw_resident_waves(c->num_cu) * 16)) & ~63;  // rows the scratch holds per job
    if (cap_t >= c->sw_geom_tlen) {
      HIP_TRY(hipEventRecord(c->ev[6], s));
      HIP_TRY(launch_sw_kernel(dev, sc, cap_q, cap_t, (int32_t*)d_out, (uint32_t*)c->d_sw_scratch.ptr, c->num_cu, s, d_pre));
      HIP_TRY(hipEventRecord(c->ev[7], s));
      HIP_TRY(hipMemcpyAsync(h_pre, d_pre, sizeof(SwPrepass), hipMemcpyDeviceToHost, s));
      c->have_sw_ev = true;
      c->pend_sw.active = true; c->pend_sw.jobs = *j; c->pend_sw.opt = *opt; c->pend_sw.d_out = d_out; c->pend_sw.s = s;
      c->pend_sw.cap_qlen = cap_q; c->pend_sw.cap_tlen = cap_t; c->pend_sw.verified = false;
      return BPSW_OK;
    }
  }
  HIP_TRY(hipMemcpyAsync(h_pre, d_pre, sizeof(SwPrepass), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (h_pre->error) return fail(BPSW_ERR_ARG, "swalign_device: job sequence outside its pool (or window outside / bridging the reference)");
  if (h_pre->max_qlen > BPSW_SW_MAX_QLEN || h_pre->max_tlen > BPSW_SW_MAX_TLEN)
    return fail(BPSW_ERR_LIMIT, "swalign_device: sequence longer than the kernel limit");
  const size_t scratch = sw_scratch_bytes_per_wave(h_pre->max_tlen) * (size_t)sw_resident_waves(c->num_cu);
  if (scratch > c->d_sw_scratch.cap) {
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(c->d_sw_scratch.reserve(scratch));
  }
  HIP_TRY(hipEventRecord(c->ev[6], s));
  HIP_TRY(launch_sw_kernel(dev, sc, h_pre->max_qlen, h_pre->max_tlen, (int32_t*)d_out, (uint32_t*)c->d_sw_scratch.ptr,
                           c->num_cu, s));
  HIP_TRY(hipEventRecord(c->ev[7], s));
  c->have_sw_ev = true;
  c->sw_geom_qlen = h_pre->max_qlen; c->sw_geom_tlen = h_pre->max_tlen;
  // the launch uses the context's scratch rows and may sit on a caller's stream: the next call on the context (which may use
  // another stream) and bpsw_destroy wait for it through the same record the speculative path leaves
  c->pend_sw.active = true; c->pend_sw.jobs = *j; c->pend_sw.opt = *opt; c->pend_sw.d_out = d_out; c->pend_sw.s = s;
  c->pend_sw.verified = true;
  return BPSW_OK;
}

// ---- the 2-bit reference resident in HBM -------------------------------------------------------
int bpsw_ref_load(bpsw_ctx_t* c, const uint8_t* pac, int64_t l_pac) {
  if (!c || !pac || l_pac < 1) return fail(BPSW_ERR_ARG, "ref_load: null reference or non-positive length");
  if (l_pac > (int64_t)1 << 40) return fail(BPSW_ERR_LIMIT, "ref_load: reference longer than 2^40 bases");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  { const int prc_ = finish_pending(c); if (prc_ != BPSW_OK) return prc_; }
  DeviceRef& r = device_ref(c->device);
  RefWriteHold wr(&r.gate);  // no call may be between its snapshot and its last wait
  std::lock_guard<std::mutex> gr(r.mu);
  const size_t bytes = (size_t)((l_pac + 3) >> 2);
  // the resident kernels of the submission rings end their epochs first (other threads' calls that do not read the reference could
  // keep one alive for as long as they keep coming); the rings are locked until the reference has been replaced
  struct RingPause { int d; explicit RingPause(int dev) : d(dev) { ring_pause(d); } ~RingPause() { ring_resume(d); } } ring_paused(c->device);
  HIP_TRY(hipDeviceSynchronize());  // nothing in flight may still read the previous reference
  HIP_TRY(r.buf.reserve(bytes + 16));
  HIP_TRY(hipMemcpy(r.buf.ptr, pac, bytes, hipMemcpyHostToDevice));
  r.l_pac = l_pac;
  return BPSW_OK;
}

int bpsw_ref_unload(bpsw_ctx_t* c) {
  if (!c) return fail(BPSW_ERR_ARG, "null context");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  { const int prc_ = finish_pending(c); if (prc_ != BPSW_OK) return prc_; }
  DeviceRef& r = device_ref(c->device);
  RefWriteHold wr(&r.gate);
  std::lock_guard<std::mutex> gr(r.mu);
  struct RingPause { int d; explicit RingPause(int dev) : d(dev) { ring_pause(d); } ~RingPause() { ring_resume(d); } } ring_paused(c->device);
  HIP_TRY(hipDeviceSynchronize());
  r.buf.release();
  r.l_pac = 0;
  return BPSW_OK;
}

int64_t bpsw_ref_length(const bpsw_ctx_t* c) {
  if (!c) return 0;
  const uint8_t* p = nullptr;
  long long l = 0;
  const RefHold ref_hold = ref_snapshot(c, &p, &l);
  return (int64_t)l;
}

int bpsw_ref_fetch(bpsw_ctx_t* c, int32_t n, const int64_t* beg, const int64_t* end, uint8_t* out_pool, size_t out_pool_bytes,
                   const int64_t* out_off, int64_t* out_len) {
  if (!c || n < 0 || (n > 0 && (!beg || !end || !out_off || !out_len)) || (out_pool_bytes > 0 && !out_pool))
    return fail(BPSW_ERR_ARG, "ref_fetch: null argument");
  if (n == 0) return BPSW_OK;
  std::lock_guard<std::mutex> g(c->mu);
  const uint8_t* d_pac = nullptr;
  long long l_pac = 0;
  const RefHold ref_hold = ref_snapshot(c, &d_pac, &l_pac);
  if (l_pac <= 0) return fail(BPSW_ERR_ARG, "ref_fetch: no reference is loaded (bpsw_ref_load)");
  HIP_TRY(hipSetDevice(c->device));
  { const int prc_ = finish_pending(c); if (prc_ != BPSW_OK) return prc_; }
  const size_t o_beg = 0, o_end = align16(8 * (size_t)n), o_off = align16(o_end + 8 * (size_t)n);
  const size_t in_total = align16(o_off + 8 * (size_t)n);
  const size_t o_len = 0, o_err = align16(8 * (size_t)n), o_pool = align16(o_err + 16);
  const size_t out_total = align16(o_pool + out_pool_bytes);
  HIP_TRY(c->h_stage_in.reserve(in_total));
  HIP_TRY(c->d_sw_in.reserve(in_total));
  HIP_TRY(c->h_stage_out.reserve(out_total));
  HIP_TRY(c->d_sw_out.reserve(out_total));
  uint8_t* h = (uint8_t*)c->h_stage_in.ptr;
  memcpy(h + o_beg, beg, 8 * (size_t)n); memcpy(h + o_end, end, 8 * (size_t)n); memcpy(h + o_off, out_off, 8 * (size_t)n);
  uint8_t* d = (uint8_t*)c->d_sw_in.ptr;
  uint8_t* dout = (uint8_t*)c->d_sw_out.ptr;
  HIP_TRY(hipMemcpyAsync(d, h, in_total, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(dout + o_err, 0, 16, c->stream));
  launch_ref_fetch(d_pac, l_pac, n, (const long long*)(d + o_beg), (const long long*)(d + o_end),
                   dout + o_pool, out_pool_bytes, (const long long*)(d + o_off), (long long*)(dout + o_len),
                   (int*)(dout + o_err), c->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(c->h_stage_out.ptr, dout, out_total, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const uint8_t* ho = (const uint8_t*)c->h_stage_out.ptr;
  memcpy(out_len, ho + o_len, 8 * (size_t)n);
  if (*(const int*)(ho + o_err)) return fail(BPSW_ERR_CAPACITY, "ref_fetch: a window does not fit in out_pool at its out_off");
  if (out_pool_bytes) memcpy(out_pool, ho + o_pool, out_pool_bytes);
  return BPSW_OK;
}

// ---- memChainToAlnBatched on the device ------------------------------------------------------------
int bpsw_chain2aln_batch(bpsw_ctx_t* c, const bpsw_opt_t* opt, const bpsw_chains_t* b, int zdrop_mode, int flags, int32_t* out_cnt,
                         bpsw_alnreg_t* out_regs, int64_t out_cap, int64_t* out_total) {
  if (!c || !opt || !b || !out_cnt || !out_total) return fail(BPSW_ERR_ARG, "chain2aln: null argument");
  if (zdrop_mode != BPSW_ZDROP_SCALA && zdrop_mode != BPSW_ZDROP_BWA) return fail(BPSW_ERR_ARG, "chain2aln: bad zdrop_mode");
  const int n = b->n_reads;
  *out_total = 0;
  if (n == 0) return BPSW_OK;
  if (n < 0 || !b->read_len || !b->read_off || !b->read_pool || !b->chain_cnt) return fail(BPSW_ERR_ARG, "chain2aln: null read arrays");
  if (opt->a < 1 || opt->o_del < 0 || opt->o_ins < 0 || opt->e_del < 1 || opt->e_ins < 1)
    return fail(BPSW_ERR_ARG, "chain2aln: scoring must have a >= 1, gap opens >= 0 and gap extensions >= 1");
  if (opt->w < 1 || opt->w > 254) return fail(BPSW_ERR_LIMIT, "chain2aln: band width must be 1..254");
  std::lock_guard<std::mutex> g(c->mu);
  const uint8_t* d_pac = nullptr;
  long long l_pac = 0;
  const RefHold ref_hold = ref_snapshot(c, &d_pac, &l_pac);
  if (l_pac <= 0) return fail(BPSW_ERR_ARG, "chain2aln: no reference is loaded (bpsw_ref_load)");
  HIP_TRY(hipSetDevice(c->device));
  { const int prc_ = finish_pending(c); if (prc_ != BPSW_OK) return prc_; }

  // ---- host twin of a table scan: validate, build the prefix arrays the kernel indexes with ----
  std::vector<int32_t> chain_base((size_t)n);
  std::vector<long long> reg_base((size_t)n);
  long long nchains = 0, nseeds = 0;
  int max_seeds = 1;
  for (int r = 0; r < n; ++r) {
    const int ql = b->read_len[r];
    const long long qo = b->read_off[r];
    if (ql < 1 || qo < 0 || (unsigned long long)(qo + ql) > b->read_pool_bytes) return fail(BPSW_ERR_ARG, "chain2aln: read outside read_pool");
    if (ql > 256) return fail(BPSW_ERR_LIMIT, "chain2aln: read longer than 256 bases");
    if (b->chain_cnt[r] < 0) return fail(BPSW_ERR_ARG, "chain2aln: negative chain count");
    chain_base[(size_t)r] = (int32_t)nchains;
    nchains += b->chain_cnt[r];
    if (nchains > 0x7fffffffll) return fail(BPSW_ERR_LIMIT, "chain2aln: too many chains in one batch");
  }
  if (nchains > 0 && (!b->seed_cnt || !b->seed_rbeg || !b->seed_qbeg || !b->seed_len)) return fail(BPSW_ERR_ARG, "chain2aln: null seed arrays");
  std::vector<long long> seed_base((size_t)(nchains > 0 ? nchains : 1));
  {
    long long ch = 0;
    for (int r = 0; r < n; ++r) {
      reg_base[(size_t)r] = nseeds;
      const int ql = b->read_len[r];
      for (int k = 0; k < b->chain_cnt[r]; ++k, ++ch) {
        const int ns = b->seed_cnt[ch];
        if (ns < 0) return fail(BPSW_ERR_ARG, "chain2aln: negative seed count");
        seed_base[(size_t)ch] = nseeds;
        const bool fwd = ns > 0 && b->seed_rbeg[nseeds] < l_pac;
        for (int i = 0; i < ns; ++i) {
          const long long rb = b->seed_rbeg[nseeds + i];
          const int qb = b->seed_qbeg[nseeds + i], ln = b->seed_len[nseeds + i];
          if (ln < 1 || qb < 0 || qb + ln > ql) return fail(BPSW_ERR_ARG, "chain2aln: seed outside its read");
          if (rb < 0 || rb + ln > (l_pac << 1) || (rb < l_pac) != fwd || (fwd && rb + ln > l_pac))
            return fail(BPSW_ERR_ARG, "chain2aln: seed outside the reference or chain across both strands");
        }
        if (ns > max_seeds) max_seeds = ns;
        nseeds += ns;
      }
    }
  }
  *out_total = nseeds;  // upper bound until the kernel has run: one region per seed at most
  if (nseeds > out_cap || (nseeds > 0 && !out_regs)) return fail(BPSW_ERR_CAPACITY, "chain2aln: out_regs must hold one region per seed");

  // ---- one staging block, one H2D copy ----
  const size_t o_rlen = 0, o_roff = align16(4 * (size_t)n), o_ccnt = align16(o_roff + 8 * (size_t)n);
  const size_t o_cbase = align16(o_ccnt + 4 * (size_t)n), o_rbase = align16(o_cbase + 4 * (size_t)n);
  const size_t o_scnt = align16(o_rbase + 8 * (size_t)n), o_sbase = align16(o_scnt + 4 * (size_t)nchains);
  const size_t o_srb = align16(o_sbase + 8 * (size_t)nchains), o_sqb = align16(o_srb + 8 * (size_t)nseeds);
  const size_t o_sln = align16(o_sqb + 4 * (size_t)nseeds), o_pool = align16(o_sln + 4 * (size_t)nseeds);
  const size_t total = align16(o_pool + b->read_pool_bytes);
  const size_t o_cnt = 0, o_regs = align16(4 * (size_t)n);
  const size_t out_bytes = o_regs + sizeof(bpsw_alnreg_t) * (size_t)(nseeds > 0 ? nseeds : 1);
  const int srt_per_wave = (max_seeds + 15) & ~15;
  HIP_TRY(c->h_stage_in.reserve(total));
  HIP_TRY(c->d_sw_in.reserve(total));
  HIP_TRY(c->h_stage_out.reserve(out_bytes));
  HIP_TRY(c->d_sw_out.reserve(out_bytes));
  HIP_TRY(c->d_sw_scratch.reserve(4 * (size_t)srt_per_wave * (size_t)chain2aln_resident_waves(c->num_cu)));
  uint8_t* h = (uint8_t*)c->h_stage_in.ptr;
  memcpy(h + o_rlen, b->read_len, 4 * (size_t)n); memcpy(h + o_roff, b->read_off, 8 * (size_t)n);
  memcpy(h + o_ccnt, b->chain_cnt, 4 * (size_t)n); memcpy(h + o_cbase, chain_base.data(), 4 * (size_t)n);
  memcpy(h + o_rbase, reg_base.data(), 8 * (size_t)n);
  if (nchains) { memcpy(h + o_scnt, b->seed_cnt, 4 * (size_t)nchains); memcpy(h + o_sbase, seed_base.data(), 8 * (size_t)nchains); }
  if (nseeds) {
    memcpy(h + o_srb, b->seed_rbeg, 8 * (size_t)nseeds); memcpy(h + o_sqb, b->seed_qbeg, 4 * (size_t)nseeds);
    memcpy(h + o_sln, b->seed_len, 4 * (size_t)nseeds);
  }
  memcpy(h + o_pool, b->read_pool, b->read_pool_bytes);
  uint8_t* d = (uint8_t*)c->d_sw_in.ptr;
  uint8_t* dout = (uint8_t*)c->d_sw_out.ptr;
  ChainBatchDev B;
  B.n_reads = n;
  B.read_len = (const int32_t*)(d + o_rlen); B.read_off = (const long long*)(d + o_roff); B.read_pool = d + o_pool;
  B.chain_cnt = (const int32_t*)(d + o_ccnt); B.chain_base = (const int32_t*)(d + o_cbase);
  B.seed_cnt = (const int32_t*)(d + o_scnt); B.seed_base = (const long long*)(d + o_sbase);
  B.seed_rbeg = (const long long*)(d + o_srb); B.seed_qbeg = (const int32_t*)(d + o_sqb); B.seed_len = (const int32_t*)(d + o_sln);
  B.reg_base = (const long long*)(d + o_rbase);
  B.pac = d_pac; B.l_pac = l_pac;
  ChainParams P;
  P.mat = pack_mat(opt->mat);
  P.mat_max = opt->mat[0];
  for (int k = 1; k < 25; ++k) P.mat_max = opt->mat[k] > P.mat_max ? opt->mat[k] : P.mat_max;
  P.a = opt->a; P.o_del = opt->o_del; P.e_del = opt->e_del; P.o_ins = opt->o_ins; P.e_ins = opt->e_ins;
  P.pen_clip5 = opt->pen_clip5; P.pen_clip3 = opt->pen_clip3; P.w = opt->w; P.zdrop = opt->zdrop; P.zmode = zdrop_mode;
  apply_shortcuts(c->shortcut_mask, opt->mat, &P.exact_a, &P.certify, &P.tail_bound);

  HIP_TRY(hipEventRecord(c->ev[0], c->stream));
  HIP_TRY(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(c->ev[1], c->stream));
  HIP_TRY(launch_chain2aln_kernel(B, P, (bpsw_alnreg_t*)(dout + o_regs), (int32_t*)(dout + o_cnt), (int32_t*)c->d_sw_scratch.ptr,
                                  srt_per_wave, c->num_cu, (int*)((char*)c->d_pre.ptr + 384), c->stream));  // its own queue head: +128 belongs to ext_kernel, which expects it zero
  HIP_TRY(hipEventRecord(c->ev[2], c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_stage_out.ptr, dout, out_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, c->ev[1], c->ev[2]);
  c->stats.ext_kernel_ms += ms;
  c->last_ext_ms = ms;
  c->stats.ext_calls++; c->stats.ext_tasks += (uint64_t)nseeds;

  // ---- compact (and optionally memSortAndDedup, as bwaMemWorker1Batched does right after, BWAMemWorker1Batched.scala:128-133) ----
  const uint8_t* ho = (const uint8_t*)c->h_stage_out.ptr;
  const int32_t* cnt = (const int32_t*)(ho + o_cnt);
  const bpsw_alnreg_t* regs = (const bpsw_alnreg_t*)(ho + o_regs);
  int64_t at = 0;
  std::vector<bpsw_alnreg_t> v;
  for (int r = 0; r < n; ++r) {
    const bpsw_alnreg_t* first = regs + reg_base[(size_t)r];
    int m = cnt[r];
    if (flags & BPSW_C2A_SORT_DEDUP) {
      v.assign(first, first + m);
      m = sort_dedup_regs(v, opt->mask_level_redun, (flags & BPSW_C2A_DEDUP_SCALA) ? BPSW_RESCUE_SCALA : BPSW_RESCUE_C);
      first = v.data();
    }
    out_cnt[r] = m;
    for (int k = 0; k < m; ++k) out_regs[at++] = first[k];
  }
  *out_total = at;
  return BPSW_OK;
}

int bpsw_global_batch(bpsw_ctx_t* c, const bpsw_opt_t* opt, const bpsw_global_jobs_t* j, int32_t* out_score,
                      int32_t* out_ncigar, uint32_t* out_cigar) {
  if (!c || !j || !out_score || !out_ncigar || !out_cigar) return fail(BPSW_ERR_ARG, "global: null argument");
  SwScoring sc;
  int rc = make_scoring(opt, 0, &sc);
  if (rc != BPSW_OK) return rc;
  const int n = j->n;
  if (n == 0) return BPSW_OK;
  if (n < 0 || !j->q_len || !j->t_len || !j->w || !j->q_off || !j->t_off || !j->q_pool || !j->t_pool)
    return fail(BPSW_ERR_ARG, "global: null job arrays");
  if (j->max_cigar < 1 || j->max_cigar > BPSW_GLOBAL_MAX_CIGAR) return fail(BPSW_ERR_ARG, "global: max_cigar must be 1..512");
  int mq = 0;
  size_t mz = 0;
  for (int i = 0; i < n; ++i) {  // host twin of global_prepass_kernel
    const int ql = j->q_len[i], tl = j->t_len[i], w = j->w[i];
    const long long qo = j->q_off[i], to = j->t_off[i];
    if (ql < 1 || tl < 1 || w < 0 || qo < 0 || to < 0 || (unsigned long long)(qo + ql) > j->q_pool_bytes ||
        (unsigned long long)(to + tl) > j->t_pool_bytes)
      return fail(BPSW_ERR_ARG, "global: job sequence outside its pool (or empty)");
    if (ql > BPSW_GLOBAL_MAX_QLEN || tl > BPSW_GLOBAL_MAX_TLEN) return fail(BPSW_ERR_LIMIT, "global: sequence longer than the kernel limit");
    const long long ncol = ql < 2ll * w + 1 ? ql : 2ll * w + 1;
    if ((size_t)(ncol * tl) > mz) mz = (size_t)(ncol * tl);
    if (ql > mq) mq = ql;
  }
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  { const int prc_ = finish_pending(c); if (prc_ != BPSW_OK) return prc_; }
  const size_t o_qlen = 0, o_tlen = align16(4 * (size_t)n), o_w = align16(o_tlen + 4 * (size_t)n);
  const size_t o_qoff = align16(o_w + 4 * (size_t)n), o_toff = align16(o_qoff + 8 * (size_t)n);
  const size_t o_qpool = align16(o_toff + 8 * (size_t)n), o_tpool = align16(o_qpool + j->q_pool_bytes);
  const size_t total = align16(o_tpool + j->t_pool_bytes);
  const size_t o_score = 0, o_nc = align16(4 * (size_t)n), o_cig = align16(o_nc + 4 * (size_t)n);
  const size_t out_bytes = o_cig + 4 * (size_t)n * (size_t)j->max_cigar;
  const size_t z_per_wave = (mz + 255) & ~(size_t)255;
  const int qcap = (mq + 31) & ~31;
  HIP_TRY(c->h_stage_in.reserve(total));
  HIP_TRY(c->d_sw_in.reserve(total));
  HIP_TRY(c->h_stage_out.reserve(out_bytes));
  HIP_TRY(c->d_sw_out.reserve(out_bytes));
  HIP_TRY(c->d_gl_z.reserve(z_per_wave * (size_t)global_resident_waves(c->num_cu, qcap)));
  uint8_t* h = (uint8_t*)c->h_stage_in.ptr;
  memcpy(h + o_qlen, j->q_len, 4 * (size_t)n); memcpy(h + o_tlen, j->t_len, 4 * (size_t)n); memcpy(h + o_w, j->w, 4 * (size_t)n);
  memcpy(h + o_qoff, j->q_off, 8 * (size_t)n); memcpy(h + o_toff, j->t_off, 8 * (size_t)n);
  memcpy(h + o_qpool, j->q_pool, j->q_pool_bytes); memcpy(h + o_tpool, j->t_pool, j->t_pool_bytes);
  uint8_t* d = (uint8_t*)c->d_sw_in.ptr;
  GlobalJobsDev dev;
  dev.n = n; dev.max_cigar = j->max_cigar;
  dev.q_len = (const int32_t*)(d + o_qlen); dev.t_len = (const int32_t*)(d + o_tlen); dev.w = (const int32_t*)(d + o_w);
  dev.q_off = (const int64_t*)(d + o_qoff); dev.t_off = (const int64_t*)(d + o_toff);
  dev.q_pool = d + o_qpool; dev.t_pool = d + o_tpool;
  uint8_t* dout = (uint8_t*)c->d_sw_out.ptr;
  HIP_TRY(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(launch_global_kernel(dev, sc, mq, z_per_wave, (int32_t*)(dout + o_score), (int32_t*)(dout + o_nc),
                               (uint32_t*)(dout + o_cig), (uint8_t*)c->d_gl_z.ptr, c->num_cu, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_stage_out.ptr, dout, out_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const uint8_t* ho = (const uint8_t*)c->h_stage_out.ptr;
  memcpy(out_score, ho + o_score, 4 * (size_t)n);
  memcpy(out_ncigar, ho + o_nc, 4 * (size_t)n);
  memcpy(out_cigar, ho + o_cig, 4 * (size_t)n * (size_t)j->max_cigar);
  return BPSW_OK;
}

int bpsw_ring_integrity(uint64_t* checked, uint64_t* faults) {
  ring_integrity_stats(checked, faults);
  return ring_integrity_on() ? 1 : 0;
}

int bpsw_sw_batches_in_flight(int device) { return device >= 0 && device < 64 ? bpsw::sw_launches_in_flight(device) : -1; }

int bpsw_ring_stats(bpsw_ctx_t* c, uint64_t* epochs, uint64_t* submitted, uint64_t* carried, double* epochs_ms, uint64_t* epochs_timed) {
  if (!c) return fail(BPSW_ERR_ARG, "null context");
  HIP_TRY(hipSetDevice(c->device));
  ring_get_stats(c->device, epochs, submitted, carried, epochs_ms, epochs_timed);
  return BPSW_OK;
}

int bpsw_last_kernel_ms(bpsw_ctx_t* c, float* ext_ms, float* sw_ms) {
  if (!c) return fail(BPSW_ERR_ARG, "null context");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  { int prc = finish_pending(c); if (prc != BPSW_OK) return prc; }  // deferred errors of the asynchronous device entries surface here
  if (c->have_ext_ev) {
    HIP_TRY(hipEventSynchronize(c->ev[5]));
    HIP_TRY(hipEventElapsedTime(&c->last_ext_ms, c->ev[4], c->ev[5]));
    c->stats.ext_kernel_ms += c->last_ext_ms;
    c->have_ext_ev = false;
  }
  if (c->have_sw_ev) {
    HIP_TRY(hipEventSynchronize(c->ev[7]));
    HIP_TRY(hipEventElapsedTime(&c->last_sw_ms, c->ev[6], c->ev[7]));
    c->stats.sw_kernel_ms += c->last_sw_ms;
    c->have_sw_ev = false;
  }
  if (ext_ms) *ext_ms = c->last_ext_ms;
  if (sw_ms) *sw_ms = c->last_sw_ms;
  return BPSW_OK;
}

}  // extern "C"
