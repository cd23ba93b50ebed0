// bpsw_sw_runtime.cpp -- C ABI entry points for the local-SW (mate rescue) jobs.
#include <string.h>

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
void ref_snapshot(const bpsw_ctx* c, const uint8_t** pac, long long* l_pac) {
  DeviceRef& r = device_ref(c->device);
  std::lock_guard<std::mutex> g(r.mu);
  *pac = (const uint8_t*)r.buf.ptr;
  *l_pac = r.l_pac;
}

// Stage the job table + pools in one pinned buffer, one H2D copy, one launch, one D2H copy.
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
  ref_snapshot(c, &d_pac, &l_pac);
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

  // layout of the staging block
  const size_t o_qlen = 0, o_tlen = align16(o_qlen + 4 * (size_t)n), o_qoff = align16(o_tlen + 4 * (size_t)n);
  const size_t o_toff = align16(o_qoff + 8 * (size_t)n), o_qrev = align16(o_toff + 8 * (size_t)n);
  const size_t o_qpool = align16(o_qrev + (size_t)n), o_tpool = align16(o_qpool + j->q_pool_bytes);
  const size_t total = align16(o_tpool + t_pool_bytes);
  const size_t out_bytes = 28 * (size_t)n;
  HIP_TRY(c->h_stage_in.reserve(total));
  HIP_TRY(c->d_sw_in.reserve(total));
  HIP_TRY(c->h_stage_out.reserve(out_bytes));
  HIP_TRY(c->d_sw_out.reserve(out_bytes));
  const size_t scratch = sw_scratch_bytes_per_wave(mt) * (size_t)sw_resident_waves(c->num_cu);
  HIP_TRY(c->d_sw_scratch.reserve(scratch));
  uint8_t* h = (uint8_t*)c->h_stage_in.ptr;
  memcpy(h + o_qlen, j->q_len, 4 * (size_t)n); memcpy(h + o_tlen, j->t_len, 4 * (size_t)n);
  memcpy(h + o_qoff, j->q_off, 8 * (size_t)n); memcpy(h + o_toff, j->t_off, 8 * (size_t)n);
  memcpy(h + o_qrev, j->q_rev, (size_t)n);
  memcpy(h + o_qpool, j->q_pool, j->q_pool_bytes);
  if (!pac_mode) memcpy(h + o_tpool, j->t_pool, t_pool_bytes);
  uint8_t* d = (uint8_t*)c->d_sw_in.ptr;
  SwJobsDev dev;
  dev.n = n;
  dev.q_len = (const int32_t*)(d + o_qlen); dev.t_len = (const int32_t*)(d + o_tlen);
  dev.q_off = (const int64_t*)(d + o_qoff); dev.t_off = (const int64_t*)(d + o_toff);
  dev.q_rev = d + o_qrev; dev.q_pool = d + o_qpool; dev.t_pool = pac_mode ? nullptr : d + o_tpool;
  dev.pac = d_pac; dev.l_pac = l_pac;

  HIP_TRY(hipEventRecord(c->ev[0], c->stream));
  HIP_TRY(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(c->ev[1], c->stream));
  HIP_TRY(launch_sw_kernel(dev, sc, mq, mt, (int32_t*)c->d_sw_out.ptr, (uint32_t*)c->d_sw_scratch.ptr, c->num_cu, c->stream));
  HIP_TRY(hipEventRecord(c->ev[2], c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_stage_out.ptr, c->d_sw_out.ptr, out_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipEventRecord(c->ev[3], c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(out, c->h_stage_out.ptr, out_bytes);
  float a = 0, b = 0, e = 0;
  (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
  (void)hipEventElapsedTime(&b, c->ev[1], c->ev[2]);
  (void)hipEventElapsedTime(&e, c->ev[2], c->ev[3]);
  c->stats.sw_calls++; c->stats.sw_jobs += (uint64_t)n;
  c->stats.sw_h2d_ms += a; c->stats.sw_kernel_ms += b; c->stats.sw_d2h_ms += e;
  c->last_sw_ms = b;
  c->have_sw_ev = false;
  return BPSW_OK;
}

}  // namespace bpsw

extern "C" {

int bpsw_swalign2_batch(bpsw_ctx_t* c, const bpsw_opt_t* opt, const bpsw_sw_jobs_t* jobs, int32_t* out) {
  if (!c || !jobs) return fail(BPSW_ERR_ARG, "swalign: null argument");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  return run_sw_jobs_host(c, opt, jobs, out);
}

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
  ref_snapshot(c, &d_pac, &l_pac);
  if (!j->t_pool && l_pac <= 0) return fail(BPSW_ERR_ARG, "swalign_device: t_pool is null and no reference is loaded (bpsw_ref_load)");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
  SwJobsDev dev;
  dev.n = j->n; dev.q_len = j->q_len; dev.t_len = j->t_len; dev.q_off = j->q_off; dev.t_off = j->t_off;
  dev.q_rev = j->q_rev; dev.q_pool = j->q_pool; dev.t_pool = j->t_pool;
  dev.pac = d_pac; dev.l_pac = l_pac;
  SwPrepass* d_pre = (SwPrepass*)c->d_pre.ptr;
  SwPrepass* h_pre = (SwPrepass*)c->h_pre.ptr;
  HIP_TRY(hipMemsetAsync(d_pre, 0, sizeof(SwPrepass), s));
  launch_sw_prepass(dev, j->q_pool_bytes, j->t_pool_bytes, d_pre, s);
  HIP_TRY(hipGetLastError());
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
  c->stats.sw_calls++; c->stats.sw_jobs += (uint64_t)j->n;
  return BPSW_OK;
}

// ---- the 2-bit reference resident in HBM -------------------------------------------------------
int bpsw_ref_load(bpsw_ctx_t* c, const uint8_t* pac, int64_t l_pac) {
  if (!c || !pac || l_pac < 1) return fail(BPSW_ERR_ARG, "ref_load: null reference or non-positive length");
  if (l_pac > (int64_t)1 << 40) return fail(BPSW_ERR_LIMIT, "ref_load: reference longer than 2^40 bases");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  DeviceRef& r = device_ref(c->device);
  std::lock_guard<std::mutex> gr(r.mu);
  const size_t bytes = (size_t)((l_pac + 3) >> 2);
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
  DeviceRef& r = device_ref(c->device);
  std::lock_guard<std::mutex> gr(r.mu);
  HIP_TRY(hipDeviceSynchronize());
  r.buf.release();
  r.l_pac = 0;
  return BPSW_OK;
}

int64_t bpsw_ref_length(const bpsw_ctx_t* c) {
  if (!c) return 0;
  const uint8_t* p = nullptr;
  long long l = 0;
  ref_snapshot(c, &p, &l);
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
  ref_snapshot(c, &d_pac, &l_pac);
  if (l_pac <= 0) return fail(BPSW_ERR_ARG, "ref_fetch: no reference is loaded (bpsw_ref_load)");
  HIP_TRY(hipSetDevice(c->device));
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

int bpsw_last_kernel_ms(bpsw_ctx_t* c, float* ext_ms, float* sw_ms) {
  if (!c) return fail(BPSW_ERR_ARG, "null context");
  std::lock_guard<std::mutex> g(c->mu);
  HIP_TRY(hipSetDevice(c->device));
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
