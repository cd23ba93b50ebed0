// bpsw_internal.h -- declarations shared by the translation units of libbPSW_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "bpsw.h"
#include "bpsw_ring.h"

struct bpsw_ctx;

#include <hip/hip_ext.h>

namespace bpsw {

// Events attached to ONE kernel dispatch (hipExtLaunchKernelGGL): start / stop carry the dispatch's own begin and end timestamps,
// which is what a kernel trace reports.  An event recorded before and after a launch brackets the dispatch latency too (65 us per
// launch on the loaded bench).  Both null: a plain launch.
struct KernelEvents {
  hipEvent_t start = nullptr, stop = nullptr;
};
#define BPSW_LAUNCH(ev, kernel, grid, block, lds, stream, ...)                                                          \
  do {                                                                                                                   \
    if ((ev).start || (ev).stop) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, (ev).start, (ev).stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                               \
  } while (0)


// ---- scoring block handed to the kernels by value -------------------------------------------
struct MatRows {
  // row k of the 5x5 matrix packed little-endian: byte c = mat[k*5 + c]
  unsigned long long row[5];
};
MatRows pack_mat(const int8_t mat[25]);

struct ExtScoring {
  MatRows mat;
  int zdrop;
  int zdrop_mode;
  int mat_max;  // max(mat): bounds the scores a task can reach (selects the int16 register path)
  int exact_a;  // match score when the exact-flank shortcut is valid for this matrix (bpsw_extend_core.h), else 0
  int tail_bound;  // 1: stop a call once the rows past the query end cannot change its result (tail_row_bound)
  int certify;     // certify_level(): 1 single-gap certificate for flanks with a deficit below two gap opens, 2 also two opens
  int out_stride;     // int16 units between the result records of consecutive tasks: 10 (the caller's layout) or 16 (one 32-byte
                      // slot per task in the pinned staging buffer: a record never straddles two write sectors)
  uint8_t* side_how;  // optional (bpsw_extend_batch_classify): per task and side, 1 = resolved by an exact shortcut, 2 = DP swept
  // coordinate batches only (wire format 2, include/bpsw.h): the device-resident 2-bit reference the target flanks come from
  const uint8_t* pac;
  long long l_pac;
};
bool certify_enabled();
int certify_level(const int8_t mat[25]);  // 0 off, 1 single-gap certificate, 2 also the two-gap-open extension
bool tail_bound_enabled();  // BPSW_EXT_TAIL=0 disables (A/B runs)
// a > 0 if mat[c][c] == a for the four bases and every other entry is < a; else 0.  BPSW_EXT_EXACT=0 disables.
int exact_match_score(const int8_t mat[25]);
void apply_shortcuts(int mask, const int8_t mat[25], int* exact_a, int* certify, int* tail_bound);  // bpsw_set_ext_shortcuts

// ---- extension (boundary 2) -------------------------------------------------------------------
// Result of the device-side table scan that validates a wire batch before the main launch.
struct ExtPrepass {
  int max_qlen;   // max over tasks of max(leftQlen, rightQlen)
  int max_rlen;   // max over tasks of max(leftRlen, rightRlen)
  int error;      // != 0: malformed table (negative length, sequence outside the buffer, n mismatch)
  int reserved;
};

// Launch the table scan; `d_pre` is a device ExtPrepass that must be zeroed by the caller (stream-ordered).
void launch_ext_prepass(const uint32_t* d_wire, size_t wire_words, int n_tasks, ExtPrepass* d_pre, hipStream_t s);

// LDS bytes one wave needs for tasks up to (qcap, rcap).
size_t ext_lds_per_wave(int qcap, int rcap);

// Launch the extension kernel over a validated batch.
// d_counter: two device ints, the kernel's task queue head and the count of waves that have left; both zero when the launch
// starts (the context zeroes them once, the kernel's last wave puts them back).
// d_task_list (optional): the n_tasks task indices this launch handles (else tasks 0..n_tasks-1).
// d_pre_check (optional): device ExtPrepass written earlier on the same stream; the kernel does nothing when it reports an
// error or lengths beyond (qcap, rcap) -- the asynchronous device entry sizes the launch before anybody has read the scan back.
hipError_t launch_ext_kernel(const uint32_t* d_wire, int n_tasks, int16_t* d_out, const ExtScoring& sc, int qcap,
                             int rcap, int num_cu, int* d_counter, const int* d_task_list, hipStream_t s,
                             const ExtPrepass* d_pre_check = nullptr, bool counter_zeroed = false, KernelEvents kev = KernelEvents(),
                             bool short_kernel = false, int* d_defer = nullptr, int short_qmax = 255,
                             const uint8_t* d_sift_flag = nullptr, const uint4* d_sift_recs = nullptr, int* d_defer_post = nullptr,
                             const int* d_todo_list = nullptr);
// The sift kernel (bpsw_extend_sift.hip): the exact shortcuts of every task of a format-1 batch, one task per lane, in front of
// the short ext_kernel, which reads d_flag[task] (1: record written, skip; 2: d_recs[2 task + side] holds the verdict per side).
// dm = a - (the one mismatch score of the matrix), sift_uniform_dm(); qmax = the longest flank the short build takes.
hipError_t launch_ext_sift_kernel(const uint32_t* d_wire, int n_tasks, int16_t* d_out, const ExtScoring& sc, int dm, int qmax,
                                  uint8_t* d_flag, uint4* d_recs, hipStream_t s, KernelEvents kev = KernelEvents(),
                                  const ExtPrepass* d_pre_check = nullptr, int* d_todo_count = nullptr, int* d_todo_list = nullptr,
                                  int heavy_min = 0);
// (d_todo_list, n_tasks words: the tasks the sift kernel leaves to ext_kernel -- d_todo_count[0] of them from the front, and from
// the back the d_todo_count[1] with the longest sweeps ahead: flanks of heavy_min bases or more in all that no form resolved, or
// a flank the sift does not examine.  launch_ext_kernel with the same list takes its tickets from it, the back first.  The
// counts are the third and fourth word of the queue heads, d_counter[2..3]: zero between launches, ext_kernel's last wave puts
// them back.)
int sift_uniform_dm(const int8_t mat[25], int exact_a);  // > 0 when all twelve base-vs-other-base entries equal exact_a - dm, else 0
// ---- local SW (boundary 1) ---------------------------------------------------------------------
struct SwScoring {
  MatRows mat;
  int a, b, o_del, e_del, o_ins, e_ins;
  int xtra;
};

struct SwJobsDev {  // all device pointers
  int n;
  const int32_t* q_len;
  const int32_t* t_len;
  const int64_t* q_off;
  const int64_t* t_off;
  const uint8_t* q_rev;
  const uint8_t* q_pool;
  const uint8_t* t_pool;          // nullptr: windows are coordinates (t_off) into the 2-bit reference below
  const uint8_t* pac = nullptr;
  long long l_pac = 0;
  // optional: the same table as 32-byte records {q_off, t_off (int64), q_len, t_len, q_rev, 0 (int32)} -- what the packed kernel
  // reads when the table sits in pinned host memory: one 64-byte request per pair of jobs instead of five
  const uint32_t* packed = nullptr;
};

struct SwPrepass {
  int max_qlen;
  int max_tlen;
  int error;
  int reserved;
};
void launch_sw_prepass(const SwJobsDev& jobs, size_t q_pool_bytes, size_t t_pool_bytes, SwPrepass* d_pre, hipStream_t s);
void launch_ref_fetch(const uint8_t* d_pac, long long l_pac, int n, const long long* d_beg, const long long* d_end,
                      uint8_t* d_out_pool, size_t out_pool_bytes, const long long* d_out_off, long long* d_out_len,
                      int* d_error, hipStream_t s);
size_t sw_scratch_bytes_per_wave(int max_tlen);
bool sw_quad_enabled();  // four rescue jobs per wavefront for mates <= 160 bases (BPSW_SW_QUAD=0 disables)
int sw_resident_waves(int num_cu);
// d_pre_check: as for launch_ext_kernel (the launch is sized for max_qlen / max_tlen speculatively).
hipError_t launch_sw_kernel(const SwJobsDev& jobs, const SwScoring& sc, int max_qlen, int max_tlen, int32_t* d_out,
                            uint32_t* d_scratch, int num_cu, hipStream_t s, const SwPrepass* d_pre_check = nullptr, KernelEvents kev = KernelEvents());

// The resident form of the packed rescue kernel behind the per-device submission ring (bpsw_ring.h, bpsw_ring.cpp):
// sw_ring_class: 3 or 5 (the kernel's columns per lane) when a batch with these longest sequences and this scoring can go through
// the ring (*bias_out = the packed kernel's score bias), 0 when it needs a launch of its own.
int sw_ring_class(const SwScoring& sc, int max_qlen, int max_tlen, int* bias_out);
hipError_t launch_swp_resident(int c_class, const RingArgs& A, int blocks, hipStream_t s);
hipError_t launch_ext_resident(const RingArgs& A, int blocks, hipStream_t s);  // bpsw_extend.hip: the extension ring's kernel (RING_CLASS_EXT)

// host side of the ring (bpsw_ring.cpp).  BPSW_RING=0 sends every call through a launch of its own, as before round 5.
bool ring_enabled();
bool ring_usable(int device, int c_class);  // false once that ring has failed: its callers take a launch per batch again
int ring_submit(int device, int c_class, int num_cu, const RingDesc& desc);
int ring_wait(int device, int c_class, const RingDone* done, uint32_t value, double* est_ms);
// ring_wait's "take a launch of your own": the ring of this class failed to LAUNCH an epoch (no resident kernel exists) and the caller's
// descriptor is among those nobody consumed -- nothing will ever run it, and nothing of it has reached the device (internal code, > 0)
constexpr int BPSW_RING_RELAUNCH = 1001;
double ring_ticks_per_ms(int device, int c_class);
// the integrity tripwire (bpsw_ring.cpp): poison `n_records` records of `stride_words` words each at their first word (and at
// `second_word_offset` when not 0) before publishing; afterwards every record must have been overwritten
bool ring_integrity_on();
void ring_poison(uint32_t* first_word, size_t stride_words, size_t n_records, size_t second_word_offset);
int ring_check(const uint32_t* first_word, size_t stride_words, size_t n_records, size_t second_word_offset, const char* what);
void ring_integrity_stats(uint64_t* checked, uint64_t* faults);
void ring_pause(int device);   // close the device's open epochs, wait for their kernels, keep the rings locked ...
void ring_resume(int device);  // ... until here (bpsw_ref_load / unload: a device-wide synchronisation in between)
void ring_get_stats(int device, uint64_t* epochs, uint64_t* submitted, uint64_t* carried, double* epochs_ms = nullptr, uint64_t* epochs_timed = nullptr);

// ---- global alignment + CIGAR -------------------------------------------------
struct GlobalJobsDev {  // all device pointers
  int n;
  int max_cigar;
  const int32_t* q_len;
  const int32_t* t_len;
  const int32_t* w;
  const int64_t* q_off;
  const int64_t* t_off;
  const uint8_t* q_pool;
  const uint8_t* t_pool;
};
struct GlobalPrepass {
  int max_qlen;
  int error;
  unsigned long long max_z;  // max over jobs of nCol * tLen (bytes of backtrack matrix)
};
void launch_global_prepass(const GlobalJobsDev& jobs, size_t q_pool_bytes, size_t t_pool_bytes, GlobalPrepass* d_pre, hipStream_t s);
int global_resident_waves(int num_cu, int qcap);
hipError_t launch_global_kernel(const GlobalJobsDev& jobs, const SwScoring& sc, int max_qlen, size_t z_per_wave,
                                int32_t* d_score, int32_t* d_ncigar, uint32_t* d_cigar, uint8_t* d_z, int num_cu,
                                hipStream_t s);

// ---- memRegToAln on the device (bpsw_reg2aln.hip) ----------------------------------------------------
struct Reg2AlnDev {  // all device pointers
  int n;
  int max_cigar, max_md;
  int flavour;  // BPSW_TAIL_SCALA / BPSW_TAIL_C: the band rule of bwaGenCigar2
  int opt_w, a;
  const int32_t* read_len;
  const long long* read_off;
  const uint8_t* read_pool;
  const bpsw_alnreg_t* regs;
  const uint8_t* pac;
  long long l_pac;
  int n_seqs;
  const long long* ann_off;
  const int32_t* ann_len;
};
struct Reg2AlnOut {  // what the kernel computes of a mem_aln_t; flag, mapq, score, sub are the host's (bpsw_tail.cpp)
  long long pos;
  int32_t rid, is_rev, NM, n_cigar, md_len, status, gscore, pad_;
};
size_t reg2aln_lds_per_wave(int qcap, int rcap, int md_cap);
int reg2aln_resident_waves(int num_cu, int qcap, int rcap, int md_cap);
hipError_t launch_reg2aln_kernel(const Reg2AlnDev& J, const SwScoring& sc, int qcap, int rcap, int md_cap, size_t z_per_wave,
                                 Reg2AlnOut* d_out, uint32_t* d_cigar, uint8_t* d_md, uint8_t* d_z, int num_cu, hipStream_t s);

// ---- memChainToAlnBatched on the device ----------------------------------------------
struct ChainParams {
  MatRows mat;
  int mat_max, a, o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, w, zdrop, zmode;
  int exact_a;  // see ExtScoring
  int tail_bound, certify;
};
struct ChainBatchDev {  // all device pointers
  int n_reads;
  const int32_t* read_len;
  const long long* read_off;
  const uint8_t* read_pool;
  const int32_t* chain_cnt;
  const int32_t* chain_base;   // first chain of read r
  const int32_t* seed_cnt;     // per chain
  const long long* seed_base;  // first seed of chain c
  const long long* seed_rbeg;
  const int32_t* seed_qbeg;
  const int32_t* seed_len;
  const long long* reg_base;   // first output slot of read r (= seeds before it)
  const uint8_t* pac;
  long long l_pac;
};
int chain2aln_resident_waves(int num_cu);
hipError_t launch_chain2aln_kernel(const ChainBatchDev& B, const ChainParams& P, bpsw_alnreg_t* d_out_regs, int32_t* d_out_cnt,
                                   int32_t* d_srt_scratch, int srt_per_wave, int num_cu, int* d_counter, hipStream_t s);
// memSortAndDedup on a host vector (bpsw_rescue.cpp): mode BPSW_RESCUE_C or BPSW_RESCUE_SCALA; returns the new size
int sort_dedup_regs(std::vector<bpsw_alnreg_t>& v, float mask_level_redun, int mode);
void rescue_scratch_free(void* p);

// ---- error text -----------------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

// ---- context ----------------------------------------------------------------------------------------
struct DeviceBuffer {
  void* ptr = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes);  // grow-only
  void release();
};
struct PinnedBuffer {
  void* ptr = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes);
  void release();
};

// The 2-bit reference (bpsw_ref_load) is shared by every context of a device: the JNI shim keeps one context per
// Spark task thread, and 20 copies of a 0.8 GB genome would be pointless.
// Readers (every call that launches kernels over the resident reference) hold the gate for the duration of the call; a load /
// unload holds it exclusively while it frees or reallocates the buffers, so that no thread can sit between taking the device
// pointer and launching while another context replaces the reference (one context per task thread share one reference).
// Writers are not starved: a queued writer (bpsw_ref_load / unload / bns_load) keeps NEW readers out, so that a stream of
// overlapping calls from 20-32 task threads cannot hold `readers` above zero for ever; a NESTED read of a thread that already
// holds the gate (a call that takes a snapshot inside another) is still admitted, or it would deadlock against that writer.
// The nesting depth is counted PER GATE and per thread (a thread that reads device A's reference is not "nested" on device B's
// gate), in a small thread-local table; a hold must be released by the thread that took it (RefHold asserts it).
struct RefGate;
struct RefDepthSlot { const RefGate* gate; int depth; };
inline thread_local RefDepthSlot t_ref_depth[8] = {};
inline int* ref_depth_of(const RefGate* g) {
  RefDepthSlot* free_slot = nullptr;
  for (RefDepthSlot& s : t_ref_depth) {
    if (s.gate == g) return &s.depth;
    if (!free_slot && s.depth == 0) free_slot = &s;
  }
  if (!free_slot) abort();  // a thread inside more than eight devices' gates at once: not something this library does
  free_slot->gate = g;
  free_slot->depth = 0;
  return &free_slot->depth;
}
struct RefGate {
  std::mutex m;
  std::condition_variable cv;
  int readers = 0, writers_waiting = 0;
  bool writing = false;
  void read_lock() {
    int* depth = ref_depth_of(this);
    std::unique_lock<std::mutex> lk(m);
    if (*depth == 0) cv.wait(lk, [&] { return !writing && writers_waiting == 0; });
    else cv.wait(lk, [&] { return !writing; });
    ++readers;
    ++*depth;
  }
  void read_unlock() {
    int* depth = ref_depth_of(this);
    if (*depth <= 0) abort();  // released on another thread than the one that took it: the writer preference would silently go
    std::lock_guard<std::mutex> lk(m);
    --*depth;
    if (--readers == 0) cv.notify_all();
  }
  void write_lock() {
    std::unique_lock<std::mutex> lk(m);
    ++writers_waiting;
    cv.wait(lk, [&] { return !writing && readers == 0; });
    --writers_waiting;
    writing = true;
  }
  void write_unlock() { { std::lock_guard<std::mutex> lk(m); writing = false; } cv.notify_all(); }
};
struct RefHold {  // RAII read side of the gate; movable within the thread that took it
  RefGate* g = nullptr;
  RefHold() = default;
  explicit RefHold(RefGate* gate) : g(gate) { if (g) g->read_lock(); }
  RefHold(RefHold&& o) noexcept : g(o.g) { o.g = nullptr; }
  RefHold& operator=(RefHold&& o) noexcept { if (this != &o) { release(); g = o.g; o.g = nullptr; } return *this; }
  RefHold(const RefHold&) = delete;
  RefHold& operator=(const RefHold&) = delete;
  void release() { if (g) { g->read_unlock(); g = nullptr; } }
  ~RefHold() { release(); }
};
struct RefWriteHold {
  RefGate* g;
  explicit RefWriteHold(RefGate* gate) : g(gate) { g->write_lock(); }
  ~RefWriteHold() { g->write_unlock(); }
  RefWriteHold(const RefWriteHold&) = delete;
  RefWriteHold& operator=(const RefWriteHold&) = delete;
};

struct DeviceRef {
  RefGate gate;
  std::mutex mu;
  DeviceBuffer buf;
  long long l_pac = 0;
  // the contig table (bntann1_t offset/len, bpsw_bns_load): device copy for the kernels, host copy for the tail
  DeviceBuffer ann;  // n_seqs x int64 offsets, then n_seqs x int32 lengths
  std::vector<long long> ann_off;
  std::vector<int32_t> ann_len;
  std::vector<std::string> ann_name;
};
DeviceRef& device_ref(int device);

// Device streams for the blocking host-buffer entry points.  A calling thread needs a stream only for the device phase of its
// call (H2D, kernel, D2H), not while it stages bytes or replays bookkeeping on the host; and the GPU has a fixed number of
// hardware queues (GPU_MAX_HW_QUEUES): beyond about 22 busy queues throughput collapses (measured on the bench: 16 queues
// 131, 18: 136, 20: 138, 22: 141 M reads/s, 24: 59, 32: 34).  So the device phases of all contexts of a device
// share a pool of BPSW_STREAM_POOL streams (default 20; 0 = every context uses its own stream): an executor may run more
// task threads than the device has queues, and their host phases overlap the others' device phases.
struct StreamLease {
  int device;
  hipStream_t s;
  bool pooled;
  int slot = -1;   // index of the pooled stream
  double wait_ms;  // time spent waiting for a free stream
  explicit StreamLease(bpsw_ctx* c);
  ~StreamLease();
  StreamLease(const StreamLease&) = delete;
  StreamLease& operator=(const StreamLease&) = delete;
};
// the device's lane for bulk copies that are made before a call takes a stream (extend_batch_impl): one at a time
struct CopyLane {
  std::mutex mu;
  hipStream_t s = nullptr;
  hipEvent_t ev = nullptr;
};
CopyLane& copy_lane(int device);
double wall_ms();
void ext_call_mark(int device);      // an extension call of the device begins / ends now (bpsw_runtime.cpp)
double ext_call_age_ms(int device);  // ms since the last such mark (huge: never)
double stat_ms();  // wall_ms, or the thread CPU clock with BPSW_STATS_CLOCK=cpu (bpsw_runtime.cpp)
// rescue launches (sw_stage_run) between their launch and the end of their wait, per device: the extension path shapes its bulk copies
// by it (extend_batch_impl)
int sw_launches_in_flight(int device);
void sw_launch_in_flight(int device, int delta);
double wait_est_update(double est, double took_ms, int polls, bool napped);  // the next estimate of a kind of wait
bool wait_naps(double est_ms);                                       // whether wait_nap sleeps at all for this estimate
void wait_nap(double est_ms);                                        // the sleep before the first look (BPSW_WAIT_MODE, bpsw_runtime.cpp)
void wait_poll_pause(int polls, double waited_ms, double est_ms);    // between two looks
hipError_t wait_event(bpsw_ctx* c, hipEvent_t ev, int kind);  // kind 0: extension call, 1: SW call (separate duration estimates)
int zerocopy_mask();  // BPSW_ZEROCOPY, see bpsw_runtime.cpp
bool spin_wait();  // BPSW_SPIN_WAIT=1: busy-wait for the device instead of sleeping on a blocking event
// snapshot of the reference loaded on c's device (l_pac == 0: none).  The returned hold keeps load / unload out until it is
// destroyed: keep it alive until the kernels that use `pac` have been waited for (the asynchronous device entries, which return
// before that, rely on the hipDeviceSynchronize a load / unload performs after it has got the gate).
[[nodiscard]] RefHold ref_snapshot(const bpsw_ctx* c, const uint8_t** pac, long long* l_pac);

// Asynchronous device entries (bpsw_extend_batch_device, bpsw_swalign2_batch_device) leave a launch whose table scan has not
// been read back; every entry point that reuses the context's scan buffers or stream resolves it first.  Caller holds c->mu
// and has set the device.  Scan buffers: ExtPrepass at d_pre+0 (bin counts +64, queue heads +128), SwPrepass at d_pre+256.
int finish_pending_ext(bpsw_ctx* c);
int finish_pending_sw(bpsw_ctx* c);
inline int finish_pending(bpsw_ctx* c) {
  const int a = finish_pending_ext(c), b = finish_pending_sw(c);
  return a != BPSW_OK ? a : b;
}

// runs SWAlign2 jobs whose arrays live in host memory; used by bpsw_swalign2_batch.
// Caller holds ctx->mu and has set the device.
int run_sw_jobs_host(bpsw_ctx* c, const bpsw_opt_t* opt, const bpsw_sw_jobs_t* jobs, int32_t* out);
// The same in two steps for callers that build the job table in place (the rescue layer packs straight into the pinned
// staging block): sw_stage_begin lays the block out for n jobs and the two pool sizes, the caller fills
// base + o_* (q_len/t_len int32, q_off/t_off int64 relative to the pools, q_rev bytes, the pools), sw_stage_run launches.
// The caller is responsible for what run_sw_jobs_host validates (1 <= q_len, sequences inside their pools / the reference).
struct SwStage {
  int n;
  size_t o_qlen, o_tlen, o_qoff, o_toff, o_qrev, o_qpool, o_tpool, o_packed, total, q_pool_bytes, t_pool_bytes;
  uint8_t* base;
};
int sw_stage_begin(bpsw_ctx* c, int n, size_t q_pool_bytes, size_t t_pool_bytes, SwStage* st);
int sw_stage_run(bpsw_ctx* c, const bpsw_opt_t* opt, int xtra, const SwStage& st, int mq, int mt, bool pac_mode, const int32_t** results);

}  // namespace bpsw

struct bpsw_ctx {
  int device = 0;
  int num_cu = 256;
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  std::mutex mu;
  bpsw::ExtScoring ext_sc;
  int8_t ext_mat[25];
  // persistent arenas (grow-only; no hipMalloc on the steady-state path)
  bpsw::DeviceBuffer d_wire, d_out, d_pre, d_sw_in, d_sw_out, d_sw_scratch, d_gl_z, d_ext_lists;
  size_t staged_bytes = 0;    // what the last bpsw_extend_stage was asked for (0: nothing staged / already committed)
  bpsw::DeviceBuffer d_sift;  // the sift kernel's verdicts (bpsw_extend_sift.hip): [flag byte per task | two 16-byte records per task]
  // asynchronous device entries: a launch whose table scan has not been read back yet (resolved by finish_pending)
  struct PendingExt { bool active = false; const void* d_wire = nullptr; size_t wire_bytes = 0; int n_tasks = 0; void* d_out = nullptr;
                      hipStream_t s = nullptr; int qcap = 0, rcap = 0; } pend_ext;
  int ext_geom_q = 0, ext_geom_r = 0;  // longest query / target side of the verified extension launches on this context (speculation)
  struct PendingSw { bool active = false; bpsw_sw_jobs_t jobs; bpsw_opt_t opt; void* d_out = nullptr; hipStream_t s = nullptr;
                     int cap_qlen = 0, cap_tlen = 0;
                     bool verified = false;  // launched after the scan was read back: nothing to resolve, only to wait for
  } pend_sw;
  int sw_geom_qlen = 0, sw_geom_tlen = 0;  // geometry of the last verified SW launch on this context (speculation for the next)
  float last_tail_ms = 0.f;
  int last_tail_jobs = 0, last_tail_resubmitted = 0;
  double tail_host_ms[3] = {0., 0., 0.};  // plan, device round trip (staging + copies + kernel), emit
  bool have_tail_ev = false;

  bpsw::PinnedBuffer h_stage_in, h_stage_out, h_pre;
  int shortcut_mask = 63;  // bpsw_set_ext_shortcuts
  std::vector<int> ext_long_tasks, ext_mid_tasks;  // scratch of bpsw_extend_batch: the tasks of the current batch that go to the full kernel / have a flank of 128-255 bases
  double wait_est_ms[6] = {0., 0., 0., 0., 0., 0.};  // wait_event / ring_wait: running average of the device-phase waits ([0] extension launches, [1] SW launches, [2] ring copy-in, [3] spare, [4] extension ring, [5] SW ring: a thread that alternates 1 ms launched batches with 0.085 ms ring batches must not nap through the short ones on the long ones' estimate)
  bool ring_abandoned = false;       // a ring batch of this context ran into the watchdog: a late unit may still write into the pinned blocks, so the context refuses further calls and bpsw_destroy leaks them
  uint32_t ring_seq = 0;             // completion values of this context's ring submissions (RingDone at h_pre + 448)
  void* rescue_scratch = nullptr;  // bpsw_rescue.cpp: vectors reused across bpsw_matesw_group calls (freed by rescue_scratch_free)
  bpsw_stats_t stats;
  float last_ext_ms = 0.f, last_sw_ms = 0.f;
  bool have_ext_ev = false, have_sw_ev = false;
};
