// bpsw_ring.h -- the per-device submission ring behind the blocking entry points (round 5).
//
// What it replaces.  A blocking call used to be "take one of twenty pooled streams, launch a kernel over this call's jobs,
// wait for the stream": every call paid the command processors' turn-around on twenty busy queues (0.15-0.2 ms between the
// launch call's return and the first wave), the 60-65 us by which a launch's first waves are spread over the eight XCDs, and
// the tail of its own launch -- a fifth of all thread time of the bench step -- and a call of ten pairs paid
// the same as one of four thousand.  The reference has no such structure to mirror: its boundary is one synchronous C call
// per group (native/jni_mate_sw.c:534) or one shm hop per batch (jni_fpga/sw_extend_fpga.c:116-193).
//
// What it is.  Per device and kernel class ONE resident kernel (an "epoch") whose worker wavefronts pull units of work --
// for the rescue kernel a pair of SW jobs -- from a ring of descriptors; task threads append a descriptor (any number of
// units) to the ring in pinned host memory, a poller wavefront mirrors new descriptors into device memory, workers take
// units by an atomic counter per descriptor, and the worker that finishes a descriptor's last unit writes the caller's
// completion record straight into the caller's pinned block.  Calls of different threads land in the same wave population:
// nothing is launched per call, a small call costs its own units only.
//
// Exit protocol (no wave may be left behind; hipDeviceSynchronize must return):
//   * the poller closes the epoch when it has seen no new descriptor for `idle_ticks` while every mirrored descriptor has
//     been handed out, when the host asks for it (close_req: bpsw_ref_load / unload, process exit), or when the epoch's
//     ring is used up (descriptor slots are never reused within an epoch, so no worker can meet a recycled slot).  Closing
//     is a two-phase handshake over PCIe so that a descriptor published concurrently is either served or reported as not
//     consumed (state = CLOSED | consumed count), never lost: device W(state = CLOSING) ; fence ; R(tail) against host
//     W(tail) ; fence ; R(state).  A host thread that finds the epoch closed starts the next one (stream-ordered behind the
//     old kernel) and carries the unconsumed descriptors over.
//   * workers leave when the poller has set `quit` and nothing is left to hand out, or -- a safety net that needs no
//     cooperation -- after `worker_idle_ticks` without work.  Waiting callers have a watchdog (BPSW_RING_TIMEOUT_MS) that turns
//     a ring that makes no progress into an error code; nothing ever re-executes the process.
#pragma once
#include <stdint.h>

#include <atomic>

namespace bpsw {

constexpr uint32_t RING_DESC_WORDS = 64;  // 256 bytes: one coalesced wave load

// phases of RingHostCtl::state (low byte); bits 8..39 = descriptors consumed, bits 40..63 = epoch
constexpr uint64_t RING_OPEN = 1, RING_CLOSING = 2, RING_CLOSED = 3;
inline uint64_t ring_state(uint32_t epoch, uint32_t consumed, uint64_t phase) {
  return ((uint64_t)(epoch & 0xffffffu) << 40) | ((uint64_t)consumed << 8) | phase;
}
inline uint32_t ring_state_epoch(uint64_t s) { return (uint32_t)(s >> 40); }
inline uint32_t ring_state_consumed(uint64_t s) { return (uint32_t)((s >> 8) & 0xffffffffu); }
inline uint64_t ring_state_phase(uint64_t s) { return s & 0xffu; }

// Control block in pinned host memory: one line the host writes, one the device writes.  The words both sides touch are std::atomic
// on the host (lock-free, same size and layout as the plain word: the device code addresses them through casts, bpsw_ring_dev.h) --
// round 5 had `volatile` fields and free-standing fences, which is neither a data-race-free C++ program nor something a thread
// sanitizer can follow (tests/ring_host builds this file's host half under -fsanitize=thread against a C++ thread that plays the kernel).
struct RingHostCtl {
  std::atomic<uint32_t> tail;       // host: descriptors published in the current epoch
  std::atomic<uint32_t> close_req;  // host: != 0 -> close the epoch of this number as soon as possible
  uint32_t pad0[30];
  std::atomic<uint64_t> state;      // device (the host initialises it to OPEN before the launch)
  std::atomic<uint64_t> heartbeat;  // device: the poller's clock at its last pass (diagnostics)
  std::atomic<uint32_t> workers_seen;  // device: worker wavefronts that took at least one unit in this epoch (diagnostics)
  std::atomic<uint32_t> close_reason;  // device: why the epoch closed -- 1 asked by the host, 2 ring used up, 3 idle, 4 no progress (diagnostics)
  std::atomic<uint64_t> diag_claim_ticks, diag_unit_ticks, diag_units;  // copies of RingDevCtl's at the epoch's close
  uint32_t pad1[20];
};
static_assert(sizeof(RingHostCtl) == 256, "ring control block layout");
static_assert(sizeof(std::atomic<uint32_t>) == 4 && sizeof(std::atomic<uint64_t>) == 8, "the device side addresses these words through casts");

// Control block in device memory.  The three words a waiting worker looks at share sixteen bytes: ONE load per look.
struct RingDevCtl {
  uint32_t tail;   // poller -> workers: descriptors mirrored into d_desc
  uint32_t cur;    // workers: first descriptor that may still have units to hand out
  uint32_t quit;   // poller: the epoch is closed
  uint32_t pad0[29];
  uint32_t workers; uint32_t pad3[31];
  // diagnostics builds (-DBPSW_RING_DIAG): ticks between a descriptor's publication and each of its units being taken; ticks a unit took
  unsigned long long diag_claim_ticks, diag_unit_ticks, diag_units, diag_pad[13];
};
struct RingCtr {  // per descriptor, device memory, zero at epoch start
  uint32_t next;     // units handed out
  uint32_t n_units;  // the descriptor's unit count (the poller's copy: a worker sees "handed out" with one load, without an atomic)
  uint32_t done;     // units finished
  uint32_t pad;
  uint64_t t0;       // device clock when unit 0 was taken
  uint64_t t_pub;    // device clock when the poller published the descriptor
};
static_assert(sizeof(RingCtr) == 32, "ring counters");

// A descriptor: 64 words.  Words 0-7 are the ring's, the rest is the kernel class's payload.
struct RingDescHead {
  uint32_t n_units;     // > 0
  uint32_t done_value;  // what the completion record's first word becomes
  uint64_t done_ptr;    // pinned host address of the completion record: {u32 value, u32 0, u64 t_first_unit, u64 t_done}
  uint32_t reserved[4];
};
static_assert(sizeof(RingDescHead) == 32, "ring descriptor head");
struct RingDesc {
  uint32_t w[RING_DESC_WORDS];
};
struct RingDone {  // the completion record a caller waits on (pinned host memory, 32 bytes)
  std::atomic<uint32_t> value;
  uint32_t pad;
  std::atomic<uint64_t> t_first, t_done;
  uint64_t pad2;
};
static_assert(sizeof(RingDone) == 32, "completion record layout");

// what a caller writes over every record of its result block before it publishes a descriptor (ring_poison / ring_check, bpsw_ring.cpp):
// no kernel writes it -- a rescue job's score and an extension record's upper half of the width word are never negative
constexpr uint32_t RING_POISON = 0x80005a5au;

// payload of the rescue kernel's descriptors (words 8..): what swp_kernel takes as arguments
struct SwRingPayload {
  uint64_t packed, q_pool, t_pool, pac;  // device-visible addresses (pinned host memory for the first three)
  int64_t l_pac;
  uint64_t out;                          // 7 int32 per job
  int32_t n_jobs, bias;
  uint64_t mat_row[5];
  int32_t a, b, o_del, e_del, o_ins, e_ins, xtra, pad;
};
static_assert(sizeof(RingDescHead) + sizeof(SwRingPayload) <= sizeof(RingDesc), "rescue payload fits a descriptor");

// payload of the EXTENSION kernel's descriptors (words 8..): a wire batch (either format) in device-visible memory whose flanks all have at
// most 255 bases (nothing for the full kernel) and that is too small for the sift kernel to pay -- the calls a launch costs most
// (-FPGASWExtThreshold 64, the later rounds of memChainToAlnBatched: worker1/MemChainToAlignBatched.scala:471-615).  A unit is
// `per_unit` consecutive tasks.
constexpr int RING_CLASS_EXT = 8;
constexpr int EXT_RING_QCAP = 256, EXT_RING_RCAP = 640;  // LDS geometry of the resident kernel: flanks of <= 255 query, <= 640 target bases
struct ExtRingPayload {
  uint64_t wire, out;  // the batch (device-visible memory) and its 10-int16 records (the caller's pinned block)
  int32_t n_tasks, per_unit, out_stride, zdrop, zdrop_mode, mat_max, exact_a, tail_bound, certify;
  int32_t coord;       // != 0: a coordinate batch (wire format 2): the target flanks come from the reference below
  uint64_t mat_row[5];
  uint64_t pac;        // the device-resident 2-bit reference (bpsw_ref_load) and its length
  int64_t l_pac;
};
static_assert(sizeof(RingDescHead) + sizeof(ExtRingPayload) <= sizeof(RingDesc), "extension payload fits a descriptor");

// what an epoch's kernel is launched with
struct RingArgs {
  RingHostCtl* H;
  const RingDesc* h_desc;  // pinned host memory, `capacity` descriptors
  RingDevCtl* D;
  RingDesc* d_desc;        // device memory mirror
  RingCtr* ctr;
  uint32_t epoch, capacity;
  unsigned long long idle_ticks;         // poller: close after this long without a new descriptor and nothing left to hand out
  unsigned long long worker_idle_ticks;  // worker: leave after this long without work, whatever the poller does
  unsigned long long sleep_ticks_us;     // device clock ticks per microsecond (hipDeviceAttributeWallClockRate / 1000)
  uint32_t nap_rounds_max;               // a waiting worker's longest nap, in rounds of ~3.4 us (16: ~55 us)
  uint32_t pad;
};

}  // namespace bpsw
