#!/usr/bin/env python3
"""bench.py -- reads aligned/sec through the MI355X batched Smith-Waterman path, measured the way SURVEY.md 8(d) defines it:
wall clock of the HOST-BUFFER C ABI calls the two JNI symbols make, H2D / D2H and the whole boundary-1 host layer included.

One step = the hot path over 16 777 216 synthetic read pairs: sixteen passes over a set of 1 048 576 DISTINCT pairs (2.3 GB of
host inputs, far beyond any cache, so a pass never finds its data warm; sixteen so that the driver's 20 steps time more than
three seconds and its GPU-busy samples see the device at work).  The feeder threads go round the passes without a barrier: each
takes the next call when it has finished one, as the task threads of an executor do.  Default workload: BASELINE.json configs[2], 2x150 bp, 1 % substitutions, 0.1 % indels, 10 % of the pairs need mate
rescue.  Per pass:
  * boundary 2: one bpsw_extend_batch call per wire batch of 32 768 reads (the reference's -bSWExtSize 32768, run_test.sh:7;
    the call behind MemChainToAlignBatched.scala:175-176), host wire bytes in, host int16 results out;
  * boundary 1: one bpsw_matesw_group call per group of 4 096 pairs (the call behind native/jni_mate_sw.c:534 /
    MemSamPe.scala:2091-2092): speculation, H2D, the SW kernel, D2H, the sequential replay and sort/dedup, host arrays out.
The calls are made by T = 32 native host threads (csrc/bpsw_feeder.cpp), one context each, the way T Spark task threads of one
executor call the JNI symbols; inputs are generated up front and live in host memory, outputs land in host memory, and a
sample of the TIMED outputs is compared with the oracle after the timed region ("verified").
`breakdown.device_resident` keeps the round-1 figure (inputs already in HBM, asynchronous device entries, kernels only).

N>1: one process per GPU (torch.distributed over RCCL for the barrier and the max-over-ranks only; the path has no
data-path collective), each rank streams its own shard of pairs ("weak" scaling: Spark partition -> device).

--config {2,3,5} (or BENCH_CONFIG) selects the SURVEY.md 8(d) workload: 3 (default) as above, 2 = single-end 150 bp
(extension only), 5 = 2x250 bp at 8 % / 2 % error with a 1 % tail at 20 % (the wide-band regime).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import traceback
import os
import sys
import time

# The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4), round robin; streams that
# share a queue run one after the other.  Every feeder thread owns a context with its own stream.  Must be set before the
# runtime initialises; an executor that runs several task threads wants the same.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "cloud-scale-bwamem_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

READS_PER_EXT_BATCH = 32768       # reference: -bSWExtSize 32768 (run_test.sh:7); idx travels as int16
PAIRS_PER_GROUP = int(os.environ.get("BENCH_GROUP_PAIRS", "4096"))   # boundary-1 group size (4096); the override is a diagnostic
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
INT_VALU_PEAK_OPS = 256 * 4 * 32 * 2.4e9   # MI355X_MICROARCH.md ("Terms", "Wave scheduling", cycles table): 256 CUs x 4 SIMD-32 x 2.4 GHz --
                                            # a wave64 VALU instruction issues over 2 cycles, 7.86e13 int32 lane-ops/s
PROFILE_TAG = "r06"              # the committed rocprofv3 summaries this round's line is cross-checked against (profiles/r06_*)
PROTOCOL = "r03b"                # what a step IS: changes whenever values stop being comparable with earlier rounds (see `protocol` in the line)

# SURVEY.md 8(d) workloads, keyed by the survey's config number
WORKLOADS = {
    3: dict(label="configs[2]: 10M-pair-shaped stream of pair-end 2x150bp synthetic reads (1% sub, 0.1% indel) vs a chr21-sized "
                  "coordinate space, batched seed extension + batched pair-end SW rescue (10% of pairs)",
            metric="pair-end 2x150bp reads aligned/sec", read_len=150, sub=0.01, indel=0.001, tail_frac=0.0, tail_sub=0.2,
            tail_indel=0.02, p_resc=0.10, mate_sub=0.02, mate_indel=0.002, paired=True, ext_batches=64, groups=256, passes=16),
    2: dict(label="configs[1]: single-end 150bp synthetic reads (1% sub, 0.1% indel), HIP seed extension only",
            metric="single-end 150bp reads aligned/sec", read_len=150, sub=0.01, indel=0.001, tail_frac=0.0, tail_sub=0.2,
            tail_indel=0.02, p_resc=0.0, mate_sub=0.0, mate_indel=0.0, paired=False, ext_batches=32, groups=0, passes=32),
    5: dict(label="configs[4]: pair-end 2x250bp high-error synthetic reads (8% sub, 2% indel, 1% of reads at 20%/2%), "
                  "wide-band extension + pair-end SW rescue (25% of pairs)",
            metric="pair-end 2x250bp reads aligned/sec", read_len=250, sub=0.08, indel=0.02, tail_frac=0.01, tail_sub=0.20,
            tail_indel=0.02, p_resc=0.25, mate_sub=0.08, mate_indel=0.02, paired=True, ext_batches=32, groups=128, passes=2),
}


def ext_seed(cfg_no, rank, b):
    from bpsw_hip import synth
    return synth.CONFIG_SEED_BASE + cfg_no + 1000 * rank + 7 * b


def grp_seed(cfg_no, rank, g):
    from bpsw_hip import synth
    return synth.CONFIG_SEED_BASE + cfg_no + 1000 * rank + 500_000 + 13 * g


def make_ext_soa(W, cfg_no, rank, b):
    from bpsw_hip import synth
    return synth.ext_tasks(READS_PER_EXT_BATCH, read_len=W["read_len"], sub_rate=W["sub"], indel_rate=W["indel"], n_rate=0.001,
                           tail_frac=W["tail_frac"], tail_sub_rate=W["tail_sub"], tail_indel_rate=W["tail_indel"],
                           seed=ext_seed(cfg_no, rank, b))


def make_group(W, cfg_no, rank, g):
    from bpsw_hip import synth
    return synth.rescue_group_fast(PAIRS_PER_GROUP, read_len=W["read_len"], seed=grp_seed(cfg_no, rank, g), p_resc=W["p_resc"],
                                   sub_rate=W["mate_sub"], indel_rate=W["mate_indel"])


def build_inputs(W, cfg_no, rank, workers):
    """every wire batch and rescue group of one step, distinct, in host memory (numpy); generation is threaded (the C
    generators release the GIL)"""
    from concurrent.futures import ThreadPoolExecutor
    import bpsw_hip

    def one_wire(b):
        soa = make_ext_soa(W, cfg_no, rank, b)
        return bpsw_hip.wire_pack(soa), soa.n

    with ThreadPoolExecutor(workers) as ex:
        wn = list(ex.map(one_wire, range(W["ext_batches"])))
        groups = list(ex.map(lambda g: make_group(W, cfg_no, rank, g), range(W["groups"] * 4096 // PAIRS_PER_GROUP)))
    return [w for w, _ in wn], [n for _, n in wn], groups


def gpu_numa_cpus(dev_index):
    """CPUs of the NUMA node the GPU hangs off (sysfs), restricted to this process's affinity; [] when unknown"""
    try:
        import torch
        bus = torch.cuda.get_device_properties(dev_index).pci_bus_id if hasattr(torch.cuda.get_device_properties(dev_index), "pci_bus_id") else None
        dom = getattr(torch.cuda.get_device_properties(dev_index), "pci_domain_id", 0)
        devn = getattr(torch.cuda.get_device_properties(dev_index), "pci_device_id", 0)
        if bus is None:
            return [], None
        path = f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{devn:02x}.0/numa_node"
        node = int(open(path).read().strip())
        if node < 0:
            return [], node
        cpus = []
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.extend(range(int(a), int(b or a) + 1))
        allowed = os.sched_getaffinity(0)
        return [c for c in cpus if c in allowed], node
    except Exception:
        return [], None


def reduce_over_ranks(elapsed, device):
    """the N>1 protocol of the contract: MAX of the ranks' elapsed times, and how many ranks took part (SUM of ones);
    `device` is where the reduction tensors live (the rank's GPU under RCCL, "cpu" under gloo)"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return float(t.item()), int(one.item())


def whole_job_rate(reads_per_step_per_rank, steps, world, elapsed_max):
    """`value`: the reads ALL ranks aligned in the timed region over the slowest rank's time"""
    return reads_per_step_per_rank * steps * world / elapsed_max


def load_committed_json(name):
    """a committed profiles/<name> as a dict; {} only when the file is not there (a fresh workload without counters) -- a file that is
    there and does not parse is an error of the repository and stops the bench"""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except FileNotFoundError:
        return {}


def trace_figures(csv_path, sw_launch_kernel="swp_kernel"):
    """Per-kernel averages and shares of summed kernel time from a committed `rocprofv3 --kernel-trace --stats` summary of the bench command.

    Returns (trace_avg, trace_share):
      trace_avg["extend"]   = a format-1 extension CALL's launches summed (sift kernel + short kernel + the occasional full kernel: the
                              `false` instantiations, back to back on the call's stream, as the library's events see them) / short-kernel launches, ms
      trace_avg["swalign2_resident"], ["swalign2"] = AverageNs of the resident rescue kernel / of the launched one, ms
      trace_share[k]        = k's share of the summed kernel time of the trace
    ({}, {}) only when the file does not exist; anything else wrong with it raises."""
    import csv
    trace_avg, trace_share = {}, {}
    try:
        f = open(csv_path, newline="")
    except FileNotFoundError:
        return {}, {}
    with f:
        ext_total_ns, ext_calls = 0.0, 0
        for r in csv.DictReader(f):
            nm = r["Name"]
            if "ext_sift_kernel<false>" in nm or "ext_kernel<false" in nm:
                ext_total_ns += float(r["TotalDurationNs"])
                trace_share["extend"] = trace_share.get("extend", 0.0) + float(r["Percentage"]) * 1e-2
                if "ext_kernel<false, 1>" in nm:   # the short kernel
                    ext_calls += int(r["Calls"])
            elif "swp_resident_kernel" in nm and "swalign2_resident" not in trace_avg:
                trace_avg["swalign2_resident"] = round(float(r["AverageNs"]) * 1e-6, 4)
                trace_avg["swalign2_resident_share_of_kernel_time"] = round(float(r["Percentage"]) * 1e-2, 4)
                trace_share["swalign2_resident"] = float(r["Percentage"]) * 1e-2
            elif sw_launch_kernel in nm and "swalign2" not in trace_avg:
                trace_avg["swalign2"] = round(float(r["AverageNs"]) * 1e-6, 4)
                trace_share["swalign2"] = float(r["Percentage"]) * 1e-2
        if ext_calls:
            trace_avg["extend"] = round(ext_total_ns / ext_calls * 1e-6, 4)
    return trace_avg, trace_share


def pick_dominant(trace_share, ext_instr, sw_instr, ext_kernel_ms, sw_kernel_ms, tag):
    """(dominant, dominant_by): the kernel family with the largest share of summed kernel time in the committed kernel trace of this
    command -- the extension call's kernels together against the rescue path's; without a trace, the one that issues the most
    wave-instructions in a step (launches x the committed per-launch counts); without either, the larger sum of launch durations."""
    if trace_share.get("extend") is not None and (trace_share.get("swalign2_resident") is not None or trace_share.get("swalign2") is not None):
        sw_share = (trace_share.get("swalign2_resident") or 0.0) + (trace_share.get("swalign2") or 0.0)
        return ("extend" if trace_share["extend"] >= sw_share else "swalign2"), \
            f"share of summed kernel time in profiles/{tag}_kernel_stats_bench.csv (extension kernels {trace_share['extend']:.3f}, rescue kernels {sw_share:.3f})"
    if ext_instr is not None:
        return ("extend" if ext_instr >= sw_instr else "swalign2"), "issued wave-instructions (profiles/pmc_issue.json x launches)"
    return ("extend" if ext_kernel_ms >= sw_kernel_ms else "swalign2"), "summed launch durations (no trace or counters committed for this workload)"


def cpu_quota():
    """CPUs worth of time the cgroup gives this process (cpu.max: quota / period), or None when there is no readable quota"""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                return None if txt[0] == "max" else round(int(txt[0]) / int(txt[1]), 2)
            q = int(txt[0])
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return None if q <= 0 else round(q / per, 2)
        except Exception:
            continue
    return None


def cpu_baseline(W, cfg_no, rank, xtra):
    """The reference's CPU kernels timed on this box's host cores, on a bounded sample of the same workload.

    kind "reference": oracle/_ref/libbwaref.so (the reference's own C built in place by oracle/Makefile): the reference's
    ksw_extend2 under the builder's restatement of the extension() control loop (oracle/ref_shim.c) for boundary 2, and the
    reference's own mem_group_matesw (SSE2 ksw_align2 + its bookkeeping, native/bwamem_pair.c:115-228) for boundary 1 -- NOT
    the Scala+JNI path under a JVM (no JVM exists here); one unit per thread on every host core.
    kind "port": the scalar oracle (oracle/bpsw_oracle.c), one thread, when the reference build did not travel."""
    import pyoracle as po
    soa = make_ext_soa(W, cfg_no, rank, 0)
    n_grp = (READS_PER_EXT_BATCH // 2) // PAIRS_PER_GROUP if W["paired"] else 0   # groups holding the same number of reads as one wire batch
    groups = [make_group(W, cfg_no, rank, g) for g in range(n_grp)]
    unit_reads = READS_PER_EXT_BATCH
    if os.path.exists(po.REF_SO):
        from concurrent.futures import ThreadPoolExecutor
        ref = po.Ref()
        mat = po.default_mat()
        ropt = po.Oracle().default_opt()
        threads = max(1, min(len(os.sched_getaffinity(0)), 64))
        quota = cpu_quota()
        # `cores`: the CPUs the box really gives this process -- its cgroup quota when there is one (16 on a one-GPU box whose affinity
        # mask shows 64), else the affinity mask; the sample runs one thread per CPU of the mask all the same, the quota decides what they get
        cores = max(1, int(round(quota))) if quota else threads

        def unit(_):
            ref.extend_batch(soa, mat)
            for g in groups:
                ref.matesw_group(ropt, g)

        one = time.perf_counter()
        unit(0)
        one = time.perf_counter() - one
        reps = max(1, int(round(4.0 * cores / threads / max(one, 1e-3))))          # ~4 s of wall in all
        t0 = time.perf_counter()
        cpu_t0 = os.times()
        with ThreadPoolExecutor(threads) as ex:                     # ctypes releases the GIL inside the C loops
            list(ex.map(unit, range(threads * reps)))
        dt = time.perf_counter() - t0
        cpu_s = os.times()
        return {
            "value": round(threads * reps * unit_reads / dt, 1), "unit": "reads/s", "cores": cores, "kind": "reference",
            "threads": threads, "cpu_quota": quota, "cpus_busy": round(((cpu_s.user - cpu_t0.user) + (cpu_s.system - cpu_t0.system)) / dt, 1),
            "cores_note": "`cores` is the CPU quota of the box (cgroup cpu.max; the affinity mask when none is readable), `threads` what the sample ran on "
                          "(one per CPU of the affinity mask, at most 64), `cpus_busy` what the threads actually got",
            "sample": f"{threads * reps} units of {unit_reads} reads ({soa.n} extension tasks + {n_grp} rescue groups of {PAIRS_PER_GROUP} pairs) "
                      f"in {dt:.2f}s on {threads} threads; reference C kernels (scalar ksw_extend2 under the builder's batch loop, "
                      f"mem_group_matesw with SSE2 ksw_align2) from oracle/_ref, no JVM; 1 thread alone: {unit_reads / one:.0f} reads/s",
        }
    import bpsw_hip
    orc = po.Oracle()
    wire = bpsw_hip.wire_pack(soa)
    t0 = time.perf_counter()
    orc.wire_extend(wire)
    t_ext = time.perf_counter() - t0
    t_grp = 0.0
    if groups:
        t0 = time.perf_counter()
        orc.matesw_group(orc.default_opt(), groups[0], po.RESCUE_C)
        t_grp = time.perf_counter() - t0
    sec_per_read = t_ext / unit_reads + (t_grp / (2 * PAIRS_PER_GROUP) if groups else 0.0)
    return {"value": round(1.0 / sec_per_read, 1), "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"one wire batch of {unit_reads} reads in {t_ext:.2f}s + one rescue group of {PAIRS_PER_GROUP} pairs in {t_grp:.2f}s, "
                      f"oracle/bpsw_oracle.c single thread"}


def verify_sample(W, cfg_no, rank, wires, ext_outs, groups, grp_cnts, grp_regs, grp_totals, rng, ctx):
    """a random sample of the TIMED outputs against the oracle: ~2k extension tasks out of two wire batches, and whole
    rescue groups until >= 500 SW jobs are covered; raises on any difference.  The same sample gives the useful work per
    task / job for the GCUPS figures: DP cell updates as the oracle counts them (the cells the reference's SWExtend /
    SWAlign2 loops touch), split for the extension by how the kernel produced the side's result (bpsw_extend_batch_classify:
    exact shortcut = no cell was computed, DP swept = the cells were computed)."""
    import bpsw_hip
    import pyoracle as po
    orc = po.Oracle()
    n_ext = n_pairs = n_jobs = 0
    cells_closed = cells_dp = cells_sw = 0
    if wires:
        for b in rng.choice(len(wires), size=min(2, len(wires)), replace=False):
            soa = make_ext_soa(W, cfg_no, rank, int(b))
            sel = np.sort(rng.choice(soa.n, size=min(1024, soa.n), replace=False))
            sub_wire = bpsw_hip.wire_pack(soa.subset(sel))
            want, _, side_cells = orc.wire_extend_sides(sub_wire)
            got = ext_outs[int(b)].reshape(-1, 10)[sel]
            if not np.array_equal(got, np.asarray(want).reshape(-1, 10)):
                raise SystemExit(f"bench: extension outputs of timed wire batch {b} differ from the oracle")
            again, how = ctx.extend_batch_classify(sub_wire)
            if not np.array_equal(again, np.asarray(want)):
                raise SystemExit("bench: classify run differs from the oracle")
            cells_closed += int(side_cells[how == 1].sum())
            cells_dp += int(side_cells[how == 2].sum())
            n_ext += len(sel)
    gi = list(rng.permutation(len(groups)))
    while gi and n_jobs < 500:
        g = int(gi.pop())
        wcnt, wregs, jobs, cells = orc.matesw_group(orc.default_opt(), groups[g], po.RESCUE_C)
        got_cnt, got = grp_cnts[g], grp_regs[g][: int(grp_totals[g])]
        if not (np.array_equal(got_cnt, wcnt) and len(got) == len(wregs) and all(np.array_equal(got[f], wregs[f]) for f in got.dtype.names)):
            raise SystemExit(f"bench: rescue outputs of timed group {g} differ from the oracle")
        n_pairs += groups[g].group_size
        n_jobs += int(jobs)
        cells_sw += int(cells)
    work = {"ext_cells_per_task_closed_form": cells_closed / max(n_ext, 1), "ext_cells_per_task_dp_run": cells_dp / max(n_ext, 1),
            "sw_cells_per_job": cells_sw / max(n_jobs, 1)}
    return {"ext_tasks": int(n_ext), "rescue_pairs": int(n_pairs), "rescue_jobs": int(n_jobs), "mismatches": 0}, work


CHR21_LEN = 46_709_983   # GRCh38 chr21 (BASELINE.json configs[1..2]): the length of the counter-hash reference of the coordinate batches


def coordinate_batches_breakdown(W, cfg_no, rank, F, fd, wires, groups, structs, grp_cnts, grp_regs, passes, workers):
    """SURVEY.md 8f.2 in the timed path: the SAME step with the extension batches shipped as COORDINATE batches (wire format 2,
    include/bpsw.h: query flanks + the seed's reference coordinates; the target flanks are read from the 2-bit reference resident
    on the device) instead of format 1 (the as-is contract, which `value` is measured on).  The reads are drawn from a chr21-sized
    counter-hash reference (csrc/bpsw_synth.cpp: bpsw_synth_hash_pac / bpsw_synth_ext_tasks_ref -- the same generator as the
    format-1 batches with a reference behind it), a sample of the outputs is compared with the oracle run on the byte form of the
    same tasks.  Reports the rate and the bytes per task of both formats."""
    from concurrent.futures import ThreadPoolExecutor
    import bpsw_hip
    import pyoracle as po
    from bpsw_hip import synth
    pac = synth.hash_pac(CHR21_LEN, synth.CONFIG_SEED_BASE + 21)
    F.ctxs[0].ref_load(pac, CHR21_LEN)   # one copy per device, seen by every context of the device

    def one(b):
        by, co = synth.ext_tasks_ref(READS_PER_EXT_BATCH, pac, CHR21_LEN, read_len=W["read_len"], sub_rate=W["sub"], indel_rate=W["indel"],
                                     n_rate=0.001, tail_frac=W["tail_frac"], tail_sub_rate=W["tail_sub"], tail_indel_rate=W["tail_indel"],
                                     seed=ext_seed(cfg_no, rank, b) + 77)
        return bpsw_hip.wire_coords_pack(co), co.n, (int(bpsw_hip.wire_pack(by).size) if b < 2 else 0)

    with ThreadPoolExecutor(workers) as ex:
        made = list(ex.map(one, range(W["ext_batches"])))
    cw = [m[0] for m in made]
    cn = [m[1] for m in made]
    outs = [np.zeros(10 * n, np.int16) for n in cn]
    items, order = fd.make_items(cw, outs, groups, structs, grp_cnts, grp_regs)
    reps = max(2, passes // 2)
    F.run(items)                       # warm-up pass
    t0 = time.perf_counter()
    F.run(items, reps)
    dt = time.perf_counter() - t0
    # a sample of these outputs against the oracle on the byte form of the same tasks
    orc = po.Oracle()
    checked = 0
    for b in (0, len(cw) - 1):
        by, co = synth.ext_tasks_ref(READS_PER_EXT_BATCH, pac, CHR21_LEN, read_len=W["read_len"], sub_rate=W["sub"], indel_rate=W["indel"],
                                     n_rate=0.001, tail_frac=W["tail_frac"], tail_sub_rate=W["tail_sub"], tail_indel_rate=W["tail_indel"],
                                     seed=ext_seed(cfg_no, rank, b) + 77)
        sel = np.arange(0, by.n, max(1, by.n // 512))
        want, _ = orc.wire_extend(bpsw_hip.wire_pack(by.subset(sel)))
        if not np.array_equal(outs[b].reshape(-1, 10)[sel], np.asarray(want).reshape(-1, 10)):
            raise SystemExit(f"bench: coordinate-batch outputs of batch {b} differ from the oracle on the byte form of the same tasks")
        checked += len(sel)
    reads = READS_PER_EXT_BATCH * W["ext_batches"]
    f1 = sum(m[2] for m in made[:2]) / max(sum(cn[:2]), 1)
    f2 = sum(int(w.size) for w in cw) / max(sum(cn), 1)
    return {"reads_per_s": round(reads * reps / dt, 1), "ms_per_pass": round(1e3 * dt / reps, 3), "passes_timed": reps,
            "wire_bytes_per_task": {"format_1_as_is": round(f1, 1), "format_2_coordinates": round(f2, 1)},
            "reference": f"counter-hash 2-bit reference of {CHR21_LEN} bases (chr21-sized), resident on the device ({(CHR21_LEN + 3) // 4} bytes)",
            "verified_tasks": int(checked),
            "note": "same timed region as `value` (host buffers in and out, every boundary-1 group of the step included) with the extension "
                    "batches in wire format 2; needs the Scala driver to ship (rBeg, len) instead of leftRs / rightRs (INTEGRATION.md), so it is a "
                    "breakdown, not the metric"}


def device_resident_breakdown(W, cfg_no, rank, wires, ntasks, dev, local_rank, reps):
    """inputs already in HBM: sixteen wire batches + the rescue jobs of the same 262 144 pairs (in four launches), issued through
    the asynchronous device entries on twenty contexts; kernels only (no H2D/D2H, no boundary-1 host layer).  (Rounds 1-3 issued four
    batches and one rescue launch at a time -- five launch chains in flight against the host path's twenty: a lower rate than
    `value` for want of concurrency, not because of the entries.)"""
    import torch
    import bpsw_hip
    from bpsw_hip import synth
    nb = min(16, len(wires))
    n_sw = 4
    pairs = nb * READS_PER_EXT_BATCH // 2
    ctxs = [bpsw_hip.Context(local_rank) for _ in range(nb + n_sw)]
    opt = bpsw_hip.default_opt()
    xtra = bpsw_hip.KSW_XSUBO | bpsw_hip.KSW_XSTART | bpsw_hip.KSW_XBYTE | 19
    d_wires = [torch.from_numpy(w).to(dev) for w in wires[:nb]]
    d_outs = [torch.zeros(10 * n, dtype=torch.int16, device=dev) for n in ntasks[:nb]]
    n_jobs = int(pairs * 0.11) if W["paired"] else 0
    sjs = []
    keep = []
    if n_jobs:
        per = n_jobs // n_sw
        for q in range(n_sw):
            jobs = synth.sw_jobs(per, read_len=W["read_len"], win_min=400, win_max=400, sub_rate=W["mate_sub"], indel_rate=W["mate_indel"],
                                 unrelated_frac=0.05, decoy_frac=0.1, rev_frac=1.0, seed=ext_seed(cfg_no, rank, 0) + 100 + q)
            d_jobs = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in jobs.items()}
            d_sw_out = torch.zeros((per, 7), dtype=torch.int32, device=dev)
            sj = bpsw_hip.SwJobs()
            sj.n, sj.xtra = per, xtra
            for k in ("q_len", "t_len", "q_off", "t_off", "q_rev", "q_pool", "t_pool"):
                setattr(sj, k, d_jobs[k].data_ptr())
            sj.q_pool_bytes, sj.t_pool_bytes = d_jobs["q_pool"].numel(), d_jobs["t_pool"].numel()
            sjs.append((sj, d_sw_out))
            keep.append(d_jobs)
        n_jobs = per * n_sw
    torch.cuda.synchronize(dev)

    def step():
        for cx, w, n, dw, do in zip(ctxs, wires[:nb], ntasks[:nb], d_wires, d_outs):
            cx.extend_batch_device(dw.data_ptr(), int(w.size), n, do.data_ptr(), 0)
        for q, (sj, d_sw_out) in enumerate(sjs):
            ctxs[nb + q].swalign2_batch_device(opt, sj, d_sw_out.data_ptr(), 0)

    def wait():
        for cx in ctxs:
            cx.last_kernel_ms()

    for _ in range(3):
        step()
        wait()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
        wait()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    for cx in ctxs:
        cx.close()
    return {"reads_per_s": round(2 * pairs / dt, 1), "ms_per_65536_pairs": round(1e3 * dt * 65536 / pairs, 4), "pairs": pairs, "rescue_jobs": n_jobs,
            "note": "inputs resident in HBM, asynchronous *_device entries, kernels only: no H2D/D2H, no boundary-1 host layer; NOT `value`"}


def tail_breakdown(ctx, opt, n_pairs: int = 4096):
    """bpsw_sam_pe_batch on a synthetic group (regions from the device's own memChainToAln + memSortAndDedup): reads/s of the
    whole call for ONE calling thread (host passes + staging + kernel; host buffers in, SAM text out) and the kernel's rate."""
    import bpsw_hip
    from bpsw_hip import synth
    pac, bases, off, ln, names, dups = synth.contig_reference([400_000, 300_000, 200_000, 100_000], seed=synth.CONFIG_SEED_BASE + 40)
    tb, rn, quals, pes = synth.tail_pairs(n_pairs, bases, off, ln, dups, seed=synth.CONFIG_SEED_BASE + 41)
    ctx.ref_load(pac, int(off[-1] + ln[-1]))
    ctx.bns_load(off, ln, names)
    cnt, regs = ctx.chain2aln_batch(opt, tb, flags=bpsw_hip.C2A_SORT_DEDUP)
    g = bpsw_hip.make_tail_group(tb, rn, quals, pes, cnt, regs, off, ln, names, id0=0)
    topt = bpsw_hip.default_tail_opt()
    ctx.sam_pe_batch(opt, topt, g)
    reps, k_ms, host = 3, 0.0, (0.0, 0.0, 0.0)
    for _ in range(reps):
        texts, _ = ctx.sam_pe_batch(opt, topt, g)
        ms, jobs = ctx.last_tail_kernel()
        k_ms += ms
        host = tuple(a + b for a, b in zip(host, ctx.last_tail_host_ms()))
    call_ms = sum(host) / reps
    return {"pairs": n_pairs, "reg2aln_jobs": int(jobs), "call_ms": round(call_ms, 3), "kernel_ms": round(k_ms / reps, 4),
            "reads_per_s_one_thread": round(2 * n_pairs / (call_ms * 1e-3), 1), "alignments_per_s_kernel": round(jobs / (k_ms / reps * 1e-3), 1),
            "host_ms": {"plan": round(host[0] / reps, 3), "device_roundtrip": round(host[1] / reps, 3), "sam_text": round(host[2] / reps, 3)},
            "sam_bytes": int(sum(len(t) for t in texts)), "note": "host buffers in, SAM text out; not part of `value`",
            "pool": tail_pool_rate(ctx, opt, topt, g, sum(len(t) for t in texts))}


def tail_pool_rate(ctx, opt, topt, g, text_bytes, workers: int = 8, in_flight: int = 32, rounds: int = 3):
    """The same group through bpsw_tail_pool_*: THIS thread only enqueues `in_flight` tickets and collects them (tools/tail_pool_rate.py
    sweeps the number of workers: profiles/r05_tail_pool.json)."""
    import ctypes as C
    from bpsw_hip import _pairs_struct, _ptr
    lib = ctx.lib
    st, keep, _ = _pairs_struct(g)
    cap = int(text_bytes) + 4096
    bufs = [np.empty(cap, np.uint8) for _ in range(in_flight)]
    offs = [np.zeros(2 * g.group_size + 1, np.int64) for _ in range(in_flight)]
    pool, need = C.c_void_p(), C.c_size_t(0)
    rc = lib.bpsw_tail_pool_create(lib.bpsw_device_of(ctx.h), workers, C.byref(pool))   # (the rank's device, not device 0: advisor, round 5)
    if rc != 0:
        return {"error": lib.bpsw_last_error().decode()}
    tick = [0] * in_flight

    def one_round():
        for k in range(in_flight):
            t = C.c_int64(0)
            if lib.bpsw_tail_pool_submit(pool, C.byref(opt), C.byref(topt), C.byref(st), -1, _ptr(bufs[k]), cap, _ptr(offs[k]), None, None, 0, C.byref(t)) != 0:
                raise RuntimeError(lib.bpsw_last_error().decode())
            tick[k] = t.value
        for k in range(in_flight):
            if lib.bpsw_tail_pool_wait(pool, tick[k], C.byref(need), None) != 0:
                raise RuntimeError(lib.bpsw_last_error().decode())
    try:
        one_round()
        t0 = time.perf_counter()
        for _ in range(rounds):
            one_round()
        dt = time.perf_counter() - t0
    finally:
        lib.bpsw_tail_pool_destroy(pool)
    return {"workers": workers, "groups_in_flight": in_flight, "reads_per_s_one_calling_thread": round(2 * g.group_size * in_flight * rounds / dt, 1),
            "ms_per_group": round(1e3 * dt / (in_flight * rounds), 3),
            "note": "the calling thread enqueues and collects; plan, kernel and text run on the library's tail workers"}


def host_cpu_sweep(fd, items, n_threads, local_rank, opt, mode, stage_commit, cpu_pool, counts, reads_per_step, passes, steps=3):
    """The timed region once more with the feeder's threads confined to the first c CPUs of `cpu_pool`, c in `counts`: what the step
    makes of fewer host CPUs per GPU -- the proxy for the 8-GPU node, where eight ranks share the node's cores and memory system and
    the slope of the scaling curve is set by what a rank needs of them.  The same items, the same number of
    threads (they block on the device most of the time), `steps` steps after one warm-up pass; the HIP runtime's own helper threads
    are not confined.  Outside the timed region of `value`."""
    out = {}
    for c in counts:
        if c > len(cpu_pool):
            continue
        F2 = fd.Feeder(n_threads, local_rank, opt, mode, cpus=cpu_pool[:c])
        try:
            if stage_commit:
                F2.use_stage_commit(True)
            F2.run(items, passes)          # warm-up: buffers grown, rings started
            c0 = os.times()
            t0 = time.perf_counter()
            F2.run(items, steps * passes)
            dt = time.perf_counter() - t0
            c1 = os.times()
            out[str(c)] = {"reads_per_s": round(reads_per_step * steps / dt, 1),
                           "cpus_busy": round(((c1.user - c0.user) + (c1.system - c0.system)) / dt, 2)}
        finally:
            F2.close()
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: the N ranks as children of torch.distributed.run on 127.0.0.1 (this process never
    initialises the GPU; the children inherit stdout, so rank 0's JSON line is this command's line).  Returns the exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def stub_main(args, W):
    """BENCH_STUB=1: the launch / rendezvous / reduction skeleton of main() with a sleep for a workload and no GPU call -- what the CPU
    tests run to check that `--gpus N` really is N ranks (tests/test_multigpu_cpu.py).  Its line is marked `stub` and is not a measurement."""
    import torch  # noqa: F401
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("BENCH_BACKEND", "gloo")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.02 * (rank + 1) * max(args.steps, 1))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ranks_seen = world
    if world > 1:
        elapsed, ranks_seen = reduce_over_ranks(elapsed, "cpu")
    if ranks_seen != args.gpus:
        print(f"bench: {ranks_seen} ranks took part, --gpus {args.gpus}", file=sys.stderr)
        sys.exit(3)
    reads_per_step = READS_PER_EXT_BATCH * W["ext_batches"] * W["passes"]
    if rank == 0:
        print(json.dumps({"metric": W["metric"], "value": whole_job_rate(reads_per_step, args.steps, world, elapsed), "unit": "reads/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True,
                          "scaling": "weak", "ranks_seen": ranks_seen, "stub": True, "data": "none (BENCH_STUB=1: launch skeleton only)"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=int(os.environ.get("BENCH_CONFIG", "3")), choices=sorted(WORKLOADS),
                    help="SURVEY.md 8(d) workload number (3 = BASELINE.json configs[2], the metric's configuration)")
    ap.add_argument("--threads", type=int, default=int(os.environ.get("BENCH_THREADS", "0")), help="host feeder threads per GPU (0: auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tail", action="store_true", help="skip the worker2-tail breakdown entry")
    ap.add_argument("--no-extras", action="store_true", help="skip every breakdown outside the timed region")
    ap.add_argument("--cpu-sweep", action="store_true",
                    help="after the timed region, run its items again with the feeder confined to 4 / 8 / 12 / 16 CPUs (host.reads_per_s_vs_cpus measured live; "
                         "without it the line carries the committed sweep of profiles/<tag>_cpu_sweep.json, labelled as such)")
    args = ap.parse_args()
    W = dict(WORKLOADS[args.config])
    shrink = max(1, int(os.environ.get("BENCH_SHRINK", "1")))   # rehearsals (tests): 1/k of the batches, groups and passes; the line says so and is not the metric
    if shrink > 1:
        W["ext_batches"] = max(1, W["ext_batches"] // shrink)
        W["groups"] = W["groups"] // shrink if W["groups"] else 0
        W["passes"] = max(1, W["passes"] // shrink)

    # ---- N ranks: one process per GPU (the reference's parallelism is mapPartitions over Spark partitions,
    # FastMap.scala:266-293).  Under a launcher (WORLD_SIZE set) the world must be what --gpus says; without one, --gpus N > 1 starts
    # the N ranks itself -- before anything in this process has touched the GPU -- and passes rank 0's line through.
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    if world_env is not None and int(world_env) != args.gpus:
        print(f"bench: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("BENCH_STUB") == "1":
        return stub_main(args, W)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    # rehearsal knobs for a one-GPU box: BENCH_FORCE_DEVICE=0 puts every rank on that device, BENCH_BACKEND=gloo replaces RCCL
    # (two ranks cannot share one GPU under RCCL).  The driver's multi-GPU runs use neither.
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    backend = os.environ.get("BENCH_BACKEND", "nccl")

    # ---- host placement: feeder threads (and the pinned staging they first touch) on the NUMA node of this rank's GPU ----
    numa_cpus, numa_node = gpu_numa_cpus(local_rank)
    allowed = sorted(os.sched_getaffinity(0))
    ranks_on_node = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    share = numa_cpus if numa_cpus else allowed[(local_rank % ranks_on_node)::ranks_on_node] if distributed else allowed
    # 32: the device phases of the calls share a pool of 20 streams (bpsw_internal.h, StreamLease), and half as many threads
    # again keep it full while the others stage bytes or replay bookkeeping; more change nothing
    # (configs[2]: 48 -- forty-eight task threads keep a few more calls in flight while others stage or replay: 2.39-2.52 x 10^8 against 2.19-2.48
    # with 40 over rounds 5-6, 14.5 of the 16 CPUs busy; 56 is another 1-2 % at 15.5 CPUs busy, too close to the quota.  The other workloads
    # stay at 40: configs[1]'s forty-eight callers crowd the copy lane and the twenty streams (3.6 -> 2.6 x 10^8), configs[4] loses 6 %.)
    n_threads = args.threads if args.threads > 0 else max(2, min(48 if args.config == 3 else 40, len(share)))
    if share and (numa_cpus or distributed):
        try:
            os.sched_setaffinity(0, share)   # before the inputs are generated: first touch puts them on the GPU's node
        except OSError:
            pass

    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import bpsw_hip
    from bpsw_hip import feeder as fd
    import ctypes as C
    opt = bpsw_hip.default_opt()
    xtra = bpsw_hip.KSW_XSUBO | bpsw_hip.KSW_XSTART | bpsw_hip.KSW_XBYTE | 19   # MemSamPe.scala:1187-1189

    t_gen = time.perf_counter()
    wires, ntasks, groups = build_inputs(W, args.config, rank, workers=max(2, min(16, len(allowed))))
    t_gen = time.perf_counter() - t_gen
    ext_outs = [np.zeros(10 * n, np.int16) for n in ntasks]
    structs = [g.as_struct() for g in groups]
    grp_cnts = [np.zeros(2 * g.group_size, np.int32) for g in groups]
    grp_regs = [np.empty(int(g.regs.shape[0] + g.ref_rb.shape[0] + 16), bpsw_hip.ALNREG_DTYPE) for g in groups]
    only = os.environ.get("BENCH_ONLY", "")      # diagnostics: "ext" / "grp" time one boundary alone (the JSON line then is not the metric)
    if only == "ext":
        groups, structs, grp_cnts, grp_regs = [], [], [], []
    elif only == "grp":
        wires, ntasks, ext_outs = [], [], []
    items, order = fd.make_items(wires, ext_outs, groups, structs, grp_cnts, grp_regs)
    F = fd.Feeder(n_threads, local_rank, opt, bpsw_hip.RESCUE_C, cpus=share if (numa_cpus or distributed) else None)  # no fallback: raises without a gfx950 device
    # the extension calls the way the JNI shim makes them (csrc/bpsw_jni.cpp: bpsw_extend_stage, GetByteArrayRegion into the pinned block,
    # bpsw_extend_commit, SetShortArrayRegion from the pinned result block); BENCH_EXT_ENTRY=batch: bpsw_extend_batch, as rounds 1-4 timed
    ext_entry = os.environ.get("BENCH_EXT_ENTRY", "stage_commit")
    if ext_entry == "stage_commit":
        F.use_stage_commit(True)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    passes = W["passes"]
    if args.warmup > 0:
        F.run(items, args.warmup * passes)
    F.reset_stats()
    barrier()
    # (the device-wide wait in barrier() has let the open epoch of the submission ring close: the epochs of the timed region are whole ones)
    ring0 = F.ctxs[0].ring_stats() + F.ctxs[0].ring_epoch_times()
    cpu0 = os.times()
    t0 = time.perf_counter()
    # the K steps in one go: the feeder threads go round the passes without waiting for each other, as the task threads of an
    # executor would (rounds 1-2 put a barrier behind every pass -- with configs[1]'s 32 batches per pass on 32 threads every pass
    # was one burst of calls, all staging and copying at once, and the device sat idle between bursts: DESIGN.md 5.2)
    F.run(items, args.steps * passes)   # every call complete, results in host memory
    barrier()
    elapsed = time.perf_counter() - t0
    cpu1 = os.times()
    ring1 = F.ctxs[0].ring_stats() + F.ctxs[0].ring_epoch_times()
    cpu_busy = ((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)) / elapsed   # host CPUs busy during the timed region (this rank)
    st = F.stats_sum()
    call_ms = {"extend": [it.ms for it in items if it.kind == 0], "matesw_group": [it.ms for it in items if it.kind == 1]}
    call_cpu_ms = {"extend": [it.cpu_ms for it in items if it.kind == 0], "matesw_group": [it.cpu_ms for it in items if it.kind == 1]}
    grp_totals = [0] * len(groups)
    for it, (kind, i) in zip(items, order):
        if kind == 1:
            grp_totals[i] = it.out_total
    ranks_seen = world
    if distributed:
        elapsed, ranks_seen = reduce_over_ranks(elapsed, dev if backend == "nccl" else "cpu")
    if ranks_seen != args.gpus:
        print(f"bench: {ranks_seen} ranks took part in the timed region, --gpus {args.gpus}", file=sys.stderr)
        sys.exit(3)

    reads_per_pass = READS_PER_EXT_BATCH * W["ext_batches"]
    assert only or shrink > 1 or not W["paired"] or reads_per_pass // 2 == PAIRS_PER_GROUP * W["groups"] or PAIRS_PER_GROUP != 4096
    reads_per_step = reads_per_pass * passes
    pairs_per_step = reads_per_step // 2 if W["paired"] else 0
    value = whole_job_rate(reads_per_step, args.steps, world, elapsed)

    # ---- a sample of the timed outputs against the oracle (outside the timed region; the only use of oracle/ besides cpu_baseline)
    verified, work = verify_sample(W, args.config, rank, wires, ext_outs, groups, grp_cnts, grp_regs, grp_totals,
                                   np.random.default_rng(1234 + rank), F.ctxs[0]) if rank == 0 else (None, None)

    # ---- per-kernel figures from the HIP events the library records on its launch streams during the timed region ----
    ext_launches, sw_launches = int(st["ext_calls"]), int(st["sw_calls"])
    ext_avg_ms = st["ext_kernel_ms"] / max(ext_launches, 1)
    sw_avg_ms = st["sw_kernel_ms"] / max(sw_launches, 1)
    ext_bytes = (sum(int(w.size) for w in wires) + 20 * sum(ntasks)) / max(len(wires), 1)      # per launch (SURVEY.md 8d B_ext)
    win_len = float(np.mean([float(g.ref_len[g.ref_len > 0].mean()) for g in groups[:8]])) if groups else 0.0
    sw_bytes = (st["sw_jobs"] / max(sw_launches, 1)) * (W["read_len"] + win_len + 28)           # per launch (B_sw)
    # per-launch instruction counts of this command from the committed rocprofv3 --pmc passes (profiles/pmc_issue.json)
    pi = load_committed_json("pmc_issue.json")
    e_cnt = (pi.get("extend_per_call") or pi.get("extend")) if args.config == 3 else None
    w_cnt = pi.get("swalign2") if args.config == 3 else None
    # the committed rocprofv3 --kernel-trace --stats summary of this same command: every kernel's average duration there (only for the
    # workload it was taken on, configs[2])
    trace_avg, trace_share = trace_figures(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_kernel_stats_bench.csv")) if args.config == 3 and not only else ({}, {})
    if e_cnt and w_cnt:
        ext_instr = ext_launches * (e_cnt["valu"] + e_cnt["salu"])
        sw_instr = sw_launches * (w_cnt["valu"] + w_cnt["salu"])
    else:
        ext_instr = sw_instr = None
    # Both kernels' figures are in `kernels`, their fractions of the HBM peak side by side in `frac_by_kernel`.
    dominant, dominant_by = pick_dominant(trace_share, ext_instr, sw_instr, st["ext_kernel_ms"], st["sw_kernel_ms"], PROFILE_TAG)
    # The rescue batches of the timed region went through the device's submission ring (csrc/bpsw_ring.h): no launch per batch.  `swalign2`
    # below is a BATCH -- its duration the span first job pair taken -> last one finished on the device's clock -- and `swalign2_resident` is
    # the resident kernel that served them: launches = epochs, each timed by two HIP events around its launch on the ring's stream.
    ring_epochs = int(ring1[4] - ring0[4])
    ring_epoch_ms = (ring1[3] - ring0[3]) / ring_epochs if ring_epochs > 0 else 0.0
    ring_batches = int(ring1[1] - ring0[1])
    sw_via_ring = int(st.get("sw_ring_calls", 0))
    per_kernel = {"extend": (ext_bytes, ext_avg_ms, ext_launches, "ext_kernel"), "swalign2": (sw_bytes, sw_avg_ms, sw_launches, "swp_kernel")}
    dom_bytes, dom_ms = per_kernel[dominant][0], per_kernel[dominant][1]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    pmc = load_committed_json("pmc_traffic.json")   # written from rocprofv3 --pmc passes
    traffic = pmc.get(dominant)
    traffic_x2 = pmc.get("detail", {}).get(dominant, {}).get("fetch_size_x2_plus_write_size")
    step_s_all = elapsed / args.steps
    # where the dominant kernel's wave-cycles go (committed SQ counters of this command, profiles/<tag>_sq_activity.json)
    sq_activity = None
    sa = load_committed_json(f"{PROFILE_TAG}_sq_activity.json")
    if sa:
        sq_activity = {k: {q: sa[k][q] for q in ("active_valu", "active_sca", "wait_inst_any", "wait_any") if q in sa[k]} for k in ("extend", "swalign2") if k in sa}
        sq_activity["note"] = f"fractions of SQ_WAVE_CYCLES, from the committed profiles/{PROFILE_TAG}_sq_activity.json (rocprofv3 --pmc pass of this command)"
    both = {}
    for k, (b, ms, n_l, _) in per_kernel.items():
        if n_l:
            both[k] = {"achieved_GBps": round(b / (ms * 1e-3) / 1e9, 3) if ms > 0 else 0.0, "frac_of_hbm_peak": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6) if ms > 0 else 0.0,
                       "algorithmic_bytes_per_launch": int(b), "avg_launch_ms": round(ms, 4), "avg_launch_ms_kernel_trace": trace_avg.get(k),
                       "launches_per_step": round(n_l / args.steps, 1), "traffic_per_launch": pmc.get(k)}
    if sw_via_ring and "swalign2" in both:
        both["swalign2"]["via"] = f"submission ring: {sw_via_ring} of {sw_launches} batches; avg_launch_ms = a batch's span on the device clock (first job pair taken -> last finished), not a launch"
        both["swalign2"]["avg_launch_ms_kernel_trace"] = None
    if ring_epochs > 0:
        ep_bytes = sw_bytes * ring_batches / ring_epochs
        both["swalign2_resident"] = {"achieved_GBps": round(ep_bytes / (ring_epoch_ms * 1e-3) / 1e9, 3), "frac_of_hbm_peak": round(ep_bytes / (ring_epoch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
                                     "algorithmic_bytes_per_launch": int(ep_bytes), "avg_launch_ms": round(ring_epoch_ms, 3),
                                     "avg_launch_ms_kernel_trace": trace_avg.get("swalign2_resident"), "launches": ring_epochs,
                                     "batches_per_launch": round(ring_batches / ring_epochs, 1),
                                     "share_of_kernel_time_in_trace": trace_avg.get("swalign2_resident_share_of_kernel_time"),
                                     "note": "swp_resident_kernel: one launch per EPOCH of the device's submission ring (it ends after BPSW_RING_IDLE_US without a batch, or when its "
                                             "16 384 descriptor slots are used up); launches and durations from HIP events around each launch on the ring's stream"}
    algo_bytes_step = (ext_bytes * ext_launches + sw_bytes * sw_launches) / args.steps
    summed_kernel_s = (st["ext_kernel_ms"] + st["sw_kernel_ms"]) * 1e-3 / args.steps
    # instruction issue: raw rate, the vector pipe's utilisation at the guide's 2 cycles per wave64 instruction, and -- labelled as
    # what it is -- the fraction of the rate these kernels' own mix was measured to sustain (2.6 cycles per instruction)
    issue = None
    if e_cnt and w_cnt:
        simd_cycles = 1024 * 2.4e9 * elapsed
        valu_i = ext_launches * e_cnt["valu"] + sw_launches * w_cnt["valu"]
        issue = {"wave_instructions_in_timed_region": int(ext_instr + sw_instr), "valu_instructions": int(valu_i),
                 "instr_per_cycle_per_simd": round((ext_instr + sw_instr) / simd_cycles, 4),
                 "valu_pipe_utilisation_at_2_cycles": round(valu_i * 2.0 / simd_cycles, 4),
                 "share_of_instructions": {"extend": round(ext_instr / (ext_instr + sw_instr), 3), "swalign2": round(sw_instr / (ext_instr + sw_instr), 3)},
                 "self_calibrated": {"frac_of_rate_measured_on_these_kernels": round((ext_instr + sw_instr) * 2.6 / simd_cycles, 3), "cycles_per_instruction": 2.6,
                                     "note": "2.6 cycles per instruction is what a SIMD was measured to sustain on THESE kernels' own mix (DESIGN.md 4.1): this figure "
                                             "says how close the step runs to that, not to the machine's peak"},
                 "simds": 1024, "clock_hz": 2.4e9,
                 "note": "vector + scalar instructions only (branches and waits excluded); per-launch counts from profiles/pmc_issue.json (rocprofv3 --pmc "
                         "passes of this command); MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32; most of these "
                         "kernels' vector instructions (max / min / compare / select / DPP) were measured at half that rate (profiles/archive/r01_microbench_issue.txt)"}
    issue_frac = issue
    host_ms = {k: {"mean": round(float(np.mean(v)), 4), "p50": round(float(np.median(v)), 4), "max": round(float(np.max(v)), 4)} if v else None
               for k, v in call_ms.items()}
    pcie_bytes_per_step = passes * (sum(int(w.size) for w in wires) + 20 * sum(ntasks))   # boundary 2 both ways; boundary 1 below
    pcie_bytes_per_step += int((st["sw_jobs"] / max(args.steps, 1)) * (W["read_len"] + win_len + 28 + 29))

    # ---- useful work: DP cell updates per second, and what fraction of the integer-VALU ceiling they are.
    # Cells as the reference's loops count them (oracle); an extension side the kernel resolved by an exact shortcut contributes
    # its cells to `closed_form` (they were never computed), a swept side to `dp_run`.  Ceiling: 256 CUs x 4 SIMD-32 x 2.4 GHz
    # int32 lane-operations per second, at ~12 operations per extension cell (SWUtil.scala:151-170) and ~11 per
    # local-SW cell (SWUtil.scala:484-505).
    gcups = valu = None
    if work is not None:
        step_s = elapsed / args.steps
        ext_tasks_step = passes * float(sum(ntasks))
        sw_jobs_step = st["sw_jobs"] / max(args.steps, 1)
        c_closed = ext_tasks_step * work["ext_cells_per_task_closed_form"]
        c_dp = ext_tasks_step * work["ext_cells_per_task_dp_run"]
        c_sw = sw_jobs_step * work["sw_cells_per_job"]
        gcups = {"extend_dp_run": round(c_dp / step_s / 1e9, 2), "extend_closed_form": round(c_closed / step_s / 1e9, 2),
                 "rescue_sw": round(c_sw / step_s / 1e9, 2), "computed": round((c_dp + c_sw) / step_s / 1e9, 2),
                 "reference_equivalent": round((c_dp + c_closed + c_sw) / step_s / 1e9, 2),
                 "cells_per_ext_task": round(work["ext_cells_per_task_closed_form"] + work["ext_cells_per_task_dp_run"], 1),
                 "cells_per_rescue_job": round(work["sw_cells_per_job"], 1),
                 "note": "per rank; cells = the reference's DP cell updates (oracle counters on the verified sample); closed_form = cells of "
                         "extension sides an exact shortcut resolved without computing them"}
        ops_run = 12.0 * c_dp + 11.0 * c_sw
        ops_ref = 12.0 * (c_dp + c_closed) + 11.0 * c_sw
        valu = {"computed_cells": round(ops_run / step_s / INT_VALU_PEAK_OPS, 4), "reference_equivalent_cells": round(ops_ref / step_s / INT_VALU_PEAK_OPS, 4),
                "ceiling_int32_lane_ops_per_s": INT_VALU_PEAK_OPS, "ops_per_cell": {"extend": 12, "rescue_sw": 11},
                "note": "ceiling = 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md); rounds 1-2 priced against half of it (SIMD-16)"}

    extras = {}
    if not args.no_extras:
        try:
            extras["device_resident"] = device_resident_breakdown(W, args.config, rank, wires, ntasks, dev, local_rank, reps=max(3, min(args.steps, 10)))
        except Exception as e:  # noqa: BLE001 -- a breakdown line must never cost the bench its JSON
            traceback.print_exc(file=sys.stderr)   # (recorded in the line as {"error": ...} AND shown: nothing is swallowed)
            extras["device_resident"] = {"error": repr(e)}
        if rank == 0 and world == 1 and wires and only == "":
            try:
                extras["coordinate_batches"] = coordinate_batches_breakdown(W, args.config, rank, F, fd, wires, groups, structs, grp_cnts, grp_regs,
                                                                            passes, max(2, min(8, len(share))))
            except Exception as e:  # noqa: BLE001
                traceback.print_exc(file=sys.stderr)   # (recorded in the line as {"error": ...} AND shown: nothing is swallowed)
                extras["coordinate_batches"] = {"error": repr(e)}
        if rank == 0 and world == 1 and wires and groups:
            # what the JNI shim adds around the C ABI calls the timed region makes (fake JNIEnv: a JVM exists on neither box)
            try:
                from bpsw_hip import jnishim, synth
                js = jnishim.shim_rate(wires[0], ntasks[0], groups[0], reps=5)
                # the flat entry with the windows named by coordinates needs a group drawn from a reference: a small one of its own
                try:
                    l_pac_s = 2_000_003
                    pac_s, bases_s = synth.random_pac(l_pac_s, seed=synth.CONFIG_SEED_BASE + 91)
                    g_ref = synth.rescue_group(1024, seed=synth.CONFIG_SEED_BASE + 92, l_pac=l_pac_s, p_resc=W["p_resc"], ref_bases=bases_s)
                    js["mateSWFlatJNI_coordinates"] = jnishim.shim_rate(wires[0], ntasks[0], g_ref, reps=5, pac=pac_s)["mateSWFlatJNI_coordinates"]
                except Exception as e:  # noqa: BLE001
                    traceback.print_exc(file=sys.stderr)   # (recorded in the line as {"error": ...} AND shown: nothing is swallowed)
                    js["mateSWFlatJNI_coordinates"] = {"error": repr(e)}
                # what a Scala caller could reach THROUGH the shim with the bench's number of task threads: a unit of 32 768 reads is one
                # extension call and four rescue calls; per thread it takes the calls' loaded latencies (measured in the timed region)
                # plus the marshalling measured here -- an estimate, never above the device-bound `value`
                try:
                    ext_shim = (js["swExtendFPGAJNI"]["marshal_in_us"] + js["swExtendFPGAJNI"]["marshal_out_us"]) * 1e-3
                    ext_ms = host_ms["extend"]["mean"] + ext_shim
                    grp_ms = host_ms["matesw_group"]["mean"] if host_ms.get("matesw_group") else 0.0
                    est = {}
                    for key in ("mateSWJNI", "mateSWFlatJNI", "mateSWFlatJNI_coordinates"):
                        if key not in js or "marshal_in_us" not in js[key]:
                            continue
                        scale = PAIRS_PER_GROUP / max(js[key]["pairs_per_call"], 1)   # (the coordinate group is smaller: per-pair cost scaled)
                        shim_ms = (js[key]["marshal_in_us"] + js[key]["marshal_out_us"]) * 1e-3 * scale
                        unit_ms = ext_ms + (READS_PER_EXT_BATCH // (2 * PAIRS_PER_GROUP)) * (grp_ms + shim_ms) if W["paired"] else ext_ms
                        est[key] = round(min(value, n_threads * READS_PER_EXT_BATCH / (unit_ms * 1e-3)), 1)
                    js["through_shim_reads_per_s_est"] = dict(est, threads=n_threads,
                                                              note="min(value, threads x 32768 reads / (loaded call latencies of one extension call and four rescue calls + "
                                                                   "their marshalling through the fake JNIEnv)); mateSWJNI = the reference's object-array contract as it is")
                except Exception as e:  # noqa: BLE001
                    traceback.print_exc(file=sys.stderr)   # (recorded in the line as {"error": ...} AND shown: nothing is swallowed)
                    js["through_shim_reads_per_s_est"] = {"error": repr(e)}
                extras["jni_shim_fake_env"] = js
            except Exception as e:  # noqa: BLE001
                traceback.print_exc(file=sys.stderr)   # (recorded in the line as {"error": ...} AND shown: nothing is swallowed)
                extras["jni_shim_fake_env"] = {"error": repr(e)}
        if rank == 0 and world == 1 and not args.no_tail and args.config == 3:
            try:
                extras["worker2_tail"] = tail_breakdown(F.ctxs[0], opt)
            except Exception as e:  # noqa: BLE001
                traceback.print_exc(file=sys.stderr)   # (recorded in the line as {"error": ...} AND shown: nothing is swallowed)
                extras["worker2_tail"] = {"error": repr(e)}

    # (not part of the default command: its sixteen thousand extra launches under a lighter load would sit in the kernel trace of the
    # command beside the timed region's and pull the trace's per-kernel averages away from the line's own)
    cpu_sweep = None
    if rank == 0 and world == 1 and args.config == 3 and not args.cpu_sweep:
        committed = load_committed_json(f"{PROFILE_TAG}_cpu_sweep.json")
        if committed:
            cpu_sweep = dict(committed.get("reads_per_s_vs_cpus", {}), unconfined=committed.get("unconfined"),
                             source=f"COMMITTED profiles/{PROFILE_TAG}_cpu_sweep.json (python bench.py --cpu-sweep), not measured in this run")
    if rank == 0 and world == 1 and args.cpu_sweep and only == "" and args.config == 3:
        pool = list(share) if share else sorted(allowed)
        try:
            cpu_sweep = host_cpu_sweep(fd, items, n_threads, local_rank, opt, bpsw_hip.RESCUE_C, ext_entry == "stage_commit", pool,
                                       [c for c in (4, 8, 12, 16) if c <= len(pool)], reads_per_step, passes)
            cpu_sweep["note"] = ("the timed region's items with the feeder threads confined (affinity) to the first c CPUs of this rank's share, 3 steps each, "
                                 "outside the timed region of `value`; cpu_s_per_Mreads of the unconfined run says what a read costs, this says what the "
                                 "step makes of a smaller share -- eight ranks on one node get (node CPUs / 8) each")
        except Exception as e:  # noqa: BLE001
            traceback.print_exc(file=sys.stderr)   # (recorded in the line as {"error": ...} AND shown: nothing is swallowed)
            cpu_sweep = {"error": repr(e)}

    out = {
        "metric": W["metric"],
        "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32+u16", "data": "synthetic",
        "config": {"workload": W["label"], "survey_config": args.config,
                   "dtype_note": "extension DP in int32; rescue SW in packed u16, two jobs per wavefront (exact: a pass stops at the 255 score cap, SWUtil.scala:423,537)",
                   "pairs_per_step_per_gpu": pairs_per_step, "reads_per_step_per_gpu": reads_per_step,
                   "passes_per_step": passes, "distinct_reads_per_pass": reads_per_pass,
                   "ext_batches_per_step": passes * W["ext_batches"], "reads_per_ext_batch": READS_PER_EXT_BATCH,
                   "ext_tasks_per_step": passes * int(sum(ntasks)), "rescue_groups_per_step": passes * W["groups"], "pairs_per_group": PAIRS_PER_GROUP,
                   "rescue_jobs_per_step": int(st["sw_jobs"] / max(args.steps, 1)),
                   "reads_streamed_in_timed_region": int(reads_per_step * args.steps * world),
                   "ext_entry": ext_entry,
                   "timed_region": "host buffers in, host buffers out: bpsw_extend_stage + memcpy + bpsw_extend_commit + memcpy per wire batch (the JNI shim's sequence; "
                                   "BENCH_EXT_ENTRY=batch: bpsw_extend_batch) + bpsw_matesw_group per group "
                                   "(H2D, kernels, D2H, speculate/replay, sort/dedup all inside)",
                   "host_threads_per_gpu": n_threads, "numa_node": numa_node, "feeder_cpus": len(share),
                   "hip_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "ranks_seen": ranks_seen, "shrink": shrink,
                   "parallelism": f"partition->device x{world} (no collective)", "input_generation_s": round(t_gen, 1)},
        "verified": verified,
        "roofline": {"bound": "hbm", "kernel": dominant, "dominant_by": dominant_by, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_2xfetch_plus_write": traffic_x2,
                     "traffic_note": "profiles/pmc_traffic.json, per launch: L2->fabric read requests by their size (FETCH_SIZE tallies each at 64 B) + "
                                     "WRITE_SIZE; the second figure is the blanket 2*FETCH_SIZE + WRITE_SIZE",
                     "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 4),
                     "avg_launch_ms_kernel_trace": trace_avg.get(dominant),
                     "kernels": both, "frac_by_kernel": {k: v.get("frac_of_hbm_peak") for k, v in both.items()},
                     "share_of_kernel_time_in_trace": {k: round(v, 4) for k, v in trace_share.items()},
                     "aggregate_GBps": round(algo_bytes_step / step_s_all / 1e9, 3),
                     "aggregate_frac_of_hbm_peak": round(algo_bytes_step / step_s_all / 1e9 / HBM_PEAK_GBPS, 6),
                     "launch_overlap": round(summed_kernel_s / step_s_all, 2),
                     "issue": sq_activity,
                     "note": "launch duration = HIP events attached to the kernel's dispatch on its launch stream inside the library, averaged over "
                             "the timed region; launches of different host threads overlap on the device (launch_overlap = summed launch durations / "
                             "step time), so a launch's duration is the time it spends SHARING the GPU: aggregate_GBps = algorithmic bytes of a step / "
                             "step time is the figure to cross-check ms_per_step with.  The rescue batches go through the device's submission ring (one resident "
                             "kernel per epoch, `swalign2_resident`; `swalign2` is a batch and its duration the batch's span on the device clock): a batch is "
                             "copied into HBM by the copy engine before its descriptor is published, the workers read HBM and write their results into the "
                             "caller's pinned block over PCIe.  The events see "
                             "about 0.06 ms of dispatch latency per launch that a kernel trace does not.  `extend` is what an extension call launches back "
                             "to back on its stream -- the sift kernel (shortcuts, one task per lane), the short extension kernel and, only behind a launch that "
                             "deferred a task, the full kernel -- timed from the first dispatch to the end of the last; its avg_launch_ms_kernel_trace is the sum of those "
                             "kernels' durations per call, the gaps between them (waiting for wave slots) not included.  avg_launch_ms_kernel_trace is the rocprofv3 "
                             f"--kernel-trace --stats average of this command committed under profiles/{PROFILE_TAG}_kernel_stats_bench.csv.  These kernels are "
                             "integer DP with hundreds of operations per byte: the HBM fraction is ~1e-3 by construction (SURVEY.md 8d), what binds is the "
                             "vector pipe (frac_of_issue_rate)"},
        "gcups": gcups, "frac_of_valu_ceiling": valu, "frac_of_issue_rate": issue_frac,
        "protocol": {"id": PROTOCOL, "passes_per_step": passes, "barrier_between_passes": False, "feeder": "bpsw_feeder_run_repeats (next free thread takes the next call)",
                     "profile_figures": f"per-launch instruction / traffic / trace figures come from the COMMITTED profiles/{PROFILE_TAG}_* of this command, not from this run",
                     "comparable_with": "BENCH_r03 (same protocol); BENCH_r01 / r02 used 3 resp. 8 passes per step with a barrier behind every pass and priced the "
                                        "VALU ceiling at SIMD-16: their `value` is comparable only through ms per read, their fractions are not"},
        "kernels": {"extend": {"avg_ms": round(ext_avg_ms, 4), "launches": ext_launches, "bytes_per_launch": int(ext_bytes),
                               "h2d_ms_avg": round(st["ext_h2d_ms"] / max(ext_launches, 1), 4), "d2h_ms_avg": round(st["ext_d2h_ms"] / max(ext_launches, 1), 4)},
                    "swalign2": {"avg_ms": round(sw_avg_ms, 4), "launches": sw_launches, "bytes_per_launch": int(sw_bytes),
                                 "jobs": int(st["sw_jobs"]), "replay_rounds": int(st["sw_replayed_rounds"]), "wasted_jobs": int(st["sw_wasted"]),
                                 "h2d_ms_avg": round(st["sw_h2d_ms"] / max(sw_launches, 1), 4), "d2h_ms_avg": round(st["sw_d2h_ms"] / max(sw_launches, 1), 4)}},
        "host": {"cpus_busy": round(cpu_busy, 2), "cpu_quota": cpu_quota(), "call_ms": host_ms, "reads_per_s_vs_cpus": cpu_sweep,
                 # CPU time of the calling thread inside a call (CLOCK_THREAD_CPUTIME_ID around the last timed call on every item): what a
                 # read costs the executor's CPU quota on this path
                 "call_cpu_ms": {k: round(float(np.mean(v)), 4) if v else None for k, v in call_cpu_ms.items()},
                 "cpu_s_per_Mreads": round(cpu_busy * elapsed / (reads_per_step * args.steps / 1e6), 5),
                 "dram_bytes_per_read_est": round((passes * (3 * sum(int(w.size) for w in wires) + 2 * 20 * sum(ntasks)) + 3 * (pcie_bytes_per_step - passes * (sum(int(w.size) for w in wires) + 20 * sum(ntasks)))) / max(reads_per_step, 1), 1),
                 "phase_ms_per_call": {"extend": {k: round(st["ext_" + k + "_ms"] / max(ext_launches, 1), 4) for k in ("host_in", "wait", "dev", "host_out")},
                                       "matesw_group": {k: round(st["grp_" + k + "_ms"] / max(int(st["grp_calls"]), 1), 4)
                                                        for k in ("plan", "pack", "wait", "dev", "replay", "out")}}, "pcie_bytes_per_step": int(pcie_bytes_per_step),
                 "pcie_GBps": round(pcie_bytes_per_step * args.steps / elapsed / 1e9, 3),
                 "dram_bytes_per_step_est": int(passes * (3 * sum(int(w.size) for w in wires) + 2 * 20 * sum(ntasks)) + 3 * (pcie_bytes_per_step - passes * (sum(int(w.size) for w in wires) + 20 * sum(ntasks)))),
                 "dram_note": "estimate of host DRAM traffic per step: a wire batch is read by the staging memcpy, written to the pinned block and read by the "
                              "copy engine (3 touches; the JNI shim now writes it into the pinned block directly, 2 touches); results are written over PCIe and "
                              "read once; the rescue inputs that travel are read, written to pinned staging and read over PCIe"},
        "breakdown": extras,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(W, args.config, rank, xtra)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    F.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
