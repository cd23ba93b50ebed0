#!/usr/bin/env python3
"""bench.py -- pair-end 2x150 bp reads aligned/sec through the MI355X batched Smith-Waterman path.

One step = one pass of the hot path over one synthetic batch of PAIRS_PER_STEP read pairs (BASELINE.json
configs[2] shape: 2x150 bp, 1 % substitutions, 0.1 % indels, 10 % of pairs need mate rescue):
  * boundary 2: the extension wire batches of those reads (<= 32768 reads per batch, as the reference's
    run_test.sh uses -bSWExtSize 32768), device-resident, through bpsw_extend_batch_device;
  * boundary 1: the SWAlign2 rescue jobs of those pairs, device-resident, through bpsw_swalign2_batch_device.
Inputs are already in HBM when the timed region starts.  N>1: one process per GPU (torch.distributed over
RCCL for the barrier and the max-over-ranks only; the path has no data-path collective), each rank
aligns its own shard of pairs ("weak" scaling: Spark partition -> device).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4), round robin; streams that
# share a queue run one after the other.  A step's five batches each have their own stream, so with four queues two of them
# were serialised (the kernel trace showed the rescue kernel and one extension launch on the same queue).  Must be set
# before the runtime initialises; an executor that runs several task threads wants the same.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "cloud-scale-bwamem_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

READ_LEN = 150
READS_PER_EXT_BATCH = 32768       # reference: -bSWExtSize 32768 (run_test.sh:7); idx travels as int16
EXT_BATCHES_PER_STEP = 4
PAIRS_PER_STEP = READS_PER_EXT_BATCH * EXT_BATCHES_PER_STEP // 2   # 65536 pairs = 131072 reads
RESCUE_JOBS_PER_PAIR = 0.11       # p_resc = 10 % (+ multi-anchor pairs), SURVEY.md 8(d) config 3
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_inputs(rank: int):
    from bpsw_hip import synth, wire_pack
    seed0 = synth.CONFIG_SEED_BASE + 3 + 1000 * rank
    wires, ntasks, soas = [], [], []
    for b in range(EXT_BATCHES_PER_STEP):
        soa = synth.ext_tasks(READS_PER_EXT_BATCH, read_len=READ_LEN, sub_rate=0.01, indel_rate=0.001, n_rate=0.001,
                              seed=seed0 + b)
        wires.append(wire_pack(soa))
        soas.append(soa)
        ntasks.append(soa.n)
    n_jobs = int(PAIRS_PER_STEP * RESCUE_JOBS_PER_PAIR)
    jobs = synth.sw_jobs(n_jobs, read_len=READ_LEN, win_min=400, win_max=400, sub_rate=0.02, indel_rate=0.002,
                         unrelated_frac=0.05, decoy_frac=0.1, rev_frac=1.0, seed=seed0 + 100)
    return soas, wires, ntasks, jobs


def cpu_baseline(soas, wires, ntasks, jobs, xtra):
    """The reference's CPU path timed on this box's host cores, on a bounded sample of the same step.

    kind "reference": oracle/_ref/libbwaref.so (the reference's own C built in place by oracle/Makefile): ksw_extend2 under
    the extension() control for boundary 2 and the SSE2 ksw_align2 that jniNative.so runs for boundary 1, one whole step
    per thread on every host core (the reference is one Spark task thread per core).  kind "port": the scalar oracle
    (oracle/bpsw_oracle.c), one thread, when the reference build did not travel with the snapshot."""
    import pyoracle as po
    n_jobs = len(jobs["q_len"])
    if os.path.exists(po.REF_SO):
        from concurrent.futures import ThreadPoolExecutor
        ref = po.Ref()
        mat = po.default_mat()
        cores = max(1, min(len(os.sched_getaffinity(0)), 64))

        def one_step(_):
            for soa in soas:
                ref.extend_batch(soa, mat)
            ref.align2_batch(mat, 6, 1, 6, 1, xtra, **jobs)

        one = time.perf_counter()
        one_step(0)
        one = time.perf_counter() - one
        reps = max(1, int(round(3.0 / max(one, 1e-3))))          # ~3 s of wall per thread, ~3 s x cores of CPU work
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:                     # ctypes releases the GIL inside the C loops
            list(ex.map(one_step, range(cores * reps)))
        dt = time.perf_counter() - t0
        return {
            "value": round(cores * reps * 2 * PAIRS_PER_STEP / dt, 1), "unit": "reads/s", "cores": cores, "kind": "reference",
            "sample": f"{cores * reps} whole steps ({sum(ntasks)} extension tasks + {n_jobs} rescue jobs each) in {dt:.2f}s on "
                      f"{cores} threads; reference ksw_extend2 (scalar) + ksw_align2 (SSE2) from oracle/_ref; "
                      f"1 thread alone: {2 * PAIRS_PER_STEP / one:.0f} reads/s",
        }
    orc = po.Oracle()
    t0 = time.perf_counter()
    _, cells_ext = orc.wire_extend(wires[0])
    t_ext = time.perf_counter() - t0
    ns = min(1500, n_jobs)
    sub = dict(jobs)
    for k in ("q_len", "t_len", "q_off", "t_off", "q_rev"):
        sub[k] = jobs[k][:ns]
    t0 = time.perf_counter()
    _, cells_sw = orc.sw_align2_jobs(orc.default_opt(), xtra, **sub)
    t_sw = time.perf_counter() - t0
    # seconds of CPU per read of the step = extension share + rescue share
    sec_per_read = t_ext / READS_PER_EXT_BATCH + (t_sw / ns) * n_jobs / (2.0 * PAIRS_PER_STEP)
    return {
        "value": round(1.0 / sec_per_read, 1), "unit": "reads/s", "cores": 1, "kind": "port",
        "sample": f"{ntasks[0]} extension tasks ({READS_PER_EXT_BATCH} reads) in {t_ext:.2f}s + {ns} SWAlign2 jobs in {t_sw:.2f}s, "
                  f"oracle/bpsw_oracle.c single thread; {cells_ext / t_ext / 1e9:.3f} / {cells_sw / t_sw / 1e9:.3f} GCUPS",
    }


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
            "sam_bytes": int(sum(len(t) for t in texts)), "note": "host buffers in, SAM text out; not part of `value`"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tail", action="store_true", help="skip the worker2-tail breakdown entry")
    args = ap.parse_args()

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
    import ctypes as C
    # One context per extension batch of the step plus one for the rescue jobs, all on this rank's GPU: each context
    # owns a stream, so the independent batches of a step overlap on the device the way concurrent Spark task
    # threads of one executor overlap their JNI calls (the reference native code is re-entrant for that reason).
    # BENCH_STEPS_IN_FLIGHT=n (default 1) rotates n such sets over the steps: the entries are asynchronous, so the host can
    # submit step k+1 while the device still works on step k; a set is reused only after its previous step has been waited
    # for, and the timed region ends with every step complete.  Measured on MI355X: 1.451 / 1.430 / 1.420 ms per step for
    # n = 1 / 2 / 3 -- the device is already busy throughout a step, so the default stays 1 and the per-launch durations
    # the roofline entry uses are not stretched by overlap between steps.
    STEPS_IN_FLIGHT = max(1, int(os.environ.get("BENCH_STEPS_IN_FLIGHT", "1")))
    SW_FIRST = os.environ.get("BENCH_SW_FIRST", "0") == "1"   # issue order of the step's five independent batches
    sets = [[bpsw_hip.Context(local_rank) for _ in range(EXT_BATCHES_PER_STEP + 1)]  # no fallback: raises without a gfx950 device
            for _ in range(STEPS_IN_FLIGHT)]
    ctxs = sets[0]
    ctx = ctxs[-1]
    opt = bpsw_hip.default_opt()
    xtra = bpsw_hip.KSW_XSUBO | bpsw_hip.KSW_XSTART | bpsw_hip.KSW_XBYTE | 19   # MemSamPe.scala:1187-1189

    soas, wires, ntasks, jobs = build_inputs(rank)
    # ---- make everything resident in HBM before the timed region ---------------------------------
    d_wires = [torch.from_numpy(w).to(dev) for w in wires]
    d_outs_all = [[torch.zeros(10 * n, dtype=torch.int16, device=dev) for n in ntasks] for _ in sets]
    d_outs = d_outs_all[0]
    d_jobs = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in jobs.items()}
    n_jobs = int(jobs["q_len"].shape[0])
    d_sw_out_all = [torch.zeros((n_jobs, 7), dtype=torch.int32, device=dev) for _ in sets]
    d_sw_out = d_sw_out_all[0]
    sj = bpsw_hip.SwJobs()
    sj.n, sj.xtra = n_jobs, xtra
    for k in ("q_len", "t_len", "q_off", "t_off", "q_rev", "q_pool", "t_pool"):
        setattr(sj, k, d_jobs[k].data_ptr())
    sj.q_pool_bytes, sj.t_pool_bytes = d_jobs["q_pool"].numel(), d_jobs["t_pool"].numel()
    torch.cuda.synchronize(dev)

    ext_ms_sum, sw_ms_sum, ext_launches, sw_launches = 0.0, 0.0, 0, 0
    busy = [False] * len(sets)      # set has a submitted step that has not been waited for
    counted = [False] * len(sets)   # ... and that step belongs to the timed region
    turn = 0

    def collect(k: int):
        """wait for the step submitted on set k; HIP events on the launch streams, recorded inside the library"""
        nonlocal ext_ms_sum, sw_ms_sum, ext_launches, sw_launches
        if not busy[k]:
            return
        for cx in sets[k][:-1]:
            e, _ = cx.last_kernel_ms()
            if counted[k]:
                ext_ms_sum += e
                ext_launches += 1
        _, s_ms = sets[k][-1].last_kernel_ms()
        if counted[k]:
            sw_ms_sum += s_ms
            sw_launches += 1
        busy[k] = False

    def step(timed: bool):
        nonlocal turn
        k = turn % len(sets)
        turn += 1
        collect(k)                  # the step this set ran STEPS_IN_FLIGHT steps ago
        if SW_FIRST:
            sets[k][-1].swalign2_batch_device(opt, sj, d_sw_out_all[k].data_ptr(), 0)
        for cx, w, n, dw, do in zip(sets[k], wires, ntasks, d_wires, d_outs_all[k]):
            cx.extend_batch_device(dw.data_ptr(), int(w.size), n, do.data_ptr(), 0)   # asynchronous, context's own stream
        if not SW_FIRST:
            sets[k][-1].swalign2_batch_device(opt, sj, d_sw_out_all[k].data_ptr(), 0)
        busy[k], counted[k] = True, timed

    def drain():
        for k in range(len(sets)):
            collect(k)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    drain()                         # every one of the K steps complete, results in HBM
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    reads_total = 2 * PAIRS_PER_STEP * args.steps * world
    value = reads_total / elapsed

    # ---- the two boundaries on their own (outside the timed region of `value`) ------
    def time_only(fn, reps):
        fn()
        barrier()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        return (time.perf_counter() - t) / reps

    def ext_only():
        for cx, w, n, dw, do in zip(ctxs, wires, ntasks, d_wires, d_outs):
            cx.extend_batch_device(dw.data_ptr(), int(w.size), n, do.data_ptr(), 0)

    def sw_only():
        ctx.swalign2_batch_device(opt, sj, d_sw_out.data_ptr(), 0)

    reps = max(3, min(args.steps, 10))
    t_ext_only, t_sw_only = time_only(ext_only, reps), time_only(sw_only, reps)

    # ---- what an executor with ten busy task threads would see: two steps in flight (outside `value`) -----------------
    two_in_flight = None
    if STEPS_IN_FLIGHT == 1:
        try:
            extra = [bpsw_hip.Context(local_rank) for _ in range(EXT_BATCHES_PER_STEP + 1)]
            sets2 = [sets[0], extra]
            outs2 = [d_outs_all[0], [torch.zeros(10 * n, dtype=torch.int16, device=dev) for n in ntasks]]
            sw2 = [d_sw_out_all[0], torch.zeros((n_jobs, 7), dtype=torch.int32, device=dev)]

            def submit(k):
                for cx, w, n, dw, do in zip(sets2[k], wires, ntasks, d_wires, outs2[k]):
                    cx.extend_batch_device(dw.data_ptr(), int(w.size), n, do.data_ptr(), 0)
                sets2[k][-1].swalign2_batch_device(opt, sj, sw2[k].data_ptr(), 0)

            def wait(k):
                for cx in sets2[k]:
                    cx.last_kernel_ms()

            for k in (0, 1):
                submit(k)
            for k in (0, 1):
                wait(k)
            barrier()
            t = time.perf_counter()
            n2 = 2 * reps
            submit(0)
            for i in range(1, n2):
                submit(i & 1)          # the entry itself waits for the step this set ran two steps ago
            for k in (0, 1):
                wait(k)
            barrier()
            two_in_flight = round(2 * PAIRS_PER_STEP * n2 / (time.perf_counter() - t), 1)
            del extra
        except Exception as e:  # noqa: BLE001 -- an extra line must never cost the bench its JSON
            two_in_flight = repr(e)

    # ---- worker2's tail, outside `value`: host-inclusive rate of one calling thread + kernel rate ----
    tail = None
    if rank == 0 and world == 1 and not args.no_tail:   # single-GPU runs only: the scaling runs must not make the other ranks wait
        try:
            tail = tail_breakdown(ctx, opt)
        except Exception as e:  # noqa: BLE001 -- a breakdown line must never cost the bench its JSON
            tail = {"error": repr(e)}

    # ---- roofline of the dominant kernel (algorithmic bytes: DESIGN.md) -------------
    ext_bytes = sum(int(w.size) + 20 * n for w, n in zip(wires, ntasks)) / len(wires)          # per launch
    sw_bytes = float(jobs["q_len"].sum() + jobs["t_len"].sum() + 28 * n_jobs)                  # per launch
    ext_avg_ms = ext_ms_sum / max(ext_launches, 1)
    sw_avg_ms = sw_ms_sum / max(sw_launches, 1)
    dominant = "extend" if ext_ms_sum >= sw_ms_sum else "swalign2"
    dom_bytes, dom_ms = (ext_bytes, ext_avg_ms) if dominant == "extend" else (sw_bytes, sw_avg_ms)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written from rocprofv3 --pmc passes
    if os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path)).get(dominant)
        except Exception:
            traffic = None

    # What actually binds these kernels: instruction issue.  Wave-instructions per step from the committed
    # PMC pass, over the live step time, against the measured issue rates of one SIMD (tools/ubench/valu_rate.hip).
    issue = None
    try:
        pi = json.load(open(os.path.join(ROOT, "profiles", "pmc_issue.json")))
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        per_step = EXT_BATCHES_PER_STEP * (pi["extend"]["valu"] + pi["extend"]["salu"] + pi["ext_prepass"]["valu"] + pi["ext_prepass"]["salu"]) \
            + pi["swalign2"]["valu"] + pi["swalign2"]["salu"] + pi["sw_prepass"]["valu"] + pi["sw_prepass"]["salu"]
        ns = 1e9 * (elapsed / args.steps) / (per_step / (4.0 * n_cu))
        issue = {"wave_insts_per_step": int(per_step), "simds": 4 * n_cu, "ns_per_inst_per_simd": round(ns, 3),
                 "measured_issue_ns": {"one_kind_stream": 1.8, "mixed_valu_salu_streams": 1.15},
                 "frac_of_mixed_issue_rate": round(1.15 / ns, 3),
                 "note": "instruction counts: profiles/pmc_issue.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU, same command); "
                         "issue rates: profiles/r01g_valu_rate.txt, profiles/r01_microbench_issue.txt"}
    except Exception:
        issue = None

    out = {
        "metric": "pair-end 2x150bp reads aligned/sec",
        "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "configs[2]: pair-end 2x150bp synthetic reads (1% sub, 0.1% indel), batched seed "
                               "extension + batched pair-end SW rescue (10% of pairs), 1 MI355X per rank",
                   "pairs_per_step_per_gpu": PAIRS_PER_STEP, "ext_tasks_per_step": int(sum(ntasks)),
                   "rescue_jobs_per_step": n_jobs, "steps_in_flight": STEPS_IN_FLIGHT,
                   "hip_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                   "parallelism": f"partition->device x{world} (no collective)"},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                     "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 4)},
        "issue": issue,
        "kernels": {"extend": {"avg_ms": round(ext_avg_ms, 4), "launches": ext_launches, "bytes_per_launch": int(ext_bytes)},
                    "swalign2": {"avg_ms": round(sw_avg_ms, 4), "launches": sw_launches, "bytes_per_launch": int(sw_bytes)}},
        "breakdown": {"extend_only_reads_per_s": round(2 * PAIRS_PER_STEP / t_ext_only, 1), "extend_only_ms_per_step": round(1e3 * t_ext_only, 3),
                      "rescue_only_jobs_per_s": round(n_jobs / t_sw_only, 1), "rescue_only_ms_per_step": round(1e3 * t_sw_only, 3),
                      "two_steps_in_flight_reads_per_s": two_in_flight,
                      "note": "this rank only; same resident inputs, each boundary alone (SURVEY.md 8d i/ii); `value` is (iii) combined; "
                              "two_steps_in_flight: the same step with a second set of contexts one step ahead (ten batches in flight)",
                      "worker2_tail": tail},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(soas, wires, ntasks, jobs, xtra)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    for cx in ctxs:
        cx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
