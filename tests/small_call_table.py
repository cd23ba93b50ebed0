#!/usr/bin/env python3
"""At which call size does the GPU path win?  One calling thread, the HOST-BUFFER entry points the two JNI symbols call, at the
sizes the reference's flags produce (-FPGASWExtThreshold 64 / -bSWExtSize, run_test.sh:7; -sbatch 10,
commandline/BWAMEMCommand.scala:28,43), against the reference's own C on ONE host core for the same inputs (oracle/_ref:
ksw_extend2 under the builder's batch loop, mem_group_matesw).  A small call is latency-bound on the GPU -- one SW job is a serial
chain of ~0.1 ms for a single wavefront -- so the table tells a maintainer the smallest batch at which the plug-in pays
(INTEGRATION.md).  Test infrastructure (it times the oracle's reference build); usage on a GPU box: python tests/small_call_table.py"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "cloud-scale-bwamem_amd"), os.path.join(ROOT, "oracle"), ROOT):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402
from bpsw_hip import synth  # noqa: E402


def timed(fn, items, reps):
    """mean ms per call.  (The interpreter's cyclic collector is held off while the clock runs: one collection of the harness's own
    garbage -- 30-50 ms with the synthetic batches alive -- inside a few hundred 0.1 ms calls doubled the first row of round 5's
    first tables.)"""
    import gc
    for it in items:
        fn(it)
    gc.collect()
    gc.disable()
    try:
        t0 = time.perf_counter()
        for _ in range(reps):
            for it in items:
                fn(it)
        return 1e3 * (time.perf_counter() - t0) / (reps * len(items))
    finally:
        gc.enable()


def main():
    ctx = bpsw_hip.Context(0)
    opt = bpsw_hip.default_opt()
    ref = po.Ref() if po.Ref.available() else None
    ropt = po.Oracle().default_opt()
    mat = po.default_mat()
    out = {"extend": [], "matesw_group": []}
    for n in (64, 256, 1024, 4096, 32768):
        k = 16 if n <= 1024 else 4
        soas = [synth.ext_tasks(int(n * 1.08), read_len=150, seed=synth.CONFIG_SEED_BASE + 3 + 31 * j) for j in range(k)]
        soas = [s.subset(np.arange(min(n, s.n))) for s in soas]
        wires = [bpsw_hip.wire_pack(s) for s in soas]
        gpu = timed(lambda w: ctx.extend_batch(w), wires, 20 if n <= 1024 else 6)
        cpu = timed(lambda s: ref.extend_batch(s, mat), soas, 3 if n <= 1024 else 1) if ref else None
        out["extend"].append({"tasks_per_call": int(np.mean([s.n for s in soas])), "gpu_ms_per_call": round(gpu, 4),
                              "gpu_us_per_task": round(1e3 * gpu / n, 3), "reference_c_one_core_ms": round(cpu, 4) if cpu else None,
                              "gpu_speedup_vs_one_core": round(cpu / gpu, 2) if cpu else None})
    for pairs in (10, 64, 256, 1024, 4096):
        k = 40 if pairs <= 64 else (8 if pairs <= 1024 else 3)
        groups = [synth.rescue_group_fast(pairs, seed=synth.CONFIG_SEED_BASE + 3 + 17 * j, p_resc=0.10) for j in range(k)]
        s0 = ctx.stats()
        gpu = timed(lambda g: ctx.matesw_group(opt, g), groups, 10 if pairs <= 256 else 4)
        s1 = ctx.stats()
        cpu = timed(lambda g: ref.matesw_group(ropt, g), groups, 2 if pairs <= 256 else 1) if ref else None
        out["matesw_group"].append({"pairs_per_call": pairs, "sw_jobs_per_call": round((s1.sw_jobs - s0.sw_jobs) / max(s1.grp_calls - s0.grp_calls, 1), 2),
                                    "gpu_ms_per_call": round(gpu, 4), "gpu_us_per_pair": round(1e3 * gpu / pairs, 3),
                                    "reference_c_one_core_ms": round(cpu, 4) if cpu else None,
                                    "gpu_speedup_vs_one_core": round(cpu / gpu, 2) if cpu else None})
    # Concurrent callers (round 5): T task threads, each making blocking calls of the default group size (-sbatch 10) or 64 pairs, through
    # the native feeder (csrc/bpsw_feeder.cpp: one context per thread, next free thread takes the next group).  With the submission ring the
    # calls of all threads are descriptors of one resident kernel; BPSW_RING=0 gives every call a launch of its own.
    from bpsw_hip import feeder as fd
    out["matesw_group_concurrent"] = []
    for pairs in (10, 64):
        groups = [synth.rescue_group_fast(pairs, seed=synth.CONFIG_SEED_BASE + 5 + 13 * j, p_resc=0.10) for j in range(256)]
        structs = [g.as_struct() for g in groups]
        cnts = [np.zeros(2 * g.group_size, np.int32) for g in groups]
        regs = [np.empty(int(g.regs.shape[0] + g.ref_rb.shape[0] + 16), bpsw_hip.ALNREG_DTYPE) for g in groups]
        items, _ = fd.make_items([], [], groups, structs, cnts, regs)
        cpu = timed(lambda g: ref.matesw_group(ropt, g), groups[:64], 2) if ref else None
        for T in (1, 4, 16):
            F = fd.Feeder(T, 0, opt, bpsw_hip.RESCUE_C)
            F.run(items, 2)
            F.reset_stats()
            t0 = time.perf_counter()
            reps = 8
            F.run(items, reps)
            dt = time.perf_counter() - t0
            calls = reps * len(items)
            st = F.stats_sum()
            out["matesw_group_concurrent"].append({
                "pairs_per_call": pairs, "callers": T, "sw_jobs_per_call": round(st["sw_jobs"] / max(st["grp_calls"], 1), 2),
                "ms_per_call_as_a_caller_sees_it": round(float(np.mean([it.ms for it in items])), 4),
                "calls_per_s_all_callers": round(calls / dt, 1), "us_per_call_aggregate": round(1e6 * dt / calls, 2),
                "reference_c_one_core_ms": round(cpu, 4) if cpu else None,
                "per_caller_rate_vs_reference_core": round(cpu / float(np.mean([it.ms for it in items])), 3) if cpu else None,
                "ring_batches": int(st.get("sw_ring_calls", 0))})
            F.close()
    # ... and the same for small EXTENSION calls (the extension ring, round 5; BPSW_EXT_RING=0: a launch per call)
    out["extend_concurrent"] = []
    for n in (64, 128, 256):
        soas = [synth.ext_tasks(int(n * 1.08) + 2, read_len=150, seed=synth.CONFIG_SEED_BASE + 9 + 7 * j) for j in range(128)]
        soas = [s_.subset(np.arange(min(n, s_.n))) for s_ in soas]
        wires = [bpsw_hip.wire_pack(s_) for s_ in soas]
        outs = [np.zeros(10 * s_.n, np.int16) for s_ in soas]
        items, _ = fd.make_items(wires, outs, [], [], [], [])
        cpu = timed(lambda s_: ref.extend_batch(s_, mat), soas[:32], 1) if ref else None
        for T in (1, 4, 16):
            F = fd.Feeder(T, 0, opt, bpsw_hip.RESCUE_C)
            F.run(items, 2)
            F.reset_stats()
            t0 = time.perf_counter()
            reps = 8
            F.run(items, reps)
            dt = time.perf_counter() - t0
            calls = reps * len(items)
            st = F.stats_sum()
            out["extend_concurrent"].append({
                "tasks_per_call": int(np.mean([s_.n for s_ in soas])), "callers": T,
                "ms_per_call_as_a_caller_sees_it": round(float(np.mean([it.ms for it in items])), 4),
                "calls_per_s_all_callers": round(calls / dt, 1), "reference_c_one_core_ms": round(cpu, 4) if cpu else None,
                "per_caller_rate_vs_reference_core": round(cpu / float(np.mean([it.ms for it in items])), 2) if cpu else None,
                "ring_batches": int(st.get("ext_ring_calls", 0))})
            F.close()
    # ... and a MIXED stream on one context, as a task thread of memChainToAlnBatched makes it: a first-round batch of thousands of tasks
    # (a launch, ~1 ms) then later rounds of a few dozen (the extension ring, ~0.085 ms).  Round 5 kept ONE running estimate of the
    # device-phase wait for both and napped 0.7 x ~1 ms before its first look at a ring call's completion record (advisor, round 5);
    # since round 6 the ring waits have estimates of their own (bpsw_ctx::wait_est_ms[4..5]).
    big = bpsw_hip.wire_pack(synth.ext_tasks(4400, read_len=150, seed=synth.CONFIG_SEED_BASE + 77))
    smalls = []
    for j in range(24):
        s_ = synth.ext_tasks(72, read_len=150, seed=synth.CONFIG_SEED_BASE + 78 + j)
        smalls.append(bpsw_hip.wire_pack(s_.subset(np.arange(min(64, s_.n)))))
    for w in [big] + smalls:
        ctx.extend_batch(w)
    import gc
    gc.collect(); gc.disable()
    try:
        t_small_alone = timed(lambda w: ctx.extend_batch(w), smalls, 10)
        t_small_mixed, t_big_mixed, n_small = 0.0, 0.0, 0
        for _ in range(10):
            t0 = time.perf_counter(); ctx.extend_batch(big); t_big_mixed += time.perf_counter() - t0
            for w in smalls[:8]:
                t0 = time.perf_counter(); ctx.extend_batch(w); t_small_mixed += time.perf_counter() - t0; n_small += 1
    finally:
        gc.enable()
    out["extend_mixed_stream"] = {"small_call_tasks": 64, "big_call_tasks": 4400, "small_ms_in_a_stream_of_small_calls": round(t_small_alone, 4),
                                  "small_ms_behind_a_big_call_on_the_same_context": round(1e3 * t_small_mixed / n_small, 4),
                                  "big_ms": round(1e3 * t_big_mixed / 10, 4),
                                  "note": "one context, one thread: ten rounds of one 4 400-task call followed by eight 64-task calls"}
    out["note"] = ("one calling thread, distinct inputs per call, host buffers in and out; reference C = oracle/_ref (the reference's own ksw_extend2 / "
                   "mem_group_matesw with SSE2 ksw_align2) on one host core of the GPU box; p_resc = 10 % of the pairs need rescue (configs[2])")
    print(json.dumps(out, indent=1))
    print("\n| boundary | call size | GPU ms / call | reference C, one core, ms | GPU / one core |\n|---|---|---|---|---|")
    for r in out["extend"]:
        print(f"| swExtendFPGAJNI | {r['tasks_per_call']} tasks | {r['gpu_ms_per_call']} | {r['reference_c_one_core_ms']} | {r['gpu_speedup_vs_one_core']}x |")
    for r in out["matesw_group"]:
        print(f"| mateSWJNI | {r['pairs_per_call']} pairs ({r['sw_jobs_per_call']} SW jobs) | {r['gpu_ms_per_call']} | {r['reference_c_one_core_ms']} | {r['gpu_speedup_vs_one_core']}x |")
    print("\n| callers | pairs per call | ms per call (caller's view) | calls/s, all callers | reference C, one core, ms | per caller vs one core |\n|---|---|---|---|---|---|")
    for r in out["matesw_group_concurrent"]:
        print(f"| {r['callers']} | {r['pairs_per_call']} ({r['sw_jobs_per_call']} SW jobs) | {r['ms_per_call_as_a_caller_sees_it']} | {r['calls_per_s_all_callers']} | "
              f"{r['reference_c_one_core_ms']} | {r['per_caller_rate_vs_reference_core']}x |")

    print("\n| callers | tasks per call | ms per call (caller's view) | calls/s, all callers | reference C, one core, ms | per caller vs one core | through the ring |\n|---|---|---|---|---|---|---|")
    for r in out["extend_concurrent"]:
        print(f"| {r['callers']} | {r['tasks_per_call']} | {r['ms_per_call_as_a_caller_sees_it']} | {r['calls_per_s_all_callers']} | {r['reference_c_one_core_ms']} | "
              f"{r['per_caller_rate_vs_reference_core']}x | {r['ring_batches']} |")

    m = out["extend_mixed_stream"]
    print(f"\nmixed stream on one context: a 64-task call takes {m['small_ms_behind_a_big_call_on_the_same_context']} ms behind {m['big_call_tasks']}-task calls "
          f"({m['big_ms']} ms each), {m['small_ms_in_a_stream_of_small_calls']} ms in a stream of its own kind")


if __name__ == "__main__":
    main()
