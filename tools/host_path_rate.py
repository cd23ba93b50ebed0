#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points (what the JNI shim calls): never the bench `value`,
reported in DESIGN.md.  Usage on a GPU box:  python tools/host_path_rate.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd"))
import bpsw_hip  # noqa: E402
from bpsw_hip import synth  # noqa: E402

ctx = bpsw_hip.Context(0)
out = {}
soa = synth.ext_tasks(32768, read_len=150, seed=synth.CONFIG_SEED_BASE + 3)
wire = bpsw_hip.wire_pack(soa)
for _ in range(3):
    ctx.extend_batch(wire)
s0 = ctx.stats()
t0 = time.perf_counter()
R = 20
for _ in range(R):
    ctx.extend_batch(wire)
dt = time.perf_counter() - t0
s1 = ctx.stats()
out["extend_host_entry"] = {"reads_per_s": 32768 * R / dt, "tasks": soa.n, "wire_bytes": int(wire.size), "ms_per_call": 1e3 * dt / R,
                            "h2d_ms": (s1.ext_h2d_ms - s0.ext_h2d_ms) / R, "kernel_ms": (s1.ext_kernel_ms - s0.ext_kernel_ms) / R,
                            "d2h_ms": (s1.ext_d2h_ms - s0.ext_d2h_ms) / R}
# the same tasks as a byte batch and as a coordinate batch (wire format 2: target flanks read from the device-resident reference)
l_pac = 4_000_037
pac, bases = synth.random_pac(l_pac, seed=91)
chains = synth.read_chains(16384, bases, l_pac, read_len=150, sub_rate=0.01, indel_rate=0.001, seed=92, p_subseed=0.0, p_shifted=0.0, p_decoy=0.0)
co, by = synth.coord_ext_tasks(chains, bases, seed=93)
ctx.ref_load(pac, l_pac)
for name, w in (("extend_bytes_batch", bpsw_hip.wire_pack(by)), ("extend_coordinate_batch", bpsw_hip.wire_coords_pack(co))):
    for _ in range(3):
        ctx.extend_batch(w)
    s0 = ctx.stats()
    t0 = time.perf_counter()
    for _ in range(R):
        ctx.extend_batch(w)
    dt = time.perf_counter() - t0
    s1 = ctx.stats()
    out[name] = {"tasks": co.n, "wire_bytes": int(w.size), "bytes_per_task": round(w.size / co.n, 1), "ms_per_call": 1e3 * dt / R,
                 "h2d_ms": (s1.ext_h2d_ms - s0.ext_h2d_ms) / R, "kernel_ms": (s1.ext_kernel_ms - s0.ext_kernel_ms) / R}
for pairs in (10, 256, 4096, 16384):
    # -sbatch 10 is the reference default (commandline/BWAMEMCommand.scala:28): most calls then carry 0-2 SW jobs,
    # so average over many different groups
    ngroups = 40 if pairs <= 256 else 1
    groups = [synth.rescue_group_fast(pairs, seed=synth.CONFIG_SEED_BASE + 3 + 17 * k, p_resc=0.10) for k in range(max(ngroups, 8))]
    ngroups = len(groups)   # several distinct groups per size: a call never finds its inputs warm in the host caches
    opt = bpsw_hip.default_opt()
    for g in groups:
        ctx.matesw_group(opt, g)
    s0 = ctx.stats()
    R = 5
    t0 = time.perf_counter()
    for _ in range(R):
        for g in groups:
            ctx.matesw_group(opt, g)
    dt = time.perf_counter() - t0
    s1 = ctx.stats()
    calls = R * ngroups
    out[f"matesw_group_{pairs}_pairs"] = {"pairs_per_s": pairs * calls / dt, "ms_per_call": 1e3 * dt / calls,
                                          "sw_jobs_per_call": (s1.sw_jobs - s0.sw_jobs) / calls, "kernel_ms": (s1.sw_kernel_ms - s0.sw_kernel_ms) / calls,
                                          "h2d_ms": (s1.sw_h2d_ms - s0.sw_h2d_ms) / calls, "replay_rounds": int(s1.sw_replayed_rounds - s0.sw_replayed_rounds),
                                          "host_phases_ms": {k: round((getattr(s1, "grp_" + k + "_ms") - getattr(s0, "grp_" + k + "_ms")) / calls, 4)
                                                             for k in ("plan", "pack", "wait", "dev", "replay", "out")},
                                          "wasted_jobs": int(s1.sw_wasted - s0.sw_wasted)}
print(json.dumps(out, indent=1))
