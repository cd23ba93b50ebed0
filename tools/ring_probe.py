"""Lone-call timings of bpsw_swalign2_batch through the submission ring (or, with BPSW_RING=0, through a launch of its own):
wall time per call and the device span the library reports, for batches from one job pair to a saturating launch."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bpsw_hip
from bpsw_hip import synth
import pyoracle as po
XTRA = po.KSW_XSUBO | po.KSW_XSTART | po.KSW_XBYTE | 19
ctx = bpsw_hip.Context(0)
opt = bpsw_hip.default_opt()
print("ring", os.environ.get("BPSW_RING", "1"), "wg_per_cu", os.environ.get("BPSW_RING_WG_PER_CU"))
for n in (2, 16, 64, 428, 2000, 8000, 57664):
    jobs = synth.sw_jobs(n, seed=900 + n)
    for _ in range(3):
        ctx.swalign2_batch(opt, XTRA, **jobs)
    reps = 20 if n < 8000 else 5
    s0 = ctx.stats().sw_kernel_ms
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.swalign2_batch(opt, XTRA, **jobs)
    dt = (time.perf_counter() - t0) / reps
    span = (ctx.stats().sw_kernel_ms - s0) / reps
    print(f"n={n:6d} wall {dt*1e3:8.3f} ms  device span {span:8.3f} ms  {n/dt/1e6:7.3f} M jobs/s")
print("ring stats", ctx.ring_stats())
ctx.close()
