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
n = int(sys.argv[1])
jobs = synth.sw_jobs(n, seed=900 + n)
for _ in range(3):
    ctx.swalign2_batch(opt, XTRA, **jobs)
s0 = ctx.stats().sw_kernel_ms
t0 = time.perf_counter()
for _ in range(20):
    ctx.swalign2_batch(opt, XTRA, **jobs)
dt = (time.perf_counter() - t0) / 20
print(f"wg_per_cu {os.environ.get('BPSW_RING_WG_PER_CU')} n={n} wall {dt*1e3:.3f} ms span {(ctx.stats().sw_kernel_ms - s0)/20:.3f} ms", flush=True)
time.sleep(0.05)
ctx.close()
