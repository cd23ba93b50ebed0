"""How much does the resident rescue kernel cost the extension kernels?  Times bpsw_extend_batch (32 768 reads) alone, beside an epoch
that is kept alive by tiny rescue calls (workers resident, idle), and beside a stream of full rescue batches (workers busy)."""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import numpy as np
import bpsw_hip
from bpsw_hip import synth
import pyoracle as po
XTRA = po.KSW_XSUBO | po.KSW_XSTART | po.KSW_XBYTE | 19
opt = bpsw_hip.default_opt()
soa = synth.ext_tasks(32768, read_len=150, seed=77)
wire = bpsw_hip.wire_pack(soa)
cx = bpsw_hip.Context(0)
def time_ext(label, reps=30):
    for _ in range(3): cx.extend_batch(wire)
    s0 = cx.stats()
    t0 = time.perf_counter()
    for _ in range(reps): cx.extend_batch(wire)
    dt = (time.perf_counter() - t0) / reps
    s1 = cx.stats()
    print(f"{label:34s} ext call {dt*1e3:7.3f} ms   kernel {(s1.ext_kernel_ms - s0.ext_kernel_ms)/reps:7.3f} ms  h2d {(s1.ext_h2d_ms - s0.ext_h2d_ms)/reps:6.3f}", flush=True)
time_ext("alone")
stop = False
def bg(n, nthreads_tag):
    c = bpsw_hip.Context(0)
    jobs = synth.sw_jobs(n, seed=5 + n)
    cnt = 0
    while not stop:
        c.swalign2_batch(opt, XTRA, **jobs); cnt += 1
    print(f"   bg n={n}: {cnt} calls, ring {c.ring_stats()}", flush=True)
    c.close()
for n, nt in ((2, 1), (428, 1), (428, 8)):
    stop = False
    ths = [threading.Thread(target=bg, args=(n, nt)) for _ in range(nt)]
    [t.start() for t in ths]
    time.sleep(0.2)
    time_ext(f"beside {nt} thread(s) x {n}-job calls")
    stop = True
    [t.join() for t in ths]
cx.close()
