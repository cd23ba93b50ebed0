"""Total SW throughput of T threads that loop over bpsw_swalign2_batch(n jobs), through the ring or (BPSW_RING=0) through launches."""
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
print("BPSW_RING", os.environ.get("BPSW_RING", "1"), "WG_PER_CU", os.environ.get("BPSW_RING_WG_PER_CU"), flush=True)
for n in (428, 2000):
    jobs = synth.sw_jobs(n, seed=5 + n)
    for T in (1, 4, 16):
        counts = [0] * T
        stop = [False]
        def bg(i):
            c = bpsw_hip.Context(0)
            c.swalign2_batch(opt, XTRA, **jobs)
            while not stop[0]:
                c.swalign2_batch(opt, XTRA, **jobs); counts[i] += 1
            c.close()
        ths = [threading.Thread(target=bg, args=(i,)) for i in range(T)]
        [t.start() for t in ths]
        time.sleep(0.3)
        c0 = sum(counts); t0 = time.perf_counter()
        time.sleep(1.0)
        c1 = sum(counts); dt = time.perf_counter() - t0
        stop[0] = True
        [t.join() for t in ths]
        print(f"n={n:5d} T={T:2d}: {(c1-c0)/dt:9.0f} calls/s  {(c1-c0)*n/dt/1e6:7.3f} M jobs/s", flush=True)
