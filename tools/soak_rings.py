#!/usr/bin/env python3
"""Soak of the two submission rings together (csrc/bpsw_ring.h): T task threads for S seconds, each making a random sequence of small
and large extension calls (through the extension ring / with a launch), SW batches and rescue groups of every size (the rescue ring's
two classes), contexts created and destroyed on the way, `bpsw_ref_load` from one thread now and then (every open epoch is closed for
it) -- every result compared with the oracle's, computed once per input.  A ring that loses a batch shows as a watchdog error, a
wrong one as a difference.  Usage on a GPU box: python tools/soak_rings.py [seconds] [threads]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402
from bpsw_hip import synth  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
orc = po.Oracle()
XTRA = po.KSW_XSUBO | po.KSW_XSTART | po.KSW_XBYTE | 19
ext, sw, grp = [], [], []
for k, (n, rl) in enumerate(((3, 150), (40, 150), (64, 100), (130, 150), (250, 150), (64, 250), (240, 250), (700, 150), (3000, 150), (12000, 150))):
    soa = synth.ext_tasks(n + 8, read_len=rl, seed=9100 + k, sub_rate=0.06 if rl == 250 else 0.01, indel_rate=0.015 if rl == 250 else 0.001)
    soa = soa.subset(np.arange(min(n, soa.n)))
    wire = bpsw_hip.wire_pack(soa)
    ext.append((wire, np.asarray(orc.wire_extend(wire)[0]).reshape(-1)))
for k, (n, rl) in enumerate(((1, 150), (5, 150), (64, 150), (300, 150), (2000, 150), (40, 250), (500, 250))):
    kw = dict(read_len=rl, win_min=700, win_max=1100, sub_rate=0.06, indel_rate=0.01) if rl == 250 else {}
    jobs = synth.sw_jobs(n, seed=9200 + k, **kw)
    sw.append((jobs, orc.sw_align2_jobs(orc.default_opt(), XTRA, **jobs)[0]))
for k, (gs, p) in enumerate(((10, 0.5), (64, 0.4), (400, 0.2), (2000, 0.1))):
    g = synth.rescue_group(gs, seed=9300 + k, p_resc=p)
    cnt, regs, _, _ = orc.matesw_group(orc.default_opt(), g, po.RESCUE_C)
    grp.append((g, cnt, regs))
l_pac = 200_003
pac, _ = synth.random_pac(l_pac, seed=9400)
stop_at = time.time() + seconds
errors, counts = [], [0, 0, 0, 0]
lock = threading.Lock()


def worker(t):
    rng = np.random.default_rng(9500 + t)
    c = bpsw_hip.Context(0)
    opt = bpsw_hip.default_opt()
    n = [0, 0, 0, 0]
    try:
        while time.time() < stop_at and not errors:
            u = rng.random()
            if u < 0.45:
                wire, want = ext[int(rng.integers(len(ext)))]
                assert np.array_equal(np.asarray(c.extend_batch(wire)).reshape(-1), want), ("extend", wire.size)
                n[0] += 1
            elif u < 0.7:
                jobs, want = sw[int(rng.integers(len(sw)))]
                assert np.array_equal(c.swalign2_batch(opt, XTRA, **jobs), want), "swalign2"
                n[1] += 1
            elif u < 0.97:
                g, cnt, regs = grp[int(rng.integers(len(grp)))]
                gc, gr = c.matesw_group(opt, g)
                assert np.array_equal(gc, cnt) and gr.tobytes() == regs.tobytes(), "group"
                n[2] += 1
            elif u < 0.99:
                c.close()
                c = bpsw_hip.Context(0)
                n[3] += 1
            elif t == 0:
                c.ref_load(pac, l_pac)     # closes every open epoch of the device, then the rings go on
    except BaseException as e:   # noqa: BLE001
        errors.append((t, repr(e)))
    finally:
        c.close()
        with lock:
            for i in range(4):
                counts[i] += n[i]


threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
t0 = time.time()
for th in threads:
    th.start()
last = t0
while any(th.is_alive() for th in threads):
    time.sleep(1.0)
    if time.time() - last > 30:
        last = time.time()
        print(f"  {last - t0:.0f} s ...", flush=True)
    if time.time() > stop_at + 120:
        print("SOAK_RINGS: threads still alive two minutes after the end", flush=True)
        os._exit(3)
c = bpsw_hip.Context(0)
print("SOAK_RINGS", {"seconds": round(time.time() - t0, 1), "threads": T, "extend_calls": counts[0], "sw_batches": counts[1], "groups": counts[2],
                     "contexts_recreated": counts[3], "ext_ring_and_rescue_ring_stats": c.ring_stats(),
                     "ring_integrity_on_checked_faults": c.ring_integrity(), "errors": errors[:3]})
sys.exit(1 if errors else 0)
