"""One of the two executor processes of tests/test_two_processes_gpu.py: a 12-thread feeder over both boundaries on device 0,
every output compared with the oracle.  Prints one JSON line."""
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
from bpsw_hip import synth, feeder as fd  # noqa: E402

who = int(sys.argv[1])
go_at = float(sys.argv[2])          # both processes start their timed loop at the same wall-clock instant
soas = [synth.ext_tasks(6000, seed=9000 + 100 * who + b) for b in range(8)]
wires = [bpsw_hip.wire_pack(s) for s in soas]
groups = [synth.rescue_group_fast(512, seed=9500 + 100 * who + g, p_resc=0.3) for g in range(16)]
ext_outs = [np.zeros(10 * s.n, np.int16) for s in soas]
structs = [g.as_struct() for g in groups]
cnts = [np.zeros(2 * g.group_size, np.int32) for g in groups]
regs = [np.empty(int(g.regs.shape[0] + g.ref_rb.shape[0] + 16), bpsw_hip.ALNREG_DTYPE) for g in groups]
items, order = fd.make_items(wires, ext_outs, groups, structs, cnts, regs)
F = fd.Feeder(12, 0, bpsw_hip.default_opt())
F.run(items)                                     # warm-up: contexts, streams, arenas
while time.time() < go_at:
    time.sleep(0.001)
t0 = time.perf_counter()
reps = 6
for _ in range(reps):
    F.run(items)
dt = time.perf_counter() - t0
totals = {i: it.out_total for it, (kind, i) in zip(items, order) if kind == 1}
F.close()
orc = po.Oracle()
bad = 0
for w, out in zip(wires, ext_outs):
    want, _ = orc.wire_extend(w)
    bad += int(not np.array_equal(out, want))
for i, g in enumerate(groups):
    wcnt, wregs, _, _ = orc.matesw_group(orc.default_opt(), g, po.RESCUE_C)
    got = regs[i][: int(totals[i])]
    ok = np.array_equal(cnts[i], wcnt) and len(got) == len(wregs) and all(np.array_equal(got[f], wregs[f]) for f in got.dtype.names)
    bad += int(not ok)
reads = sum(s.n for s in soas) + 2 * sum(g.group_size for g in groups)
print(json.dumps({"who": who, "bad": bad, "seconds": round(dt, 4), "units_per_s": round(reads * reps / dt, 1),
                  "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "stream_pool": os.environ.get("BPSW_STREAM_POOL")}), flush=True)
