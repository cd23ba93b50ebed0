import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bpsw_hip, pyoracle as po
from bpsw_hip import synth
L = int(sys.argv[1]); n = int(sys.argv[2])
soa = synth.ext_tasks(n, read_len=L, sub_rate=0.01, indel_rate=0.001, tail_frac=0.0, seed=42 + L)
lq = np.maximum(soa.left_qlen, soa.right_qlen)
print("tasks", soa.n, "max side", lq.max(), "sides >127:", int((lq > 127).sum()), flush=True)
wire = bpsw_hip.wire_pack(soa)
orc = po.Oracle()
want, _ = orc.wire_extend(wire)
ctx = bpsw_hip.Context(0)
for rep in range(3):
    t0 = time.time()
    got = ctx.extend_batch(wire)
    bad = np.nonzero((got != want).reshape(-1, 10).any(axis=1))[0]
    print("rep", rep, "ms", round(1e3 * (time.time() - t0), 2), "bad", bad.size, bad[:5], flush=True)
