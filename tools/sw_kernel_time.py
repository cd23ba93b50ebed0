#!/usr/bin/env python3
"""Stand-alone time of the rescue SW kernel for n synthetic 2x150 bp jobs (device-resident entry).
Usage on a GPU box: [BPSW_SW_QUAD=0] [BPSW_SW_PACK=0] python tools/sw_kernel_time.py [n_jobs] [read_len] [window]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bpsw_hip  # noqa: E402
from bpsw_hip import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7208
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
W = int(sys.argv[3]) if len(sys.argv) > 3 else 400
dev = torch.device("cuda", 0)
ctx = bpsw_hip.Context(0)
jobs = synth.sw_jobs(n, read_len=L, win_min=W, win_max=W, sub_rate=0.02, indel_rate=0.002, unrelated_frac=0.05, decoy_frac=0.1,
                     rev_frac=1.0, seed=synth.CONFIG_SEED_BASE + 103)
d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in jobs.items()}
out = torch.zeros((n, 7), dtype=torch.int32, device=dev)
sj = bpsw_hip.SwJobs()
sj.n, sj.xtra = n, bpsw_hip.KSW_XSUBO | bpsw_hip.KSW_XSTART | bpsw_hip.KSW_XBYTE | 19
for k in ("q_len", "t_len", "q_off", "t_off", "q_rev", "q_pool", "t_pool"):
    setattr(sj, k, d[k].data_ptr())
sj.q_pool_bytes, sj.t_pool_bytes = d["q_pool"].numel(), d["t_pool"].numel()
opt = bpsw_hip.default_opt()
torch.cuda.synchronize()
ms = []
for _ in range(8):
    ctx.swalign2_batch_device(opt, sj, out.data_ptr(), 0)
    ms.append(ctx.last_kernel_ms()[1])
print({"quad": os.environ.get("BPSW_SW_QUAD", "auto"), "pack": os.environ.get("BPSW_SW_PACK", "1"), "jobs": n, "read_len": L, "window": W, "kernel_ms": round(float(np.mean(ms[2:])), 4),
       "jobs_per_s": round(n / (np.mean(ms[2:]) * 1e-3))})
