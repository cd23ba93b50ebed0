#!/usr/bin/env python3
"""Stand-alone time of the extension launches for one bench-shaped wire batch (device-resident entry, HIP events on
the launch stream).  Usage on a GPU box:  [BPSW_EXT_MODE=lane|BPSW_EXT_QT=1] python tools/ext_kernel_time.py [n_ctx]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd"))
import torch  # noqa: E402
import bpsw_hip  # noqa: E402
from bpsw_hip import synth  # noqa: E402

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150          # read length
ES = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01   # substitution rate (indels at a tenth / a quarter of it for 250 bp)
dev = torch.device("cuda", 0)
ctxs = [bpsw_hip.Context(0) for _ in range(n_ctx)]
batches = []
for k in range(n_ctx):
    soa = synth.ext_tasks(32768, read_len=L, sub_rate=ES, indel_rate=ES / (4 if L > 200 else 10), n_rate=0.001,
                          tail_frac=0.01 if L > 200 else 0.0, seed=synth.CONFIG_SEED_BASE + 3 + k)
    w = bpsw_hip.wire_pack(soa)
    batches.append((w, soa.n, torch.from_numpy(w).to(dev), torch.zeros(10 * soa.n, dtype=torch.int16, device=dev)))
torch.cuda.synchronize()


def step():
    for cx, (w, n, dw, do) in zip(ctxs, batches):
        cx.extend_batch_device(dw.data_ptr(), int(w.size), n, do.data_ptr(), 0)
    return [cx.last_kernel_ms()[0] for cx in ctxs]


for _ in range(3):
    step()
torch.cuda.synchronize()
R = 20
t0 = time.perf_counter()
ms = [step() for _ in range(R)]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
print({"mode": os.environ.get("BPSW_EXT_MODE", "qt" if os.environ.get("BPSW_EXT_QT") else "wave"), "contexts": n_ctx, "read_len": L, "sub_rate": ES, "tasks": sum(b[1] for b in batches),
       "wall_ms_per_step": round(1e3 * dt, 3), "launch_ms_avg": round(sum(map(sum, ms)) / (R * n_ctx), 4),
       "reads_per_s": round(32768 * n_ctx / dt)})
