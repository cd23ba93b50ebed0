#!/usr/bin/env python3
"""One bench-shaped extension batch through bpsw_extend_batch with a given shortcut mask, for instruction counters:
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ... -- python3 tools/ext_batch_instr.py MASK [config]
(what the exact shortcuts cost, and what they save: compare MASK 31 with MASK 0)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", ROOT):
    sys.path.insert(0, os.path.join(ROOT, p) if p != ROOT else ROOT)
import bench  # noqa: E402
import bpsw_hip  # noqa: E402

mask = int(sys.argv[1]) if len(sys.argv) > 1 else 31
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W = bench.WORKLOADS[cfg]
soa = bench.make_ext_soa(W, cfg, 0, 0)
wire = bpsw_hip.wire_pack(soa)
ctx = bpsw_hip.Context(0)
ctx.set_ext_shortcuts(mask)
for _ in range(4):
    ctx.extend_batch(wire)
print({"mask": mask, "tasks": soa.n, "launches": 4})
