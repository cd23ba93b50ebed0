#!/usr/bin/env python3
"""Rate of the on-device memChainToAlnBatched round loop (bpsw_chain2aln_batch, host-buffer entry: PCIe inclusive).
Usage on a GPU box:  python tools/chain2aln_rate.py [n_reads]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd"))
import bpsw_hip  # noqa: E402
from bpsw_hip import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
l_pac = 8_000_003
pac, bases = synth.random_pac(l_pac, seed=synth.CONFIG_SEED_BASE + 3)
b = synth.read_chains(n, bases, l_pac, read_len=150, sub_rate=0.01, indel_rate=0.001, seed=synth.CONFIG_SEED_BASE + 3)
ctx = bpsw_hip.Context(0)
ctx.ref_load(pac, l_pac)
opt = bpsw_hip.default_opt()
for _ in range(3):
    cnt, regs = ctx.chain2aln_batch(opt, b)
s0 = ctx.stats()
R = 20
t0 = time.perf_counter()
for _ in range(R):
    ctx.chain2aln_batch(opt, b)
dt = (time.perf_counter() - t0) / R
s1 = ctx.stats()
print(json.dumps({"reads": n, "chains": int(b.chain_cnt.sum()), "seeds": int(b.seed_len.shape[0]), "regions": int(regs.shape[0]),
                  "ms_per_call": round(1e3 * dt, 3), "kernel_ms": round((s1.ext_kernel_ms - s0.ext_kernel_ms) / R, 4),
                  "reads_per_s": round(n / dt)}))
