#!/usr/bin/env python3
"""Rate of worker2's tail (bpsw_sam_pe_batch) on one MI355X: whole call, reg2aln kernel alone, and -- where oracle/_ref
travelled -- the reference's own mem_sam_pe on one host thread for the same group.
Usage on a GPU box:  python tools/tail_rate.py [n_pairs]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
from bpsw_hip import synth  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = bpsw_hip.Context(0)
pac, bases, off, ln, names, dups = synth.contig_reference([400_000, 300_000, 200_000, 100_000], seed=synth.CONFIG_SEED_BASE + 40)
tb, rn, quals, pes = synth.tail_pairs(n_pairs, bases, off, ln, dups, seed=synth.CONFIG_SEED_BASE + 41)
l_pac = int(off[-1] + ln[-1])
ctx.ref_load(pac, l_pac)
ctx.bns_load(off, ln, names)
opt = bpsw_hip.default_opt()
cnt, regs = ctx.chain2aln_batch(opt, tb, flags=bpsw_hip.C2A_SORT_DEDUP if hasattr(bpsw_hip, "C2A_SORT_DEDUP") else 1)
g = bpsw_hip.make_tail_group(tb, rn, quals, pes, cnt, regs, off, ln, names, id0=0)
topt = bpsw_hip.default_tail_opt()
for _ in range(2):
    texts, _ = ctx.sam_pe_batch(opt, topt, g)
R = 5
t0 = time.perf_counter()
kms = []
for _ in range(R):
    texts, _ = ctx.sam_pe_batch(opt, topt, g)
    kms.append(ctx.last_tail_kernel())
host = ctx.last_tail_host_ms()
dt = (time.perf_counter() - t0) / R
out = {"pairs": n_pairs, "regions": int(regs.shape[0]), "reg2aln_jobs": int(kms[-1][1]), "call_ms": round(1e3 * dt, 3),
       "kernel_ms": round(float(np.mean([k[0] for k in kms])), 4), "reads_per_s_call": round(2 * n_pairs / dt),
       "jobs_per_s_kernel": round(kms[-1][1] / (np.mean([k[0] for k in kms]) * 1e-3)),
       "host_plan_ms": round(host[0], 3), "device_roundtrip_ms": round(host[1], 3), "host_emit_ms": round(host[2], 3), "sam_bytes": sum(len(t) for t in texts), "gapped_cigars": sum(1 for t in texts if any(c in t.split(b"\t")[5] for c in (b"I", b"D")))}
try:
    import pyoracle as po
    if po.Ref.available():
        ref, orc = po.Ref(), po.Oracle()
        topt_c = orc.default_tail_opt()
        sub = min(n_pairs, 4096)
        import copy
        gs = copy.copy(g)
        gs.group_size = sub
        gs.read_len, gs.read_off, gs.name_off = g.read_len[:2 * sub], g.read_off[:2 * sub], g.name_off[:sub + 1]
        gs.reg_cnt = g.reg_cnt[:2 * sub]
        gs.regs = g.regs[:int(g.reg_cnt[:2 * sub].sum())]
        t0 = time.perf_counter()
        want = ref.sam_pe_batch(orc.default_opt(), topt_c, pac, gs)
        dtc = time.perf_counter() - t0
        out["reference_mem_sam_pe_reads_per_s_1thread"] = round(2 * sub / dtc)
        got_c, _ = ctx.sam_pe_batch(opt, bpsw_hip.default_tail_opt(bpsw_hip.TAIL_C), gs)
        out["identical_to_reference"] = bool(got_c == want)
except Exception as e:  # noqa: BLE001
    out["reference_error"] = repr(e)
print(json.dumps(out))
