#!/usr/bin/env python3
"""Soak test of worker2's tail (bpsw_sam_pe_batch / bpsw_worker2_batch) against the oracle: many seeds, read lengths, error
rates, option flags, both flavours.  Usage on a GPU box: python tools/soak_tail.py [rounds]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402
from tail_util import rescue_group_of, synthetic_group_with_bases  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx, orc = bpsw_hip.Context(0), po.Oracle()
reads = bad = 0
for rd in range(rounds):
    rng = np.random.default_rng(7000 + rd)
    L = int(rng.choice([76, 100, 150, 150, 250]))
    es = float(rng.choice([0.005, 0.02, 0.05, 0.1]))
    flag = int(rng.choice([0, 0, bpsw_hip.MEM_F_ALL, bpsw_hip.MEM_F_NO_MULTI | bpsw_hip.MEM_F_ALL, bpsw_hip.MEM_F_NOPAIRING]))
    flavour = rd & 1
    contigs = [int(x) for x in rng.integers(20_000, 90_000, int(rng.integers(1, 7)))]
    pac, bases, g = synthetic_group_with_bases(orc, 350, 9000 + 7 * rd, contigs=contigs, zdrop_mode=flavour, dedup_mode=flavour ^ 1,
                                               read_len=L, sub_rate=es, indel_rate=es / 4, p_span=0.08, p_hard=0.15, p_dup=0.3,
                                               p_unmappable=0.05, id0=int(rng.integers(0, 1 << 40)))
    names = [bytes(g.ann_name_pool[int(g.ann_name_off[i]):int(g.ann_name_off[i + 1])]).decode() for i in range(g.ann_off.shape[0])]
    ctx.ref_load(pac, g.l_pac)
    ctx.bns_load(g.ann_off, g.ann_len, names)
    opt, oopt = bpsw_hip.default_opt(), orc.default_opt()
    opt.flag = oopt.flag = flag
    # tail alone
    want, _, _ = orc.sam_pe_batch(oopt, orc.default_tail_opt(), pac, g, flavour=flavour)
    got, _ = ctx.sam_pe_batch(opt, bpsw_hip.default_tail_opt(flavour), g)
    b1 = sum(1 for a, b in zip(want, got) if a != b)
    # rescue + tail
    mode = flavour ^ 1
    rg = rescue_group_of(g, bases, oopt)
    cnt_o, regs_o, _, _ = orc.matesw_group(oopt, rg, mode)
    g2 = copy.copy(g)
    g2.reg_cnt, g2.regs = cnt_o, regs_o
    want2, _, _ = orc.sam_pe_batch(oopt, orc.default_tail_opt(), pac, g2, flavour=flavour)
    got2, cnt, _ = ctx.worker2_batch(opt, bpsw_hip.default_tail_opt(flavour), g, mode)
    b2 = sum(1 for a, b in zip(want2, got2) if a != b) + int(not np.array_equal(cnt, cnt_o))
    reads += 2 * len(want)
    bad += b1 + b2
    print("round", rd, {"L": L, "err": es, "flag": hex(flag), "flavour": flavour, "contigs": len(contigs)}, "bad", b1, b2, flush=True)
print("SOAK_TAIL", {"read_runs": reads, "bad": bad})
