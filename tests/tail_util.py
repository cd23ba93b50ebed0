"""Helpers shared by the worker2-tail tests: golden-file loaders and the synthetic pipeline reads -> regions -> group."""
import os

import numpy as np

import bpsw_hip
import pyoracle as po
from bpsw_hip import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_sam_pe_golden(stem):
    z = np.load(os.path.join(G, stem + ".npz"))
    pes = [(int(r[0]), int(r[1]), int(r[2]), float(r[3]), float(r[4])) for r in z["pes"]]
    g = bpsw_hip.TailGroupSoA(group_size=int(z["read_len"].shape[0]) // 2, l_pac=int(z["l_pac"]), id0=int(z["id0"]), pes=pes,
                              read_len=z["read_len"], read_off=z["read_off"], read_pool=z["read_pool"], qual_pool=z["qual_pool"],
                              name_off=z["name_off"], name_pool=z["name_pool"], reg_cnt=z["reg_cnt"],
                              regs=z["regs"].astype(po.ALNREG_DTYPE), ann_off=z["ann_off"], ann_len=z["ann_len"],
                              ann_name_off=z["ann_name_off"], ann_name_pool=z["ann_name_pool"])
    text, off = z["text"].tobytes(), z["text_off"]
    want = [text[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
    return z["pac"], g, int(z["flag"]), want


def synthetic_group(orc, n_pairs, seed, contigs=(70_000, 50_000, 30_000, 50_000), zdrop_mode=po.ZDROP_SCALA, dedup_mode=po.RESCUE_C,
                    id0=77, **kw):
    """pairs -> chains -> regions (the ORACLE's memChainToAln + memSortAndDedup) -> TailGroupSoA; returns (pac, group)"""
    pac, bases, off, ln, names, dups = synth.contig_reference(list(contigs), seed=seed)
    tb, rn, quals, pes = synth.tail_pairs(n_pairs, bases, off, ln, dups, seed=seed + 1, **kw)
    cnt, regs, _, _ = orc.chain2aln_batch(orc.default_opt(), pac, tb, zdrop_mode)
    out_cnt, out, at = [], [], 0
    for c in cnt:
        r = orc.sort_dedup(regs[at:at + c], mode=dedup_mode) if c else regs[0:0]
        at += c
        out_cnt.append(len(r)); out.append(r)
    g = bpsw_hip.make_tail_group(tb, rn, quals, pes, np.array(out_cnt, np.int32), np.concatenate(out), off, ln, names, id0=id0)
    return pac, g
