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
    g.rg_id = z["rg_id"].tobytes() if "rg_id" in z.files else b""   # the -R read-group ID the reference ran with
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


def rescue_group_of(g, bases, opt):
    """The arguments of mateSWJNI for a TailGroupSoA whose regs are the lists BEFORE the rescue: the test-side mirror of
    memSamPeGroupJNIPrepare (MemSamPe.scala:1895-2000) with getAlnRegRefJNI's windows (:1810-1878) cut as bytes."""
    l_pac = g.l_pac
    ref_cnt, rb_l, re_l, len_l, off_l, chunks = [], [], [], [], [], []
    at, pool_at = 0, 0
    for e in range(2 * g.group_size):
        n = int(g.reg_cnt[e])
        a = g.regs[at:at + n]
        mate_len = int(g.read_len[e ^ 1])
        cnt = 0
        for j in range(n):
            if cnt >= opt.max_matesw:
                break
            if not (a[j]["score"] >= a[0]["score"] - opt.pen_unpaired):
                continue
            for r in range(4):
                lo, hi, failed = g.pes[r][0], g.pes[r][1], g.pes[r][2]
                if failed:
                    rb_l.append(-1); re_l.append(-1); len_l.append(0); off_l.append(0)
                    continue
                is_rev, is_larger = (r >> 1) != (r & 1), (r >> 1) == 0
                arb = int(a[j]["rb"])
                if not is_rev:
                    rb = arb + lo if is_larger else arb - hi
                    re = (arb + hi if is_larger else arb - lo) + mate_len
                else:
                    rb = (arb + lo if is_larger else arb - hi) - mate_len
                    re = arb + hi if is_larger else arb - lo
                rb, re = max(rb, 0), min(re, 2 * l_pac)
                w = synth.window_bases(bases, l_pac, rb, re)
                rb_l.append(rb); re_l.append(re); len_l.append(len(w)); off_l.append(pool_at)
                chunks.append(w)
                pad = (-len(w)) % 16
                if pad:
                    chunks.append(np.zeros(pad, np.uint8))
                pool_at += len(w) + pad
            cnt += 1
        ref_cnt.append(cnt)
        at += n
    i64 = lambda v: np.array(v, np.int64)
    return bpsw_hip.RescueGroupSoA(group_size=g.group_size, l_pac=l_pac, pes=g.pes, seq_len=np.ascontiguousarray(g.read_len),
                                   seq_off=np.ascontiguousarray(g.read_off), seq_pool=np.ascontiguousarray(g.read_pool),
                                   reg_cnt=np.ascontiguousarray(g.reg_cnt), regs=np.ascontiguousarray(g.regs),
                                   ref_cnt=np.array(ref_cnt, np.int32), ref_rb=i64(rb_l), ref_re=i64(re_l), ref_len=i64(len_l),
                                   ref_off=i64(off_l), ref_pool=np.concatenate(chunks) if chunks else np.zeros(16, np.uint8))


def synthetic_group_with_bases(orc, n_pairs, seed, contigs=(70_000, 50_000, 30_000, 50_000), zdrop_mode=po.ZDROP_SCALA,
                               dedup_mode=po.RESCUE_C, id0=77, **kw):
    """like synthetic_group, also returning the unpacked reference (for cutting rescue windows)"""
    pac, bases, off, ln, names, dups = synth.contig_reference(list(contigs), seed=seed)
    tb, rn, quals, pes = synth.tail_pairs(n_pairs, bases, off, ln, dups, seed=seed + 1, **kw)
    cnt, regs, _, _ = orc.chain2aln_batch(orc.default_opt(), pac, tb, zdrop_mode)
    out_cnt, out, at = [], [], 0
    for c in cnt:
        r = orc.sort_dedup(regs[at:at + c], mode=dedup_mode) if c else regs[0:0]
        at += c
        out_cnt.append(len(r)); out.append(r)
    g = bpsw_hip.make_tail_group(tb, rn, quals, pes, np.array(out_cnt, np.int32), np.concatenate(out), off, ln, names, id0=id0)
    return pac, bases, g
