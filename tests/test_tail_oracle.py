"""Pins the oracle's restatement of worker2's tail (oracle/bpsw_oracle_tail.c) on the reference: committed golden vectors made
by the reference's own mem_sam_pe / mem_reg2aln (tests/golden/make_golden.py) and, where oracle/_ref exists, live runs of
mem_mark_primary_se / mem_pair / mem_approx_mapq_se / mem_reg2aln / mem_sam_pe on larger seeded samples.  CPU only."""
import os

import numpy as np
import pytest

import pyoracle as po
from tail_util import G, load_sam_pe_golden, synthetic_group


@pytest.mark.parametrize("stem", ["mem_sam_pe", "mem_sam_pe_all", "mem_sam_pe_rg"])
def test_sam_pe_vs_golden_text(orc, stem):
    pac, g, flag, want = load_sam_pe_golden(stem)
    opt, topt = orc.default_opt(), orc.default_tail_opt()
    opt.flag = flag
    topt.rg_id = g.rg_id                    # "mem_sam_pe_rg": the reference ran with a read group (bwa_rg_id): RG:Z on every line
    assert all((b"\tRG:Z:" + g.rg_id in w) if g.rg_id else (b"\tRG:Z:" not in w) for w in want)
    got, _, n_jobs = orc.sam_pe_batch(opt, topt, pac, g, flavour=po.TAIL_C)
    assert got == want                      # byte for byte, every field of every SAM line
    assert n_jobs >= g.group_size           # memRegToAln ran for (almost) every read
    kinds = {int(w.split(b"\t")[1]) & 0x2 for w in want}
    assert kinds == {0, 2}                  # both properly paired and unpaired outcomes are present
    assert any(b"I" in w.split(b"\t")[5] or b"D" in w.split(b"\t")[5] for w in want)   # gapped CIGARs (the DP path)
    if stem != "mem_sam_pe_rg":             # (the smaller fixture need not hold every kind)
        assert any(w.split(b"\t")[1] in (b"77", b"141", b"69", b"133", b"73", b"137", b"89", b"153") for w in want)  # unmapped ends


def test_reg2aln_vs_golden(orc):
    z = np.load(os.path.join(G, "mem_reg2aln.npz"))
    opt, topt = orc.default_opt(), orc.default_tail_opt()
    regs = z["regs"].astype(po.ALNREG_DTYPE)
    alns, cig, md = orc.reg2aln_batch(opt, topt, z["pac"], int(z["l_pac"]), z["ann_off"], z["ann_len"], z["read_len"], z["read_off"],
                                      z["read_pool"], regs, flavour=po.TAIL_C, cigar_cap=32, md_cap=160)
    want = z["alns"]
    assert int((alns["status"] != 0).sum()) == 0
    for f in ("pos", "rid", "flag", "is_rev", "mapq", "NM", "n_cigar", "score", "sub", "md_len"):
        assert np.array_equal(alns[f], want[f]), f
    assert np.array_equal(cig, z["cigar"]) and np.array_equal(md, z["md"])
    assert int((want["rid"] < 0).sum()) == 1 and int((want["flag"] & 0x100 != 0).sum()) > 0
    # the fixture reaches bwaFixXref2's repair: some region was cut at a contig boundary, so its CIGAR carries a clip that the
    # region's own qb/qe do not explain
    explained = (regs["qb"][:-1] != 0) | (regs["qe"][:-1] != z["read_len"][:-1])
    has_clip = np.array([any((int(c) & 0xf) == 3 for c in cig[j, :want["n_cigar"][j]]) for j in range(len(want) - 1)])
    assert int((has_clip & ~explained).sum()) > 0


def test_scala_flavour_differences_are_the_documented_ones(orc):
    """T1..T4: the Scala text and the C agree except where the transcription changed the arithmetic."""
    opt, topt = orc.default_opt(), orc.default_tail_opt()
    # T2: mapQ at l == mapQCoefLen
    r = np.zeros(1, po.ALNREG_DTYPE)[0]
    r["qb"], r["qe"], r["rb"], r["re"], r["score"], r["seedcov"] = 0, 50, 1000, 1050, 30, 25
    assert orc.approx_mapq(opt, topt, r, po.TAIL_C) != orc.approx_mapq(opt, topt, r, po.TAIL_SCALA)
    r["qe"], r["re"] = 51, 1051
    assert orc.approx_mapq(opt, topt, r, po.TAIL_C) == orc.approx_mapq(opt, topt, r, po.TAIL_SCALA)
    # T4: the parent index of a secondary hit (three overlapping hits and one disjoint primary)
    regs = np.zeros(4, po.ALNREG_DTYPE)
    for i, (qb, qe, sc) in enumerate(((0, 100, 90), (0, 100, 85), (110, 150, 75), (5, 95, 70))):
        regs[i]["qb"], regs[i]["qe"], regs[i]["rb"], regs[i]["re"], regs[i]["score"] = qb, qe, 5000 * (i + 1), 5000 * (i + 1) + qe - qb, sc
    c = orc.mark_primary(opt, topt, regs, 10, po.TAIL_C)
    s = orc.mark_primary(opt, topt, regs, 10, po.TAIL_SCALA)
    assert list(c["score"]) == [90, 85, 75, 70] and list(c["secondary"]) == [-1, 0, -1, 0]
    assert list(s["secondary"]) == [-1, 0, -1, 2]    # z(k) read after k was stepped (MemMarkPrimarySe.scala:93-101)
    assert list(c["sub"]) == list(s["sub"]) == [85, 0, 0, 0] and list(c["sub_n"]) == [1, 0, 0, 0]


def test_tail_vs_reference_live(orc, ref):
    opt, topt = orc.default_opt(), orc.default_tail_opt()
    total = 0
    for k, (es, ei, flag) in enumerate(((0.01, 0.002, 0), (0.04, 0.015, 0), (0.02, 0.005, po.MEM_F_ALL),
                                        (0.03, 0.01, po.MEM_F_NO_MULTI | po.MEM_F_ALL), (0.01, 0.002, po.MEM_F_NOPAIRING))):
        pac, g = synthetic_group(orc, 300, 900 + 10 * k, zdrop_mode=po.ZDROP_BWA, sub_rate=es, indel_rate=ei, p_span=0.06)
        o = orc.default_opt()
        o.flag = flag
        want = ref.sam_pe_batch(o, topt, pac, g)
        got, _, _ = orc.sam_pe_batch(o, topt, pac, g, flavour=po.TAIL_C)
        assert got == want
        total += len(want)
        # the pieces, one by one
        at = 0
        for r in range(2 * g.group_size):
            c = int(g.reg_cnt[r])
            if c:
                a = orc.mark_primary(opt, topt, g.regs[at:at + c], 2 * (g.id0 + r // 2) | (r & 1), po.TAIL_C)
                b = ref.mark_primary(opt, topt, g.regs[at:at + c], 2 * (g.id0 + r // 2) | (r & 1))
                assert a.tobytes() == b.tobytes()
                for reg in a[:2]:
                    assert orc.approx_mapq(opt, topt, reg, po.TAIL_C) == ref.approx_mapq(opt, topt, reg)
            at += c
        at = 0
        for p in range(g.group_size):
            c0, c1 = int(g.reg_cnt[2 * p]), int(g.reg_cnt[2 * p + 1])
            if c0 and c1:
                a0, a1 = g.regs[at:at + c0], g.regs[at + c0:at + c0 + c1]
                assert orc.mem_pair(opt, g.l_pac, g.pes, a0, a1, g.id0 + p, po.TAIL_C) == ref.mem_pair(opt, g.l_pac, g.pes, a0, a1, g.id0 + p)
            at += c0 + c1
    assert total == 3000
