"""Pins the oracle (oracle/bpsw_oracle.c) against the committed golden vectors, which were produced by the
reference's own C sources (tests/golden/make_golden.py), and against an independent pure-Python
transliteration of the Scala text on small cases.  CPU only."""
import os

import numpy as np
import pytest

import bpsw_hip
import pyoracle as po
import scala_text
from conftest import region_fields_equal

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAT = po.default_mat()


def _seqs(z, name):
    off, pool = z[name + "_off"], z[name + "_pool"]
    return [pool[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def test_sw_extend_vs_ksw_extend2_golden(orc):
    z = np.load(os.path.join(G, "ksw_extend2.npz"))
    qs, ts = _seqs(z, "q"), _seqs(z, "t")
    n_diverge = 0
    for q, t, (w, bonus, zdrop, h0), want in zip(qs, ts, z["params"], z["out"]):
        got, _ = orc.sw_extend(q, t, MAT, 6, 1, 6, 1, int(w), int(bonus), int(zdrop), int(h0), po.ZDROP_BWA)
        assert np.array_equal(got, want)
        got_s, _ = orc.sw_extend(q, t, MAT, 6, 1, 6, 1, int(w), int(bonus), int(zdrop), int(h0), po.ZDROP_SCALA)
        n_diverge += int(not np.array_equal(got_s, want))
    assert n_diverge < len(qs) // 4       # the Scala parse only differs inside the z-drop branch


def test_sw_align2_vs_ksw_align2_golden(orc):
    z = np.load(os.path.join(G, "ksw_align2.npz"))
    opt = orc.default_opt()
    exact2 = 0
    for q, t, want in zip(_seqs(z, "q"), _seqs(z, "t"), z["out"]):
        xtra = po.KSW_XSUBO | po.KSW_XSTART | po.KSW_XBYTE | 19
        got, _ = orc.sw_align2(q, t, opt, xtra)
        assert np.array_equal(got[[0, 1, 2, 5, 6]], want[[0, 1, 2, 5, 6]])  # score, te, qe, tb, qb
        # B8: the C's padded SSE2 rows can only ADD second-best candidates, so where the two differ the C reports the larger
        # score2 (or one where the true DP has none, -1), never a smaller one
        assert got[3] <= want[3]
        exact2 += int(np.array_equal(got[[3, 4]], want[[3, 4]]))
    assert exact2 > 0.9 * len(z["out"])   # the SSE2 padding effect is a few per cent


def test_sw_global_vs_ksw_global2_golden(orc):
    z = np.load(os.path.join(G, "ksw_global2.npz"))
    co, cp = z["cig_off"], z["cig_pool"]
    for i, (q, t) in enumerate(zip(_seqs(z, "q"), _seqs(z, "t"))):
        score, cig = orc.sw_global(q, t, MAT, 6, 1, 6, 1, int(z["w"][i]))
        assert score == int(z["score"][i])
        assert np.array_equal(cig, cp[co[i]:co[i + 1]])


def test_sort_dedup_vs_mem_sort_and_dedup_golden(orc):
    z = np.load(os.path.join(G, "mem_sort_and_dedup.npz"))
    io, oo = z["in_off"], z["out_off"]
    for i in range(len(io) - 1):
        got = orc.sort_dedup(z["regs_in"][io[i]:io[i + 1]], 0.95, po.RESCUE_C)
        region_fields_equal(got, z["regs_out"][oo[i]:oo[i + 1]])


@pytest.mark.parametrize("tag", ["fr", "all4", "250", "250_all4"])
def test_group_rescue_vs_mem_group_matesw_golden(orc, tag):
    import bpsw_hip
    z = np.load(os.path.join(G, f"mem_group_matesw_{tag}.npz"))
    g = bpsw_hip.RescueGroupSoA(group_size=int(z["group_size"]), l_pac=int(z["l_pac"]), pes=[(int(r[0]), int(r[1]), int(r[2]), float(r[3]), float(r[4])) for r in z["pes"]],
                                seq_len=z["seq_len"], seq_off=z["seq_off"], seq_pool=z["seq_pool"], reg_cnt=z["reg_cnt"],
                                regs=z["regs"], ref_cnt=z["ref_cnt"], ref_rb=z["ref_rb"], ref_re=z["ref_re"], ref_len=z["ref_len"],
                                ref_off=z["ref_off"], ref_pool=z["ref_pool"])
    opt = orc.default_opt()
    assert [opt.a, opt.b, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, opt.pen_unpaired, opt.pen_clip5, opt.pen_clip3, opt.w, opt.zdrop,
            opt.T, opt.flag, opt.min_seed_len, opt.max_ins, opt.max_matesw] == list(z["opt_ints"])     # == mem_opt_init()
    assert list(opt.mat) == list(z["opt_mat"]) and abs(opt.mask_level_redun - float(z["opt_mask"])) < 1e-9
    cnt, regs, n_sw, _ = orc.matesw_group(opt, g, po.RESCUE_C)
    assert n_sw > 0
    assert np.array_equal(cnt, z["out_cnt"])
    region_fields_equal(regs, z["out_regs"], skip=("csub",))   # csub = score2 inherits B8 in the C library
    assert (regs["csub"] != z["out_regs"]["csub"]).mean() < 0.1


# ---- independent second opinion on the Scala-only behaviours (small cases, pure Python) ------------------
def test_oracle_vs_python_transliteration_sw_extend(orc):
    rng = np.random.default_rng(1)
    n_div = 0
    for n in range(300):
        ql = int(rng.integers(1, 60))
        t = rng.integers(0, 5 if n % 9 == 0 else 4, ql + int(rng.integers(0, 50))).astype(np.uint8)
        q = t[:ql].copy()
        flip = rng.random(ql) < [0.02, 0.1, 0.3][n % 3]
        q[flip] = (q[flip] + 1) % 4
        if n % 4 == 0:
            q[ql // 2:] = rng.integers(0, 4, ql - ql // 2)
        w, zdrop, h0 = [100, 3, 10][n % 3], [100, 5, 15, 0][n % 4], int(rng.integers(1, 80))
        want = scala_text.sw_extend(q.tolist(), t.tolist(), MAT.tolist(), 6, 1, 6, 1, w, 5, zdrop, h0)
        got, _ = orc.sw_extend(q, t, MAT, 6, 1, 6, 1, w, 5, zdrop, h0, po.ZDROP_SCALA)
        assert got.tolist() == want
        bwa, _ = orc.sw_extend(q, t, MAT, 6, 1, 6, 1, w, 5, zdrop, h0, po.ZDROP_BWA)
        n_div += int(bwa.tolist() != want)
    assert n_div > 0   # the sample does reach the branch where the two parses differ


def test_zdrop_parse_known_answer(orc):
    """Hand-checkable B1 case: a 30-base exact match (max = h0+30 at row 29), then the reference continues with a
    long insertion-free stretch that only mismatches.  With zdrop = 5 the BWA parse tests the insertion arm when
    the row maximum has moved right of the diagonal (A false) and stops; the Scala parse never looks at C when A
    is false, so it keeps going until the row maximum reaches 0."""
    q = np.array([0, 1, 2, 3] * 10, np.uint8)
    t = np.concatenate([q[:30], np.full(40, 4, np.uint8)])
    s, _ = orc.sw_extend(q, t, MAT, 6, 1, 6, 1, 100, 5, 5, 20, po.ZDROP_SCALA)
    b, _ = orc.sw_extend(q, t, MAT, 6, 1, 6, 1, 100, 5, 5, 20, po.ZDROP_BWA)
    assert s.tolist() == scala_text.sw_extend(q.tolist(), t.tolist(), MAT.tolist(), 6, 1, 6, 1, 100, 5, 5, 20)
    assert s[0] == b[0] == 50 and s[1] == b[1] == 30 and s[2] == b[2] == 30      # the best cell is the same
    assert tuple(s[:3]) == (50, 30, 30)


def test_oracle_vs_python_transliteration_sw_align2(orc):
    rng = np.random.default_rng(2)
    opt = orc.default_opt()
    for n in range(60):
        L = int(rng.integers(5, 50))
        t = rng.integers(0, 4, L + int(rng.integers(10, 120))).astype(np.uint8)
        p = int(rng.integers(0, len(t) - L))
        q = t[p:p + L].copy()
        flip = rng.random(L) < 0.08
        q[flip] = (q[flip] + 1) % 4
        if n % 3 == 0 and len(t) > 2 * L + 5:
            t[-L:] = q            # a second copy -> second-best bookkeeping
        xtra = [po.KSW_XSUBO | po.KSW_XSTART | 8, po.KSW_XSTART | po.KSW_XSUBO | 19, po.KSW_XSUBO | 5, 0][n % 4]
        want = scala_text.sw_align2(q.tolist(), t.tolist(), MAT.tolist(), 1, 4, 6, 1, 6, 1, xtra)
        got, _ = orc.sw_align2(q, t, opt, xtra)
        assert got.tolist() == want


def test_bns_get_seq_vs_reference_golden(orc):
    """bnsGetSeq restatement (util/BNTSeqUtil.scala:37-79) against bns_get_seq outputs of the reference C"""
    z = np.load(os.path.join(G, "bns_get_seq.npz"))
    l_pac, pac = int(z["l_pac"]), z["pac"]
    want = _seqs(z, "seq")
    n_empty = 0
    for b, e, w in zip(z["beg"], z["end"], want):
        got = orc.bns_get_seq(l_pac, pac, int(b), int(e))
        assert np.array_equal(got, w)
        n_empty += int(len(w) == 0)
    assert 0 < n_empty < len(want) // 3   # the bridging windows (and a few empty ones) return nothing


def _chain_batch(z):
    from bpsw_hip import ChainBatchSoA
    return ChainBatchSoA(l_pac=int(z["l_pac"]), read_len=z["read_len"], read_off=z["read_off"], read_pool=z["read_pool"],
                         chain_cnt=z["chain_cnt"], seed_cnt=z["seed_cnt"], seed_rbeg=z["seed_rbeg"], seed_qbeg=z["seed_qbeg"],
                         seed_len=z["seed_len"])


def test_chain2aln_vs_mem_chain2aln_golden(orc):
    """memChainToAlnBatched restatement against mem_chain2aln outputs of the reference C (BWA z-drop parse)"""
    z = np.load(os.path.join(G, "mem_chain2aln.npz"))
    b = _chain_batch(z)
    cnt, regs, n_ext, _ = orc.chain2aln_batch(orc.default_opt(), z["pac"], b, po.ZDROP_BWA)
    assert np.array_equal(cnt, z["out_cnt"])
    for f in regs.dtype.names:
        assert np.array_equal(regs[f], z["out_regs"][f]), f
    assert n_ext > 300 and len(regs) < len(b.seed_len)       # extensions ran; contained seeds were skipped


def _ref_task_sets():
    z = np.load(os.path.join(G, "ref_extension_tasks.npz"))
    fields = ("left_qlen", "left_rlen", "right_qlen", "right_rlen", "left_q_off", "left_r_off", "right_q_off", "right_r_off",
              "reg_score", "q_beg", "h0", "idx")
    for si, (rl, sub, indel, w, zd, n) in enumerate(z["sets"].tolist()):
        soa = bpsw_hip.ExtTaskSoA(pool=z[f"s{si}_pool"], **{f: z[f"s{si}_{f}"] for f in fields})
        soa.w = int(w)
        yield si, soa, int(zd), z[f"s{si}_out"]


def test_extension_tasks_vs_reference_golden(orc):
    """whole two-sided extension tasks, band retries included: the oracle (BWA z-drop parse) against ref_extend_batch's outputs"""
    import bpsw_hip
    retried = 0
    for si, soa, zd, want in _ref_task_sets():
        got, _ = orc.wire_extend(bpsw_hip.wire_pack(soa), zdrop=zd, zdrop_mode=po.ZDROP_BWA)
        got = np.asarray(got).reshape(-1, 10)
        assert np.array_equal(got[:, 2:9].astype(np.int32), want), si
        retried += int((want[:, 6] > soa.w).sum())
    assert retried > 100   # the doubled band is exercised


def test_committed_fixtures_are_what_the_committed_script_generates():
    """tests/golden/*.npz must stay reproducible: make_golden.py --check regenerates every file from the reference's own C
    (oracle/_ref) and the seeded generators, and fails when an array differs from the committed one (a generator that changed
    after the files were written made three of them stale once).  Needs the reference build, which exists where
    /root/reference does; on the GPU box the committed files are simply used."""
    import subprocess
    import sys
    if not os.path.exists(po.REF_SO):
        pytest.skip("oracle/_ref not built here")
    r = subprocess.run([sys.executable, os.path.join(G, "make_golden.py"), "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
