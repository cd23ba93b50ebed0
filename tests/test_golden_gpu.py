"""The HIP kernels against the committed golden vectors produced by the reference's own C
(tests/golden/make_golden.py) -- no oracle in between."""
import os

import numpy as np
import pytest

import bpsw_hip
import pyoracle as po
from conftest import region_fields_equal

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _seqs(z, name):
    off, pool = z[name + "_off"], z[name + "_pool"]
    return [pool[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def test_swalign_kernel_vs_ksw_align2_golden(ctx):
    z = np.load(os.path.join(G, "ksw_align2.npz"))
    qs, ts = _seqs(z, "q"), _seqs(z, "t")
    pad = lambda a: np.concatenate([a, np.zeros((-len(a)) % 16, np.uint8)])
    q_off, t_off, qp, tp = [], [], [], []
    qa = ta = 0
    for q, t in zip(qs, ts):
        q_off.append(qa); t_off.append(ta)
        qp.append(pad(q)); tp.append(pad(t))
        qa += len(qp[-1]); ta += len(tp[-1])
    got = ctx.swalign2_batch(bpsw_hip.default_opt(), po.KSW_XSUBO | po.KSW_XSTART | po.KSW_XBYTE | 19,
                             q_len=[len(q) for q in qs], t_len=[len(t) for t in ts], q_off=q_off, t_off=t_off,
                             q_rev=np.zeros(len(qs), np.uint8), q_pool=np.concatenate(qp + [np.zeros(16, np.uint8)]),
                             t_pool=np.concatenate(tp + [np.zeros(16, np.uint8)]))
    want = z["out"]
    assert np.array_equal(got[:, [0, 1, 2, 5, 6]], want[:, [0, 1, 2, 5, 6]])       # score, te, qe, tb, qb
    assert (got[:, [3, 4]] == want[:, [3, 4]]).all(axis=1).mean() > 0.9            # B8 tolerance on score2/te2 only


def test_extend_kernel_vs_ksw_extend2_golden(ctx):
    """Each golden SWExtend call becomes a right-side-only task; cases where extension() would retry with a doubled
    band (MemChainToAlignBatched.scala:858) are left to the oracle-based tests."""
    z = np.load(os.path.join(G, "ksw_extend2.npz"))
    qs, ts = _seqs(z, "q"), _seqs(z, "t")
    checked = 0
    for w, zdrop in sorted({(int(p[0]), int(p[2])) for p in z["params"]}):
        if w > 127:
            continue  # w travels as a signed byte in the wire header
        sel = [i for i, p in enumerate(z["params"]) if int(p[0]) == w and int(p[2]) == zdrop
               and (z["out"][i][0] == int(p[3]) or z["out"][i][5] < (w >> 1) + (w >> 2))]   # first try is final
        if not sel:
            continue
        pool, off = [], []
        for i in sel:
            off.append((len(pool), len(pool) + len(qs[i])))
            pool.extend(qs[i].tolist()); pool.extend(ts[i].tolist())
        n = len(sel)
        zero = np.zeros(n, np.int32)
        h0 = np.array([int(z["params"][i][3]) for i in sel], np.int32)
        soa = bpsw_hip.ExtTaskSoA(left_qlen=zero, left_rlen=zero.copy(), right_qlen=np.array([len(qs[i]) for i in sel], np.int32),
                                  right_rlen=np.array([len(ts[i]) for i in sel], np.int32), left_q_off=np.zeros(n, np.int64),
                                  left_r_off=np.zeros(n, np.int64), right_q_off=np.array([o[0] for o in off], np.int64),
                                  right_r_off=np.array([o[1] for o in off], np.int64), reg_score=h0, q_beg=zero.copy(), h0=h0.copy(),
                                  idx=np.arange(n, dtype=np.int32), pool=np.array(pool + [0], np.uint8), w=w)
        ctx.set_ext_scoring(po.default_mat(), zdrop, bpsw_hip.ZDROP_BWA)
        try:
            got = ctx.extend_batch(bpsw_hip.wire_pack(soa)).reshape(-1, 10)
        finally:
            ctx.set_ext_scoring(po.default_mat(), 100, bpsw_hip.ZDROP_SCALA)
        for k, i in enumerate(sel):
            score, qle, tle, gtle, gscore, _ = (int(v) for v in z["out"][i])
            h = int(h0[k])
            if gscore <= 0 or gscore <= score - 5:      # MemChainToAlignBatched.scala:866-875 with penClip3 = 5
                want = (qle, tle, h + score - h)
            else:
                want = (len(qs[i]), gtle, h + gscore - h)
            assert (int(got[k][3]), int(got[k][5]), int(got[k][6]), int(got[k][7])) == (want[0], want[1], score, want[2]), (i, got[k], z["out"][i])
            checked += 1
    assert checked > 300


@pytest.mark.parametrize("mask", [0, 31, 63])
def test_extension_tasks_vs_reference_golden(ctx, mask):
    """Whole two-sided extension tasks -- band retries, bands 5 / 70 / 100 / 127, 150 and 250 bp, the hand-built long indels that
    double the band to 200 -- through bpsw_extend_batch against the outputs of the reference's own ksw_extend2 under its
    extension() control (tests/golden/ref_extension_tasks.npz, ref_extend_batch): the assembly row loops, the window moves, the
    deferral to the full kernel and the shortcuts (none / all but the sift kernel / all) meet reference outputs directly."""
    from test_golden import _ref_task_sets
    ctx.set_ext_shortcuts(mask)
    try:
        for si, soa, zd, want in _ref_task_sets():
            ctx.set_ext_scoring(po.default_mat(), zd, bpsw_hip.ZDROP_BWA)
            got = ctx.extend_batch(bpsw_hip.wire_pack(soa)).reshape(-1, 10)
            assert np.array_equal(got[:, 2:9].astype(np.int32), want), (mask, si)
            assert np.array_equal(got[:, 0].astype(np.int32) & 0xffff, soa.idx & 0xffff)
    finally:
        ctx.set_ext_scoring(po.default_mat(), 100, bpsw_hip.ZDROP_SCALA)
        ctx.set_ext_shortcuts(-1)


@pytest.mark.parametrize("tag", ["fr", "all4", "250", "250_all4"])
def test_group_rescue_vs_mem_group_matesw_golden(ctx, tag):
    z = np.load(os.path.join(G, f"mem_group_matesw_{tag}.npz"))
    g = bpsw_hip.RescueGroupSoA(group_size=int(z["group_size"]), l_pac=int(z["l_pac"]),
                                pes=[(int(r[0]), int(r[1]), int(r[2]), float(r[3]), float(r[4])) for r in z["pes"]],
                                seq_len=z["seq_len"], seq_off=z["seq_off"], seq_pool=z["seq_pool"], reg_cnt=z["reg_cnt"],
                                regs=z["regs"], ref_cnt=z["ref_cnt"], ref_rb=z["ref_rb"], ref_re=z["ref_re"], ref_len=z["ref_len"],
                                ref_off=z["ref_off"], ref_pool=z["ref_pool"])
    cnt, regs = ctx.matesw_group(bpsw_hip.default_opt(), g, bpsw_hip.RESCUE_C)
    assert np.array_equal(cnt, z["out_cnt"])
    region_fields_equal(regs, z["out_regs"], skip=("csub",))
