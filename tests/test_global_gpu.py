"""HIP banded global alignment + CIGAR (SWUtil.SWGlobal, SWUtil.scala:233-397) against the ksw_global2 golden
vectors (reference C) and against the oracle restatement on seeded and edge-case jobs.  Bit-exact score and CIGAR."""
import os

import numpy as np
import pytest

import bpsw_hip
import pyoracle as po

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAT = po.default_mat()


def _run(ctx, qs, ts, ws, max_cigar=128, opt=None):
    q_off, t_off, qp, tp = [], [], [], []
    for q, t in zip(qs, ts):
        q_off.append(sum(len(x) for x in qp)); t_off.append(sum(len(x) for x in tp))
        qp.append(np.concatenate([np.asarray(q, np.uint8), np.zeros((-len(q)) % 16, np.uint8)]))
        tp.append(np.concatenate([np.asarray(t, np.uint8), np.zeros((-len(t)) % 16, np.uint8)]))
    return ctx.global_batch(opt or bpsw_hip.default_opt(), [len(q) for q in qs], [len(t) for t in ts], ws, q_off, t_off,
                            np.concatenate(qp + [np.zeros(16, np.uint8)]), np.concatenate(tp + [np.zeros(16, np.uint8)]), max_cigar)


def test_vs_ksw_global2_golden(ctx):
    z = np.load(os.path.join(G, "ksw_global2.npz"))
    qs = [z["q_pool"][z["q_off"][i]:z["q_off"][i + 1]] for i in range(len(z["w"]))]
    ts = [z["t_pool"][z["t_off"][i]:z["t_off"][i + 1]] for i in range(len(z["w"]))]
    score, ncig, cig = _run(ctx, qs, ts, z["w"])
    assert np.array_equal(score, z["score"])
    for i in range(len(qs)):
        want = z["cig_pool"][z["cig_off"][i]:z["cig_off"][i + 1]]
        assert ncig[i] == len(want) and np.array_equal(cig[i, : ncig[i]], want), i


def test_vs_oracle_seeded_and_edges(ctx, orc):
    rng = np.random.default_rng(8)
    qs, ts, ws = [], [], []
    for n in range(300):
        ql = int(rng.integers(1, 300))
        q = rng.integers(0, 5 if n % 11 == 0 else 4, ql).astype(np.uint8)
        t = list(q)
        for _ in range(int(rng.integers(0, 6))):          # a few indels / substitutions
            p = int(rng.integers(0, max(len(t), 1)))
            r = rng.random()
            if r < 0.3 and len(t) > 1:
                del t[p:p + int(rng.integers(1, 6))]
            elif r < 0.6:
                t[p:p] = rng.integers(0, 4, int(rng.integers(1, 6))).tolist()
            elif len(t):
                t[p] = (t[p] + 1) % 4
        if not t:
            t = [0]
        w = abs(len(t) - ql) + int(rng.integers(0, 50)) + (3 if n % 5 else 0)
        qs.append(q); ts.append(np.array(t, np.uint8)); ws.append(w)
    # edges: band wider than the query (nCol = qLen), w = 0 with equal lengths, one-base sequences, long target
    q = rng.integers(0, 4, 40).astype(np.uint8)
    qs += [q, q, np.array([2], np.uint8), q[:5], rng.integers(0, 4, 150).astype(np.uint8)]
    ts += [q, q, np.array([2], np.uint8), rng.integers(0, 4, 60).astype(np.uint8), rng.integers(0, 4, 400).astype(np.uint8)]
    ws += [100, 0, 3, 60, 260]
    score, ncig, cig = _run(ctx, qs, ts, ws, max_cigar=512)
    for i, (q, t, w) in enumerate(zip(qs, ts, ws)):
        ws_, wc = orc.sw_global(q, t, MAT, 6, 1, 6, 1, int(w))
        assert score[i] == ws_, (i, score[i], ws_)
        assert ncig[i] == len(wc) and np.array_equal(cig[i, : ncig[i]], wc), (i, cig[i, : ncig[i]], wc)


def test_cigar_capacity_is_reported_not_truncated_silently(ctx, orc):
    q = np.array([0, 1] * 30, np.uint8)
    t = np.concatenate([q[:10], [3, 3, 3], q[10:20], q[25:40], [3, 3], q[40:]]).astype(np.uint8)
    ws_, wc = orc.sw_global(q, t, MAT, 6, 1, 6, 1, 20)
    score, ncig, cig = _run(ctx, [q], [t], [20], max_cigar=2)
    assert score[0] == ws_ and ncig[0] == len(wc) and len(wc) > 2        # count reported, caller must resubmit
    score, ncig, cig = _run(ctx, [q], [t], [20], max_cigar=64)
    assert np.array_equal(cig[0, : ncig[0]], wc)
