"""Every short flank over a two-letter alphabet, as extension tasks (shared by tests/test_extend_exhaustive_gpu.py and
tools/): queries of 1..max_q bases (first base fixed: the scoring matrices are symmetric under relabelling the bases),
targets of 1..qLen+3 bases, all of them; optionally the variants with an N at one query or one target position (short flanks
only).  Sequences are stored once in the pool and shared by the tasks that use them."""
from __future__ import annotations

import numpy as np

import bpsw_hip


def _all_seqs(n: int, first_fixed: bool) -> np.ndarray:
    """all sequences of length n over {0,1} (first base 0 when first_fixed), one per row"""
    free = n - 1 if first_fixed else n
    codes = np.arange(1 << free, dtype=np.int64)
    bits = ((codes[:, None] >> np.arange(free)[None, :]) & 1).astype(np.uint8)
    return np.concatenate([np.zeros((bits.shape[0], 1), np.uint8), bits], axis=1) if first_fixed else bits


def enumerate_flanks(max_q: int = 7, extra_t: int = 3, n_variants_up_to: int = 5):
    """returns (pool, q_off, q_len, t_off, t_len): one entry per flank"""
    pool, q_off, q_len, t_off, t_len = [], [], [], [], []
    at = 0

    def add(rows: np.ndarray) -> np.ndarray:   # store the rows back to back, return their offsets
        nonlocal at
        off = at + np.arange(rows.shape[0], dtype=np.int64) * rows.shape[1]
        pool.append(rows.reshape(-1))
        at += rows.size
        return off

    for ql in range(1, max_q + 1):
        qs = _all_seqs(ql, True)
        variants_q = [qs]
        if ql <= n_variants_up_to:
            for p in range(ql):
                v = qs.copy(); v[:, p] = 4
                variants_q.append(v)
        for tl in range(1, ql + extra_t + 1):
            ts = _all_seqs(tl, False)
            variants_t = [ts]
            if ql <= n_variants_up_to:
                for p in range(tl):
                    v = ts.copy(); v[:, p] = 4
                    variants_t.append(v)
            t_offs = [add(v) for v in variants_t]
            for qi, qv in enumerate(variants_q):
                qo = add(qv)
                for ti, to in enumerate(t_offs):
                    if qi > 0 and ti > 0:
                        continue   # an N in the query OR in the target, not both
                    a, b = np.meshgrid(qo, to, indexing="ij")
                    q_off.append(a.reshape(-1)); t_off.append(b.reshape(-1))
                    q_len.append(np.full(a.size, ql, np.int32)); t_len.append(np.full(a.size, tl, np.int32))
    return (np.concatenate(pool + [np.zeros(16, np.uint8)]), np.concatenate(q_off), np.concatenate(q_len), np.concatenate(t_off),
            np.concatenate(t_len))


def flank_tasks(pool, q_off, q_len, t_off, t_len, h0: int, left: bool) -> "bpsw_hip.ExtTaskSoA":
    """one task per flank, the flank on the left or on the right side of a seed of score h0, the other side empty"""
    n = q_off.shape[0]
    z32, z64 = np.zeros(n, np.int32), np.zeros(n, np.int64)
    hh = np.full(n, h0, np.int32)
    kw = dict(reg_score=hh, h0=hh, idx=np.arange(n, dtype=np.int32), pool=pool)
    if left:
        kw.update(left_qlen=q_len, left_rlen=t_len, left_q_off=q_off, left_r_off=t_off, right_qlen=z32, right_rlen=z32,
                  right_q_off=z64, right_r_off=z64, q_beg=q_len.copy())
    else:
        kw.update(right_qlen=q_len, right_rlen=t_len, right_q_off=q_off, right_r_off=t_off, left_qlen=z32, left_rlen=z32,
                  left_q_off=z64, left_r_off=z64, q_beg=z32)
    return bpsw_hip.ExtTaskSoA(**kw)
