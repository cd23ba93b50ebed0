"""Parity of the HIP local-SW kernel with the oracle's restatement of SWUtil.SWAlign2 (SWUtil.scala:417-601)."""
import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth
import pyoracle as po

pytestmark = pytest.mark.gpu

XTRA = po.KSW_XSUBO | po.KSW_XSTART | po.KSW_XBYTE | 19  # MemSamPe.scala:1187-1189


def _check(ctx, orc, jobs, xtra=XTRA, opt=None):
    opt_o = orc.default_opt() if opt is None else opt[0]
    opt_p = bpsw_hip.default_opt() if opt is None else opt[1]
    got = ctx.swalign2_batch(opt_p, xtra, **jobs)
    want, cells = orc.sw_align2_jobs(opt_o, xtra, **jobs)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size}/{len(want)} jobs differ, first {bad[:5]}: got {got[bad[:3]]} want {want[bad[:3]]}"
    return want


@pytest.mark.parametrize("read_len,n,sub", [(100, 300, 0.05), (150, 600, 0.02), (150, 300, 0.10), (250, 200, 0.10)])
def test_swalign2_matches_oracle(ctx, orc, read_len, n, sub):
    jobs = synth.sw_jobs(n, read_len=read_len, sub_rate=sub, seed=500 + read_len)
    want = _check(ctx, orc, jobs)
    assert (want[:, 0] >= 19).sum() > n // 3            # most windows contain the mate
    assert (want[:, 3] >= 0).sum() > 0                  # the second-best logic is exercised


def test_forward_only_and_stop_flags(ctx, orc):
    jobs = synth.sw_jobs(200, read_len=150, seed=11)
    _check(ctx, orc, jobs, xtra=po.KSW_XSUBO | 19)                      # no KSW_XSTART: single pass
    _check(ctx, orc, jobs, xtra=po.KSW_XSTART | po.KSW_XSUBO | 60)      # higher threshold
    _check(ctx, orc, jobs, xtra=po.KSW_XSTOP | 40)                      # early stop, no list
    _check(ctx, orc, jobs, xtra=0)


def _jobs_from(pairs):
    q_len, t_len, q_off, t_off, q_rev, qp, tp = [], [], [], [], [], [], []
    for q, t, rev in pairs:
        q_off.append(len(qp)); t_off.append(len(tp)); q_len.append(len(q)); t_len.append(len(t)); q_rev.append(rev)
        qp.extend(q); tp.extend(t)
        qp.extend([0] * ((-len(qp)) % 16)); tp.extend([0] * ((-len(tp)) % 16))
    return dict(q_len=np.array(q_len, np.int32), t_len=np.array(t_len, np.int32), q_off=np.array(q_off, np.int64),
                t_off=np.array(t_off, np.int64), q_rev=np.array(q_rev, np.uint8), q_pool=np.array(qp + [0] * 16, np.uint8),
                t_pool=np.array(tp + [0] * 16, np.uint8))


def test_edge_cases(ctx, orc):
    rng = np.random.default_rng(5)
    r = lambda n: rng.integers(0, 4, n).tolist()
    q = r(150)
    rc = [3 - b for b in q[::-1]]
    pairs = [
        (q, r(200) + q + r(200), 0),           # exact copy: score 150 < 251
        (rc, r(200) + q + r(200), 1),          # stored reverse-complemented
        (q, q, 0),                             # window == mate
        (q, q[:40], 0),                        # window shorter than the mate
        (q, [], 0),                            # empty window
        (q, r(700), 0),                        # unrelated
        ([4] * 150, r(500), 0),                # all-N mate
        (q, [4] * 500, 0),                     # all-N window
        (q, r(100) + q + r(50) + q[:80] + r(100), 0),   # second copy: score2 / te2
        (q, r(100) + q[70:] + r(30) + q + r(100), 0),   # partial copy before the full one
        (r(1), r(30), 0),                      # one-base mate
        (r(64), r(300), 0), (r(65), r(300), 0), (r(128), r(300), 0), (r(129), r(300), 0),  # lane-block boundaries
        ([0] * 150, [0] * 600, 0),             # homopolymer: every row scores, long best-score list
        ([0, 1] * 75, [0, 1] * 400, 0),        # dinucleotide repeat
    ]
    big = r(300)
    # mates of up to 256 bases: the packed two-jobs-per-wave kernel (an odd number of jobs, so the last wave holds one)
    pk = pairs + [(big[:256], r(50) + big[:256] + r(50), 0),          # 256-base exact copy: score >= 251 -> 255 cap
                  (big[:250], r(2500) + big[:250] + r(30), 0),        # window longer than the LDS staging buffer
                  (q, r(300) + [4] + r(40) + q[:70] + [4] + q[71:] + r(300), 0),   # N inside the window: the patched step
                  (q, [4] * 3 + q + r(600) + [4], 0)]
    assert len(pk) % 2 == 1
    _check(ctx, orc, _jobs_from(pk))
    _check(ctx, orc, _jobs_from(pk), xtra=po.KSW_XSTART | po.KSW_XSUBO | 1)
    _check(ctx, orc, _jobs_from(pk[::-1]))                            # other partners in each wave
    # longer mates: one job per wave, 32-bit
    pairs.append((big, r(50) + big + r(50), 0))           # 300-base exact copy: score >= 251 -> 255 cap
    pairs.append((big[:260], r(2500) + big[:260] + r(30), 0))  # window longer than the LDS staging buffer
    _check(ctx, orc, _jobs_from(pairs))
    _check(ctx, orc, _jobs_from(pairs), xtra=po.KSW_XSTART | po.KSW_XSUBO | 1)


def test_packed_kernel_shapes(ctx, orc):
    """The packed kernel's own boundaries: mate lengths around the columns-per-lane switches (57 lanes x C columns), partners
    of very different mate and window lengths in one wave, windows with N rows, early stop in the forward pass."""
    rng = np.random.default_rng(17)
    r = lambda n: rng.integers(0, 4, n).tolist()
    for top in (57, 114, 171, 228, 256):
        pairs = []
        for ql in (top, top - 1, max(top - 30, 1), 1, top):
            q = r(ql)
            noisy = [b if rng.random() > 0.05 else int(rng.integers(0, 5)) for b in q]
            pairs.append((q, r(int(rng.integers(0, 90))) + noisy + r(int(rng.integers(0, 700))), int(rng.integers(0, 2))))
        pairs.append((r(top), [], 0))
        pairs.append((r(3), r(5), 0))
        jobs = _jobs_from(pairs)
        for xtra in (XTRA, po.KSW_XSTART | po.KSW_XSUBO | 1, po.KSW_XSTOP | 30, po.KSW_XSTART | po.KSW_XSTOP | po.KSW_XSUBO | 25):
            _check(ctx, orc, jobs, xtra=xtra)
    jobs = synth.sw_jobs(801, read_len=150, win_min=200, win_max=900, sub_rate=0.04, indel_rate=0.01, unrelated_frac=0.1,
                         decoy_frac=0.2, rev_frac=0.5, seed=4242)
    tp = jobs["t_pool"].copy()
    tp[rng.integers(0, tp.size, tp.size // 150)] = 4                 # about one N per 150 window bases
    jobs["t_pool"] = tp
    want = _check(ctx, orc, jobs)
    assert (want[:, 3] >= 0).sum() > 20


def test_custom_scoring(ctx, orc):
    jobs = synth.sw_jobs(200, read_len=150, seed=77, indel_rate=0.02)
    oo, op = orc.default_opt(), bpsw_hip.default_opt()
    for o in (oo, op):
        o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins = 2, 3, 5, 2, 4, 1
        m = po.default_mat(2, 3)
        for k in range(25):
            o.mat[k] = int(m[k])
    _check(ctx, orc, jobs, xtra=po.KSW_XSUBO | po.KSW_XSTART | 38, opt=(oo, op))      # packed, different gap opens
    for o in (oo, op):                                                                 # a general matrix, still packed
        m = [1, -2, -3, -4, -1, -2, 2, -4, -3, 0, -3, -4, 1, -2, -1, -4, -3, -2, 2, -2, -1, 0, -1, -2, -1]
        for k in range(25):
            o.mat[k] = m[k]
        o.a, o.b = 2, 4
    _check(ctx, orc, jobs, xtra=po.KSW_XSUBO | po.KSW_XSTART | 30, opt=(oo, op))
    for o in (oo, op):                                                                 # max(mat) > |b| + 1: the 32-bit kernel
        o.a, o.b = 5, 2
        m = po.default_mat(5, 2)
        for k in range(25):
            o.mat[k] = int(m[k])
    _check(ctx, orc, jobs, xtra=po.KSW_XSUBO | po.KSW_XSTART | 95, opt=(oo, op))


def test_rejects_out_of_pool_job(ctx):
    jobs = synth.sw_jobs(10, seed=3)
    jobs["t_off"] = jobs["t_off"].copy(); jobs["t_off"][3] = 10 ** 9
    with pytest.raises(bpsw_hip.BpswError):
        ctx.swalign2_batch(bpsw_hip.default_opt(), XTRA, **jobs)


def test_large_batch(ctx, orc):
    """26 003 jobs of 150-base mates with ragged windows, every job vs the oracle: with the default scoring through the packed
    kernel (grid-stride over more pairs than resident waves, the last wave with a single job); with a match score the 16-bit
    halves cannot hold (max(mat) > |b| + 1) launch_sw_kernel picks sw4_kernel (four jobs per wavefront) by itself."""
    jobs = synth.sw_jobs(26003, read_len=150, win_min=300, win_max=650, sub_rate=0.03, indel_rate=0.004, unrelated_frac=0.1,
                         decoy_frac=0.15, rev_frac=0.5, seed=909)
    want = _check(ctx, orc, jobs)
    assert (want[:, 3] >= 0).sum() > 100 and (want[:, 0] < 19).sum() > 100
    oo, op = orc.default_opt(), bpsw_hip.default_opt()
    for o in (oo, op):
        o.a, o.b = 3, 1
        m = po.default_mat(3, 1)
        for k in range(25):
            o.mat[k] = int(m[k])
    want = _check(ctx, orc, jobs, xtra=po.KSW_XSUBO | po.KSW_XSTART | 57, opt=(oo, op))
    assert (want[:, 0] == 255).sum() > 100
