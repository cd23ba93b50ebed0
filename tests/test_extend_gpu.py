"""Parity of the HIP extension kernel (boundary 2) with the oracle's restatement of the Scala extension()
(MemChainToAlignBatched.scala:789-883 over SWUtil.scala:61-230).  Bit-exact: int16 results compared with ==."""
import os

import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth
import pyoracle as po

pytestmark = pytest.mark.gpu


def _check(ctx, orc, soa, zmode=po.ZDROP_SCALA, zdrop=100, mat=None):
    wire = bpsw_hip.wire_pack(soa)
    mat = po.default_mat() if mat is None else mat
    ctx.set_ext_scoring(mat, zdrop, zmode)
    try:
        got = ctx.extend_batch(wire)
    finally:
        ctx.set_ext_scoring(po.default_mat(), 100, bpsw_hip.ZDROP_SCALA)
    want, cells = orc.wire_extend(wire, mat, zdrop, zmode)
    assert got.shape == want.shape
    bad = np.nonzero((got != want).reshape(-1, 10).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size}/{soa.n} tasks differ, first {bad[:5]}: got {got.reshape(-1,10)[bad[:3]]} want {want.reshape(-1,10)[bad[:3]]}"
    return cells


@pytest.mark.parametrize("read_len,sub,indel,tail,n", [
    (100, 0.01, 0.001, 0.0, 1000),   # config 1 shape (phiX-like SE 100 bp)
    (150, 0.01, 0.001, 0.0, 4000),   # configs 2-4 (2x150 bp)
    (150, 0.05, 0.005, 0.0, 2000),
    (250, 0.08, 0.02, 0.05, 2000),   # config 5 (2x250 bp, high error, band doubling)
])
def test_extension_matches_oracle(ctx, orc, read_len, sub, indel, tail, n):
    soa = synth.ext_tasks(n, read_len=read_len, sub_rate=sub, indel_rate=indel, tail_frac=tail, seed=42 + read_len)
    cells = _check(ctx, orc, soa)
    assert cells > 0


def test_bwa_zdrop_mode_matches_oracle(ctx, orc):
    soa = synth.ext_tasks(1500, read_len=250, sub_rate=0.1, indel_rate=0.03, tail_frac=0.3, seed=7)
    _check(ctx, orc, soa, zmode=po.ZDROP_BWA)
    _check(ctx, orc, soa, zmode=po.ZDROP_SCALA, zdrop=20)  # small zdrop forces the z-drop branch
    _check(ctx, orc, soa, zmode=po.ZDROP_BWA, zdrop=20)


def _manual_tasks(tasks):
    """tasks: list of (leftQ, leftR, rightQ, rightR, h0, qBeg) as python lists of codes"""
    pool, f = [], {k: [] for k in ("left_qlen", "left_rlen", "right_qlen", "right_rlen", "left_q_off", "left_r_off",
                                     "right_q_off", "right_r_off", "reg_score", "q_beg", "h0", "idx")}
    for i, (lq, lr, rq, rr, h0, qb) in enumerate(tasks):
        for name, seq in (("left_q", lq), ("left_r", lr), ("right_q", rq), ("right_r", rr)):
            f[name + "_off"].append(len(pool))
            f[name.replace("_q", "_qlen").replace("_r", "_rlen")].append(len(seq))
            pool.extend(seq)
        f["reg_score"].append(h0); f["h0"].append(h0); f["q_beg"].append(qb); f["idx"].append(i)
    kw = {k: np.array(v, np.int64 if k.endswith("_off") else np.int32) for k, v in f.items()}
    return bpsw_hip.ExtTaskSoA(pool=np.array(pool + [0], np.uint8), **kw)


def test_edge_cases(ctx, orc):
    rng = np.random.default_rng(3)
    r = lambda n: rng.integers(0, 4, n).tolist()
    q60 = r(60)
    tasks = [
        ([], [], q60, q60 + r(50), 30, 0),                 # right side only, perfect match
        (q60, q60 + r(50), [], [], 30, 60),                # left side only
        (r(1), r(3), r(1), r(2), 19, 1),                   # one-base sides
        (r(40), r(80), r(40), r(80), 25, 40),              # unrelated flanks: m == 0 / early stop
        ([4] * 30, r(60), [4] * 30, r(60), 50, 30),        # all-N query
        (q60, [4] * 100, q60, [4] * 100, 50, 60),          # all-N reference
        (q60[:30], q60[:30], q60[30:], q60[30:], 100, 30), # reference exactly as long as the query
        (q60, q60[:20], q60, q60[:10], 40, 60),            # reference shorter than the query
        (r(131), r(262), r(131), r(262), 19, 131),         # longest 150-bp sides (> 2 chunks)
        (q60 + r(71), q60 + r(140), [], [], 200, 131),     # high h0: wide band of positive H
    ]
    # a long seed followed by unrelated sequence after a good stretch: the z-drop region
    good = r(100)
    tasks.append((good + r(120), good + r(200), good + r(120), good + r(200), 150, 220))
    soa = _manual_tasks(tasks)
    for zmode in (po.ZDROP_SCALA, po.ZDROP_BWA):
        _check(ctx, orc, soa, zmode=zmode)
        _check(ctx, orc, soa, zmode=zmode, zdrop=10)


def test_single_task_and_ragged_batch_sizes(ctx, orc):
    soa = synth.ext_tasks(700, read_len=150, seed=99)
    for n in (1, 2, 3, 5, 63, 64, 65, 257):
        _check(ctx, orc, soa.subset(np.arange(n)))


def test_custom_scoring_matrix(ctx, orc):
    soa = synth.ext_tasks(800, read_len=150, sub_rate=0.04, indel_rate=0.01, seed=123)
    mat = po.default_mat(2, 3)
    soa.mat_max = 2
    _check(ctx, orc, soa, mat=mat)


def test_rejects_malformed_batches(ctx):
    soa = synth.ext_tasks(50, seed=1)
    wire = bpsw_hip.wire_pack(soa)
    with pytest.raises(bpsw_hip.BpswError):
        ctx.extend_batch(wire[: 32 + 32 * soa.n - 4])       # table truncated
    bad = wire.copy(); bad[32 + 8: 32 + 12] = np.frombuffer(np.int32(10 ** 8).tobytes(), np.uint8)  # sequence offset outside
    with pytest.raises(bpsw_hip.BpswError):
        ctx.extend_batch(bad)
    empty = wire[:32].copy(); empty[8:12] = 0
    assert ctx.extend_batch(empty).size == 0                # empty batch is fine
    zero_ext = wire.copy(); zero_ext[1] = 0                 # eDel = 0: (qLen*max + bonus - oDel) / eDel is a division by zero in the Scala
    with pytest.raises(bpsw_hip.BpswError):
        ctx.extend_batch(zero_ext)


def test_device_resident_entry_matches_host_entry(ctx, orc):
    import torch
    soa = synth.ext_tasks(3000, read_len=150, seed=2024)
    wire = bpsw_hip.wire_pack(soa)
    want = ctx.extend_batch(wire)
    d_wire = torch.from_numpy(wire.copy()).to("cuda:0")
    d_out = torch.zeros(10 * soa.n, dtype=torch.int16, device="cuda:0")
    torch.cuda.synchronize()
    ctx.extend_batch_device(d_wire.data_ptr(), wire.size, soa.n, d_out.data_ptr())
    ext_ms, _ = ctx.last_kernel_ms()
    torch.cuda.synchronize()
    assert ext_ms > 0
    assert np.array_equal(d_out.cpu().numpy(), want)


def test_near_exact_flanks_closed_form_boundaries(ctx, orc):
    """The closed form for near-exact flanks (bpsw_extend_core.h, flank_closed_form): flanks built around its validity
    boundary -- diagonal deficit just below / at / above a gap open, h0 <= deficit, repeats that offer gapped alternatives,
    N on either strand, reference exactly as long as the query -- under several gap costs, bands, z-drops and both parses.
    The kernel must agree with the oracle's full DP whichever path it takes."""
    rng = np.random.default_rng(11)

    def flank(kind, n):
        if kind == 0:
            return rng.integers(0, 4, n)
        if kind == 1:
            return np.full(n, rng.integers(0, 4))                      # homopolymer: every shifted diagonal matches too
        return np.tile(rng.integers(0, 4, int(rng.integers(1, 4))), n)[:n]   # tandem repeat

    tasks = []
    for t in range(1500):
        sides = []
        for _ in range(2):
            n = int(rng.integers(1, 132))
            q = flank(t % 3, n).astype(np.int64)
            r = q.copy()
            for _ in range(int(rng.integers(0, 4))):                  # 0..3 defects, often on the first base (as after a maximal seed)
                p = 0 if rng.random() < 0.5 else int(rng.integers(0, n))
                u = rng.random()
                if u < 0.2:
                    r[p] = 4
                elif u < 0.3:
                    q[p] = 4
                else:
                    r[p] = (r[p] + 1 + rng.integers(0, 3)) & 3
            extra = int(rng.integers(0, 80)) if rng.random() < 0.9 else 0
            tail = rng.integers(0, 5, extra) if rng.random() < 0.5 else np.tile(q, 2)[:extra]
            sides.append((q.tolist(), np.concatenate([r, tail]).astype(np.int64).tolist()))
        h0 = int(rng.integers(1, 12)) if t % 7 == 0 else int(rng.integers(19, 140))     # h0 <= deficit now and then
        if t % 11 == 0:
            sides[0] = ([], [])
        tasks.append((sides[0][0], sides[0][1], sides[1][0], sides[1][1], h0, len(sides[0][0])))
    soa = _manual_tasks(tasks)
    for (od, ed, oi, ei), w in (((6, 1, 6, 1), 100), ((0, 1, 0, 1), 100), ((3, 2, 7, 1), 2), ((1, 1, 1, 1), 1), ((4, 2, 2, 3), 5)):
        soa.o_del, soa.e_del, soa.o_ins, soa.e_ins, soa.w = od, ed, oi, ei, w
        for zmode, zdrop in ((po.ZDROP_SCALA, 100), (po.ZDROP_BWA, 100), (po.ZDROP_BWA, 3), (po.ZDROP_SCALA, 0)):
            _check(ctx, orc, soa, zmode=zmode, zdrop=zdrop)
    soa.o_del, soa.e_del, soa.o_ins, soa.e_ins, soa.w = 6, 1, 6, 1, 100
    soa.mat_max = 3
    _check(ctx, orc, soa, mat=po.default_mat(3, 2))                     # deficit of a substitution = 5 with a = 3
    asym = po.default_mat(1, 4)
    asym[1 * 5 + 1] = 2                                                 # unequal match scores: the closed form must switch itself off
    soa.mat_max = 2
    _check(ctx, orc, soa, mat=asym)


def test_tail_row_bound_cases(ctx, orc):
    """Rows past the query end are skipped once they provably cannot change the result (bpsw_extend_core.h, tail_row_bound).
    Flanks that stress the bound: the query occurs again later in the target (an alignment restarted from a zero cell can
    out-score everything before it), long deletions (the best alignment ends far below row qLen), low h0, eDel = 0, repeats,
    and scoring matrices with a larger or uneven maximum."""
    rng = np.random.default_rng(23)

    def mutate(seq, sub, indel):
        out, i = [], 0
        while i < len(seq):
            u = rng.random()
            if u < indel / 2:
                out.append(int(rng.integers(0, 4)))
            elif u < indel:
                i += 1
            elif u < indel + sub:
                out.append(int((seq[i] + 1 + rng.integers(0, 3)) & 3)); i += 1
            else:
                out.append(int(seq[i])); i += 1
        return out

    tasks = []
    for t in range(1200):
        sides = []
        for _ in range(2):
            n = int(rng.integers(1, 125))
            base = rng.integers(0, 4, n + 200)
            if t % 6 == 1:
                base = np.full(n + 200, rng.integers(0, 4))
            if t % 6 == 2:
                base = np.tile(rng.integers(0, 4, int(rng.integers(1, 5))), n + 200)[: n + 200]
            err = (0.01, 0.05, 0.15, 0.3)[t % 4]
            q = mutate(base[: n + 30].tolist(), err, err / 3)[:n]
            if len(q) < n:
                q = q + rng.integers(0, 4, n - len(q)).tolist()
            tl = n + int(rng.integers(0, n + 60))
            r = base[:tl].tolist()
            if t % 6 == 3:      # unrelated sequence, then the query again: restart from zero
                r = rng.integers(0, 4, int(rng.integers(5, 70))).tolist() + q + rng.integers(0, 4, 20).tolist()
            if t % 6 == 4:      # a block the read does not have: long deletion
                p = int(rng.integers(0, n))
                r = base[:p].tolist() + rng.integers(0, 4, int(rng.integers(1, 45))).tolist() + base[p:tl].tolist()
            if rng.random() < 0.1 and r:
                r[int(rng.integers(0, len(r)))] = 4
            sides.append((q, r))
        h0 = int(rng.integers(1, 15)) if t % 5 == 0 else int(rng.integers(19, 140))
        tasks.append((sides[0][0], sides[0][1], sides[1][0], sides[1][1], h0, len(sides[0][0])))
    soa = _manual_tasks(tasks)
    for (od, ed, oi, ei), w in (((6, 1, 6, 1), 100), ((0, 1, 0, 1), 100), ((3, 2, 7, 1), 30), ((4, 1, 2, 3), 100), ((1, 1, 1, 1), 5)):
        soa.o_del, soa.e_del, soa.o_ins, soa.e_ins, soa.w = od, ed, oi, ei, w
        for zmode, zdrop in ((po.ZDROP_SCALA, 100), (po.ZDROP_BWA, 100), (po.ZDROP_BWA, 20), (po.ZDROP_SCALA, 0)):
            _check(ctx, orc, soa, zmode=zmode, zdrop=zdrop)
    soa.o_del, soa.e_del, soa.o_ins, soa.e_ins, soa.w = 6, 1, 6, 1, 100
    soa.mat_max = 3
    _check(ctx, orc, soa, mat=po.default_mat(3, 2))
    uneven = po.default_mat(1, 4)
    uneven[2 * 5 + 2] = 3                                               # one base scores more: amax = 3 bounds all of them
    _check(ctx, orc, soa, mat=uneven)


def test_two_gap_open_deficits(ctx, orc):
    """Flanks whose main diagonal has a deficit of two gap opens (+1) -- three substitutions under the default scoring, the
    closed form's second extension (bpsw_extend_core.h, "Two gap opens") -- in homopolymers and short-period repeats, with the
    substituted bases and the target tail chosen so that the shifted diagonals of its two exclusion tests match far more often
    than by chance; and flanks with a one-base gap at their start (flank_start_gap_form), with periodic sequence, small h0 and
    extra differences that must make the form step back; three scorings of the family, a wide and a minimal band, both z-drop
    parses.  (tools/soak_cert2.py is the long form of this test.)"""
    rng = np.random.default_rng(20261003)

    def side():
        n = int(rng.integers(8, 129))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            q = rng.integers(0, 4, n)
        elif kind == 1:
            q = np.full(n, rng.integers(0, 4)); q[rng.integers(0, n, max(1, n // 12))] = rng.integers(0, 4)
        elif kind == 2:
            q = np.tile(rng.integers(0, 4, int(rng.integers(1, 5))), n)[:n]
        else:
            q = rng.integers(0, 2, n)
        q = q.astype(np.int64); r = q.copy()
        u = rng.random()
        n_sub, n_n = (3, 0) if u < 0.65 else ((2, 2) if u < 0.85 else (1, 5))
        start, span = int(rng.integers(0, n)), (int(rng.integers(2, 40)) if rng.random() < 0.6 else n)
        pos = sorted(set(int(min(n - 1, start + rng.integers(0, span))) for _ in range(n_sub + n_n)))
        rng.shuffle(pos)
        for k, p in enumerate(pos):
            if k < n_sub:
                cand = int(q[p + int(rng.integers(1, min(4, n - p)))]) if rng.random() < 0.5 and p + 2 < n else -1
                if cand < 0 or cand == q[p] or cand > 3:
                    cand = int((q[p] + 1 + rng.integers(0, 3)) & 3)
                r[p] = cand
            elif rng.random() < 0.5:
                r[p] = 4
            else:
                q[p] = 4
        extra = int(rng.integers(0, 40))
        v = rng.random()
        if v < 0.4:
            tail = np.concatenate([rng.integers(0, 4, int(rng.integers(1, 4))), np.tile(q[-1:], extra + 3)])[: extra + 3]
        elif v < 0.7:
            tail = np.tile(q, 2)[:extra]
        else:
            tail = rng.integers(0, 5, extra)
        return q.tolist(), np.concatenate([r, tail]).astype(np.int64).tolist()

    def gap_side():   # a one-base gap at (or near) the start of the flank and (nearly) nothing else: flank_start_gap_form
        n = int(rng.integers(5, 132))
        kind = int(rng.integers(0, 3))
        q = (rng.integers(0, 4, n) if kind == 0 else np.tile(rng.integers(0, 4, int(rng.integers(1, 4))), n)[:n] if kind == 1
             else rng.integers(0, 2, n)).astype(np.int64)
        L = 1 if rng.random() < 0.8 else int(rng.integers(2, 4))
        p = 0 if rng.random() < 0.8 else int(rng.integers(1, 6))
        r = np.concatenate([q[:p], q[p + L:]]) if rng.random() < 0.5 else np.concatenate([q[:p], rng.integers(0, 4, L), q[p:]])
        r = r.copy()
        if rng.random() < 0.25 and len(r) > 2:
            x = int(rng.integers(0, len(r))); r[x] = (r[x] + 1 + rng.integers(0, 3)) & 3
        extra = int(rng.integers(0, 30))
        tail = rng.integers(0, 5, extra) if rng.random() < 0.5 else np.tile(q[-3:], extra)[:extra]
        return q.tolist(), np.concatenate([r, tail]).astype(np.int64).tolist()

    tasks = []
    for _ in range(3000):
        l = gap_side() if rng.random() < 0.35 else side()
        r = gap_side() if rng.random() < 0.35 else side()
        tasks.append((l[0], l[1], r[0], r[1], int(rng.integers(3, 30)) if rng.random() < 0.2 else int(rng.integers(16, 150)), len(l[0])))
    soa = _manual_tasks(tasks)
    for (od, ed, oi, ei), w in (((6, 1, 6, 1), 100), ((6, 1, 6, 1), 2), ((1, 1, 1, 1), 100), ((3, 1, 3, 1), 100)):
        soa.o_del, soa.e_del, soa.o_ins, soa.e_ins, soa.w = od, ed, oi, ei, w
        for zmode, zdrop in ((po.ZDROP_SCALA, 100), (po.ZDROP_BWA, 16)):
            _check(ctx, orc, soa, zmode=zmode, zdrop=zdrop)


def _flank(rng, n, sub, indel):
    """query of n bases and a target made from it (substitutions / indels) with a random tail"""
    q = rng.integers(0, 4, n).tolist()
    t = []
    for b in q:
        u = rng.random()
        if u < indel / 2:
            continue
        if u < indel:
            t.append(int(rng.integers(0, 4)))
        t.append(int((b + 1 + rng.integers(0, 3)) & 3) if rng.random() < sub else int(b))
    return q, t + rng.integers(0, 4, int(rng.integers(20, 140))).tolist()


@pytest.mark.parametrize("mid_share,w", [(0.0, 100), (0.03, 100), (0.3, 100), (0.9, 100), (0.3, 70), (0.9, 127), (0.03, 5)])
def test_launch_plans_of_the_split_kernel(ctx, orc, mid_share, w):
    """a batch runs on the 48-VGPR build without the window plus the full kernel for its listed tasks (few flanks of
    128-255 bases), or on the window build, which defers on the device what it cannot finish -- a band wider than 128 columns: the
    doubled band of a retry (w = 70 -> 140, 127 -> 254) or a first row of min(qLen, w + 1) > 127 columns -- or, with flanks above
    255 bases in the batch, also on the full kernel for the host's list.  Every mix must give the oracle's results."""
    rng = np.random.default_rng(int(1000 * mid_share) + w)
    tasks = []
    for t in range(1500):
        u = rng.random()
        if u < mid_share:
            n1, n2 = int(rng.integers(128, 256)), int(rng.integers(1, 256))
        elif u < mid_share + 0.02:
            n1, n2 = int(rng.integers(256, 400)), int(rng.integers(1, 200))       # the full kernel's own (LDS-row sweep)
        else:
            n1, n2 = int(rng.integers(1, 128)), int(rng.integers(1, 128))
        if rng.random() < 0.5:
            n1, n2 = n2, n1
        sub = float(rng.choice([0.0, 0.02, 0.08, 0.2]))
        indel = float(rng.choice([0.0, 0.01, 0.04]))
        lq, lr = _flank(rng, n1, sub, indel)
        rq, rr = _flank(rng, n2, sub, indel)
        tasks.append((lq, lr, rq, rr, int(rng.choice([19, 30, 60, 120])), n1))
    soa = _manual_tasks(tasks)
    soa.w = w
    for zmode in (po.ZDROP_SCALA, po.ZDROP_BWA):
        _check(ctx, orc, soa, zmode=zmode)


@pytest.mark.parametrize("w", [70, 100, 127])
def test_many_long_flanks_are_swept_where_they_are(ctx, orc, w):
    """A batch in which more than one task in sixteen has a flank of 128-255 bases (2x250 bp reads): the short kernel sweeps a band
    wider than 128 columns itself (round 4: with the full kernel's slot sweep, in a build of its own; since round 5 in the
    four-columns-per-lane phase of the adaptive sweep, bpsw_extend_rows.h rows_cpp4) -- no list entry, no second launch.  Retries
    that double the band to 140 / 200 / 254 columns, both parses, against the oracle."""
    rng = np.random.default_rng(500 + w)
    tasks = []
    for t in range(1200):
        if rng.random() < 0.5:
            n1, n2 = int(rng.integers(128, 256)), int(rng.integers(1, 256))
        else:
            n1, n2 = int(rng.integers(1, 128)), int(rng.integers(1, 128))
        if rng.random() < 0.5:
            n1, n2 = n2, n1
        sub = float(rng.choice([0.0, 0.02, 0.08, 0.2]))
        indel = float(rng.choice([0.0, 0.01, 0.04]))
        lq, lr = _flank(rng, n1, sub, indel)
        rq, rr = _flank(rng, n2, sub, indel)
        tasks.append((lq, lr, rq, rr, int(rng.choice([19, 30, 60, 120, 200])), n1))
    soa = _manual_tasks(tasks)
    soa.w = w
    before = ctx.stats().ext_full_relaunches
    for zmode in (po.ZDROP_SCALA, po.ZDROP_BWA):
        _check(ctx, orc, soa, zmode=zmode)
    assert ctx.stats().ext_full_relaunches == before


@pytest.mark.parametrize("w,must_defer", [(100, False), (70, False), (127, False)])
def test_wide_bands_of_mid_flanks_are_swept_in_place(ctx, orc, w, must_defer):
    """Flanks of 128-255 bases whose band outgrows the 128-column window (w = 70 / 127: the doubled band of a retry; w = 127: the first
    row) were deferred to the full kernel through round 4 -- a second launch behind the short kernel (bpsw_runtime.cpp, lazy_full).
    Since round 5 the adaptive sweep goes on four columns per lane (bpsw_extend_rows.h, rows_cpp4): same results as the oracle, and
    no second launch (bpsw_stats_t::ext_full_relaunches stays where it is).  The late launch itself is still exercised by
    test_an_unexpected_band_overflow_is_deferred_not_trapped."""
    rng = np.random.default_rng(77 + w)
    tasks = []
    for t in range(2400):
        if rng.random() < 0.04:
            n1, n2 = int(rng.integers(128, 256)), int(rng.integers(1, 128))
        else:
            n1, n2 = int(rng.integers(1, 128)), int(rng.integers(1, 128))
        if rng.random() < 0.5:
            n1, n2 = n2, n1
        sub = float(rng.choice([0.0, 0.02, 0.08, 0.2]))
        indel = float(rng.choice([0.0, 0.01, 0.04]))
        lq, lr = _flank(rng, n1, sub, indel)
        rq, rr = _flank(rng, n2, sub, indel)
        tasks.append((lq, lr, rq, rr, int(rng.choice([19, 30, 60, 120])), n1))
    soa = _manual_tasks(tasks)
    soa.w = w
    before = ctx.stats().ext_full_relaunches
    for zmode in (po.ZDROP_SCALA, po.ZDROP_BWA):
        _check(ctx, orc, soa, zmode=zmode)
    late = ctx.stats().ext_full_relaunches - before
    assert late == 0 and not must_defer
    # a batch with nothing between 128 and 255 bases never has a list
    short = _manual_tasks([t for t in tasks if max(len(t[0]), len(t[2])) < 128])
    short.w = w
    before = ctx.stats().ext_full_relaunches
    _check(ctx, orc, short)
    assert ctx.stats().ext_full_relaunches == before


@pytest.mark.parametrize("sub,indel,n_rate", [(0.005, 0.0005, 0.0), (0.01, 0.001, 0.0), (0.02, 0.004, 0.002), (0.05, 0.01, 0.0)])
def test_sift_kernel_changes_where_the_shortcuts_run_not_what_they_decide(ctx, orc, sub, indel, n_rate):
    """csrc/bpsw_extend_sift.hip evaluates the exact shortcuts one task per lane in front of ext_kernel (bit 5 of
    bpsw_set_ext_shortcuts): results AND the per-side verdicts (shortcut / DP swept) equal those of ext_kernel alone, at every
    certificate level; tasks with N bases, empty sides, flanks above the sift's 127 bases and custom gap costs ride along"""
    rng = np.random.default_rng(int(sub * 1e4) + 5)
    soa = synth.ext_tasks(6000, read_len=150, sub_rate=sub, indel_rate=indel, seed=900 + int(sub * 1e4))
    if n_rate:
        pool = soa.pool.copy()
        pool[rng.random(pool.size) < n_rate] = 4
        soa.pool = pool
    wire = bpsw_hip.wire_pack(soa)
    want, _ = orc.wire_extend(wire)
    try:
        for level in (1, 1 | 2, 1 | 2 | 4, 31):
            ctx.set_ext_shortcuts(level)
            out0, how0 = ctx.extend_batch_classify(wire)
            ctx.set_ext_shortcuts(level | 32)
            out1, how1 = ctx.extend_batch_classify(wire)
            assert np.array_equal(out0, want) and np.array_equal(out1, want), level
            assert np.array_equal(how0, how1), level
        assert sub > 0.02 or (how1 == 1).sum() > 0.3 * (how1 != 0).sum()      # the batch does exercise the forms
    finally:
        ctx.set_ext_shortcuts(-1)
    # other gap costs / a matrix with another mismatch score (dm = 4) / one with two mismatch scores (no sift: same results)
    for (mat, o, e) in ((po.default_mat(), 4, 2), (_uniform_mat(1, 3), 5, 1), (_two_score_mat(), 6, 1)):
        soa.o_del = soa.o_ins = o
        soa.e_del = soa.e_ins = e
        _check(ctx, orc, soa, mat=mat)


def _uniform_mat(a, b):
    m = np.full((5, 5), -1, np.int8)
    m[:4, :4] = -b
    for i in range(4):
        m[i, i] = a
    return m.reshape(-1)


def _two_score_mat():
    m = _uniform_mat(1, 4).reshape(5, 5)
    m[0, 2] = m[2, 0] = -2      # transitions cheaper than transversions
    return m.reshape(-1)


def test_sift_kernel_leaves_what_it_cannot_stage_to_the_extension_kernel(ctx, orc):
    """the sift kernel copies the nibble streams of 64 consecutive tasks as one run: a batch whose task table is not in stream order
    (legal: every record carries its own offset), or with tasks too long for its buffer between short ones, must come out the same"""
    soa = synth.ext_tasks(5000, read_len=150, sub_rate=0.01, indel_rate=0.001, seed=4711)
    wire = bpsw_hip.wire_pack(soa)
    want, _ = orc.wire_extend(wire)
    assert np.array_equal(ctx.extend_batch(wire), want)
    n = int(np.frombuffer(wire[8:12].tobytes(), "<i4")[0])
    rng = np.random.default_rng(4712)
    # (a) the records shuffled, the streams where they were; idx (the last word of a record) travels with its record
    perm = rng.permutation(n)
    shuffled = wire.copy()
    table = wire[32: 32 + 32 * n].reshape(n, 32)
    shuffled[32: 32 + 32 * n] = table[perm].reshape(-1)
    want_s, _ = orc.wire_extend(shuffled)
    assert np.array_equal(want_s.reshape(n, 10), want.reshape(n, 10)[perm])
    assert np.array_equal(ctx.extend_batch(shuffled), want_s)
    # (b) only every 64th pair of records swapped: most waves still stage, the others must notice
    part = wire.copy()
    t2 = table.copy()
    for i in range(10, n - 70, 64):
        t2[[i, i + 37]] = t2[[i + 37, i]]
    part[32: 32 + 32 * n] = t2.reshape(-1)
    want_p, _ = orc.wire_extend(part)
    assert np.array_equal(ctx.extend_batch(part), want_p)
    # (c) long tasks (flanks of 250 bases with their 500-base targets) sprinkled in: runs of 64 tasks that outgrow the buffer
    long_soa = synth.ext_tasks(4000, read_len=600, sub_rate=0.01, indel_rate=0.001, seed=4713)
    wl = bpsw_hip.wire_pack(long_soa)
    want_l, _ = orc.wire_extend(wl)
    assert np.array_equal(ctx.extend_batch(wl), want_l)


_INJECT = r"""
import os, sys
sys.path.insert(0, {pkg!r}); sys.path.insert(0, {orc!r})
import numpy as np
import bpsw_hip
from bpsw_hip import synth
import pyoracle as po
orc = po.Oracle()
ctx = bpsw_hip.Context(0)
for n, seed in ((3000, 1), (9000, 2)):
    soa = synth.ext_tasks(n, read_len=100, seed=seed)      # 100-base reads: no flank reaches 128 bases, nothing is EXPECTED to defer
    assert int(max(soa.left_qlen.max(), soa.right_qlen.max())) < 128
    wire = bpsw_hip.wire_pack(soa)
    want, _ = orc.wire_extend(wire)
    before = ctx.stats().ext_full_relaunches
    got = ctx.extend_batch(wire)
    assert np.array_equal(got, want), int((got != want).any(axis=1).sum())
    print("LATE", ctx.stats().ext_full_relaunches - before)
ctx.close()
"""


@pytest.mark.parametrize("lazy", ["1", "0"])
def test_an_unexpected_band_overflow_is_deferred_not_trapped(lazy):
    """Round 4's short kernel had no list to defer to when no flank of the batch reached 128 bases, and trapped -- taking the executor's
    process down (never abort the JVM) -- if a band outgrew the window all the same.  Now every deferring launch has a list:
    BPSW_EXT_INJECT_DEFER=7 sends every seventh task of such a batch down that path; the full kernel computes it (late launch, or the
    unconditional one with BPSW_EXT_LAZY_FULL=0), results bit-exact."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BPSW_EXT_INJECT_DEFER="7", BPSW_EXT_LAZY_FULL=lazy)
    src = _INJECT.format(pkg=os.path.join(root, "cloud-scale-bwamem_amd"), orc=os.path.join(root, "oracle"))
    r = subprocess.run([sys.executable, "-c", src], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    late = [int(ln.split()[1]) for ln in r.stdout.splitlines() if ln.startswith("LATE")]
    assert len(late) == 2
    if lazy == "1":
        assert all(v == 1 for v in late), late     # the list was not empty: the full kernel was launched behind the short one
