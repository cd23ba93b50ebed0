#!/usr/bin/env python3
"""Soak test aimed at the closed form's two-gap-open extension (bpsw_extend_core.h, flank_closed_form): flanks whose main
diagonal has a deficit of 14 or 15 under the default scoring (three substitutions; two plus two N; ...), in low-complexity and
periodic sequence, with the substituted bases and the target tail chosen so that the shifted diagonals of the two exclusion
tests match more often than by chance.  Usage on a GPU box: python tools/soak_cert2.py [rounds] [tasks_per_round]"""
import os
import sys

os.environ.setdefault("BPSW_EXT_SIFT_MIN", "0")   # the sift kernel on every batch, whatever its size

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402
from test_extend_gpu import _manual_tasks  # noqa: E402



# matrices of the family the forms accept (match 1, everything else <= -1) and one they must refuse (a mismatch of 0 ... -1 only for N)
def _mat(b, nscore):
    m = po.default_mat(1, b).copy()
    m[m == -1] = nscore if b != 1 else -1
    for k in range(5):
        m[4 * 5 + k] = nscore; m[k * 5 + 4] = nscore
    return m


MATS = [po.default_mat(), po.default_mat(), _mat(1, -1), _mat(2, -3), _mat(6, -1)] if os.environ.get("SOAK_MATS") == "1" else [po.default_mat()]
SHORT = os.environ.get("SOAK_SHORT") == "1"   # short flanks over tiny alphabets: shifted diagonals match all the time


def gap_side(rng):
    """a one-base (sometimes longer) gap at or near the start of the flank and nothing else, or nearly nothing: flank_start_gap_form"""
    n = int(rng.integers(4, 40)) if SHORT else int(rng.integers(5, 132))
    kind = int(rng.integers(0, 4))
    q = (rng.integers(0, 4, n) if kind == 0 else np.tile(rng.integers(0, 4, int(rng.integers(1, 4))), n)[:n] if kind == 1
         else rng.integers(0, 2, n) if kind == 2 else np.where(rng.random(n) < 0.6, rng.integers(0, 4), rng.integers(0, 4, n))).astype(np.int64)
    L = 1 if rng.random() < 0.8 else int(rng.integers(2, 4))
    p = 0 if rng.random() < 0.8 else int(rng.integers(1, 6))
    if rng.random() < 0.5:      # the read has L bases more
        r = np.concatenate([q[:p], q[p + L:]])
    else:                       # the reference has L bases more
        r = np.concatenate([q[:p], rng.integers(0, 4, L), q[p:]])
    r = r.copy()
    u = rng.random()
    if u < 0.25 and len(r) > 2:  # one more difference somewhere: the form must refuse
        x = int(rng.integers(0, len(r)))
        r[x] = 4 if rng.random() < 0.2 else (r[x] + 1 + rng.integers(0, 3)) & 3
    elif u < 0.3:
        q = q.copy(); q[int(rng.integers(0, n))] = 4
    extra = int(rng.integers(0, 30)) if rng.random() < 0.85 else 0
    v = rng.random()
    tail = rng.integers(0, 5, extra) if v < 0.5 else (np.tile(q[-3:], extra)[:extra] if v < 0.8 else np.tile(q, 2)[:extra])
    return q.tolist(), np.concatenate([r, tail]).astype(np.int64).tolist()


def side(rng):
    if rng.random() < 0.35:
        return gap_side(rng)
    n = int(rng.integers(4, 31)) if SHORT else int(rng.integers(8, 129))
    kind = int(rng.integers(1, 4)) if SHORT else int(rng.integers(0, 5))
    if kind == 0:
        q = rng.integers(0, 4, n)
    elif kind == 1:
        q = np.full(n, rng.integers(0, 4))
        q[rng.integers(0, n, max(1, n // 12))] = rng.integers(0, 4)
    elif kind == 2:
        q = np.tile(rng.integers(0, 4, int(rng.integers(1, 5))), n)[:n]
    elif kind == 3:
        q = rng.integers(0, 2, n)
    else:
        q = np.where(rng.random(n) < 0.7, np.tile(rng.integers(0, 4, 2), n)[:n], rng.integers(0, 4, n))
    q = q.astype(np.int64)
    r = q.copy()
    # deficit positions: three substitutions (15), or two + two N (14), or one + five N (15), clustered or spread
    u = rng.random()
    n_sub, n_n = (3, 0) if u < 0.6 else ((2, 2) if u < 0.8 else ((1, 5) if u < 0.9 else (int(rng.integers(0, 5)), int(rng.integers(0, 4)))))
    start = int(rng.integers(0, n))
    span = int(rng.integers(2, 40)) if rng.random() < 0.6 else n
    pos = sorted(set(int(min(n - 1, start + rng.integers(0, span))) for _ in range(n_sub + n_n)))
    rng.shuffle(pos)
    for k, p in enumerate(pos):
        if k < n_sub:
            v = rng.random()
            cand = None
            if v < 0.5 and p + 2 < n:      # make the substituted target base equal a query base one to three columns on
                cand = int(q[p + int(rng.integers(1, min(4, n - p)))])
            if cand is None or cand == q[p] or cand > 3:
                cand = int((q[p] + 1 + rng.integers(0, 3)) & 3)
            r[p] = cand
        else:
            if rng.random() < 0.5:
                r[p] = 4
            else:
                q[p] = 4
    extra = int(rng.integers(0, 40)) if rng.random() < 0.9 else 0
    v = rng.random()
    if v < 0.4:       # the query's end again, shifted down by 2 or 3 rows: the two-deletion test
        pad = rng.integers(0, 4, int(rng.integers(1, 4)))
        tail = np.concatenate([pad, np.tile(q[-1:], extra + 3)])[: extra + 3]
    elif v < 0.7:
        tail = np.tile(q, 2)[: extra]
    else:
        tail = rng.integers(0, 5, extra)
    return q.tolist(), np.concatenate([r, tail]).astype(np.int64).tolist()


def run(rounds=20, per=4000, time_limit=None, log=print):
    """`rounds` rounds of `per` two-sided tasks each (six gap-cost / band settings x four z-drop settings per round); stops early
    after `time_limit` seconds.  Returns (task runs compared, differences).  tests/test_soak_gpu.py runs a one-minute slice."""
    import time
    ctx, orc = bpsw_hip.Context(0), po.Oracle()
    # the same forms evaluated by ext_kernel alone (no sift kernel, csrc/bpsw_extend_sift.hip): the verdicts per side must agree
    ctx_wave = bpsw_hip.Context(0)
    ctx_wave.set_ext_shortcuts(31)
    t_start = time.time()
    total = bad_total = 0
    for rd in range(rounds):
        if time_limit is not None and time.time() - t_start > time_limit:
            break
        rng = np.random.default_rng(31000 + rd)
        tasks = []
        for t in range(per):
            l, r = side(rng), side(rng)
            h0 = int(rng.integers(16, 60)) if rng.random() < 0.5 else int(rng.integers(16, 150))
            if SHORT and rng.random() < 0.3:
                h0 = int(rng.integers(1, 24))
            if rng.random() < 0.1:
                l = ([], [])
            tasks.append((l[0], l[1], r[0], r[1], h0, len(l[0])))
        soa = _manual_tasks(tasks)
        for (od, ed, oi, ei), w in (((6, 1, 6, 1), 100), ((6, 1, 6, 1), 3), ((6, 1, 6, 1), 2), ((1, 1, 1, 1), 100), ((3, 1, 3, 1), 100), ((2, 1, 2, 1), 7)):
            soa.o_del, soa.e_del, soa.o_ins, soa.e_ins, soa.w = od, ed, oi, ei, w
            wire = bpsw_hip.wire_pack(soa)
            for zmode, zdrop in ((0, 100), (1, 100), (1, 16), (0, 0)):
                mat = MATS[(rd + zmode + zdrop) % len(MATS)]
                ctx.set_ext_scoring(mat, zdrop, zmode)
                got, how = ctx.extend_batch_classify(wire)
                got = got.reshape(-1, 10)
                ctx_wave.set_ext_scoring(mat, zdrop, zmode)
                got_w, how_w = ctx_wave.extend_batch_classify(wire)
                want, _ = orc.wire_extend(wire, mat, zdrop, zmode)
                want = want.reshape(-1, 10)
                bad = np.nonzero((got != want).any(axis=1) | (got_w.reshape(-1, 10) != want).any(axis=1) | (how != how_w).any(axis=1))[0]
                total += soa.n
                if bad.size:
                    bad_total += bad.size
                    log(f"round {rd} gaps {(od, ed, oi, ei)} w {w} z {zmode}/{zdrop}: {bad.size} differ; task {bad[0]} got {got[bad[0]]} want {want[bad[0]]} "
                        f"verdicts with / without the sift kernel {how[bad[0]]} / {how_w[bad[0]]}")
        log(f"round {rd} tasks so far {total} bad {bad_total}")
    ctx.close()
    ctx_wave.close()
    return total, bad_total


if __name__ == "__main__":
    total, bad_total = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 4000)
    print("SOAK", {"task_runs": total, "bad": bad_total})
    sys.exit(1 if bad_total else 0)
