#!/usr/bin/env python3
"""Soak test of the extension kernel on LONG flanks (64..255 query bases: the sliding-window sweep, its hand-over from the slot
sweep when the first rows are wider than the window, its window moves, its overflow fallback) against the oracle: read-like
flanks at 0..20 % substitutions and 0..5 % indels, long insertions / deletions (the band drifts), repeats, seeds with a high h0
(wide bands), unrelated sequence after a good stretch (z-drop), band widths 3..127 and their doubles in the retry (200, 254), several
gap-cost sets, both z-drop parses.  Usage on a GPU box: python tools/soak_long.py [rounds] [tasks_per_round]"""
import os
os.environ.setdefault("BPSW_EXT_SIFT_MIN", "0")   # the sift kernel on every batch, whatever its size
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402
from test_extend_gpu import _manual_tasks  # noqa: E402


def flank(rng):
    n = int(rng.integers(64, 256)) if rng.random() < 0.85 else int(rng.integers(120, 136))
    kind = int(rng.integers(0, 5))
    if kind <= 1:
        q = rng.integers(0, 4, n)
    elif kind == 2:
        q = np.tile(rng.integers(0, 4, int(rng.integers(2, 9))), n)[:n]
    elif kind == 3:
        q = rng.integers(0, 2, n)
    else:
        q = np.where(rng.random(n) < 0.8, rng.integers(0, 4), rng.integers(0, 4, n))
    q = q.astype(np.int64)
    sub = float(rng.choice([0.0, 0.01, 0.05, 0.08, 0.2]))
    ind = float(rng.choice([0.0, 0.005, 0.02, 0.05]))
    t = []
    for b in q:
        u = rng.random()
        if u < ind / 2:
            continue
        if u < ind:
            t.extend(rng.integers(0, 4, int(rng.integers(1, 4))).tolist())
        t.append(int((b + 1 + rng.integers(0, 3)) & 3) if rng.random() < sub else int(b))
    t = np.array(t, np.int64)
    u = rng.random()
    if u < 0.15 and len(t) > 40:      # one long deletion / insertion: the band runs off the diagonal
        p = int(rng.integers(10, len(t) - 10)); k = int(rng.integers(8, 60))
        t = np.concatenate([t[:p], t[p + k:]]) if rng.random() < 0.5 else np.concatenate([t[:p], rng.integers(0, 4, k), t[p:]])
    elif u < 0.25:                    # good stretch, then unrelated sequence (z-drop / m == 0)
        p = int(rng.integers(20, max(21, len(t))))
        t = np.concatenate([t[:p], rng.integers(0, 4, len(t))])
    if rng.random() < 0.1:
        q = q.copy(); q[rng.integers(0, n, 2)] = 4
    if rng.random() < 0.1 and len(t):
        t = t.copy(); t[rng.integers(0, len(t), 2)] = 4
    extra = int(rng.integers(0, 220))
    v = rng.random()
    tail = rng.integers(0, 4, extra) if v < 0.6 else np.tile(q, 2)[:extra]
    return q.tolist(), np.concatenate([t, tail]).astype(np.int64).tolist()


def run(rounds=10, per=1500, time_limit=None, log=print):
    import time
    ctx, orc = bpsw_hip.Context(0), po.Oracle()
    t_start = time.time()
    total = bad_total = 0
    for rd in range(rounds):
        if time_limit is not None and time.time() - t_start > time_limit:
            break
        rng = np.random.default_rng(52000 + rd)
        tasks = []
        for _ in range(per):
            l, r = flank(rng), flank(rng)
            u = rng.random()
            h0 = int(rng.integers(19, 60)) if u < 0.5 else (int(rng.integers(60, 260)) if u < 0.9 else int(rng.integers(1, 19)))
            if rng.random() < 0.15:
                l = ([], [])
            tasks.append((l[0], l[1], r[0], r[1], h0, len(l[0])))
        soa = _manual_tasks(tasks)
        # w travels as a signed byte (MemChainToAlignBatched.scala:78-84); the retry doubles it inside extension(): 100 -> 200 etc.
        for (od, ed, oi, ei), w in (((6, 1, 6, 1), 100), ((6, 1, 6, 1), 127), ((6, 1, 6, 1), 30), ((6, 1, 6, 1), 63), ((6, 1, 6, 1), 64),
                                    ((3, 2, 7, 1), 100), ((0, 1, 0, 1), 60), ((1, 1, 1, 1), 3), ((10, 1, 9, 2), 100)):
            soa.o_del, soa.e_del, soa.o_ins, soa.e_ins, soa.w = od, ed, oi, ei, w
            wire = bpsw_hip.wire_pack(soa)
            for zmode, zdrop in ((0, 100), (1, 100), (1, 20)):
                ctx.set_ext_scoring(po.default_mat(), zdrop, zmode)
                got = ctx.extend_batch(wire).reshape(-1, 10)
                want, _ = orc.wire_extend(wire, po.default_mat(), zdrop, zmode)
                want = want.reshape(-1, 10)
                bad = np.nonzero((got != want).any(axis=1))[0]
                total += soa.n
                if bad.size:
                    bad_total += bad.size
                    log(f"round {rd} gaps {(od, ed, oi, ei)} w {w} z {zmode}/{zdrop}: {bad.size} differ; task {bad[0]} (lq {soa.left_qlen[bad[0]]} rq "
                        f"{soa.right_qlen[bad[0]]} h0 {soa.h0[bad[0]]}) got {got[bad[0]]} want {want[bad[0]]}")
        log(f"round {rd} task runs so far {total} bad {bad_total}")
    ctx.close()
    return total, bad_total


if __name__ == "__main__":
    total, bad_total = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1500)
    print("SOAK", {"task_runs": total, "bad": bad_total})
    sys.exit(1 if bad_total else 0)
