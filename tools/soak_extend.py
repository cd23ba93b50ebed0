#!/usr/bin/env python3
"""Soak test of the extension kernel against the oracle on adversarial flanks (near-exact flanks around the validity boundaries
of the closed form and the single-gap certificate, repeats, restarts, long deletions), many seeds and gap-cost sets.
Usage on a GPU box: python tools/soak_extend.py [n_rounds] [tasks_per_round]"""
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

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
per = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
ctx, orc = bpsw_hip.Context(0), po.Oracle()
total = bad_total = 0
for rd in range(rounds):
    rng = np.random.default_rng(1000 + rd)
    tasks = []
    for t in range(per):
        sides = []
        for _ in range(2):
            n = int(rng.integers(1, 132))
            kind = int(rng.integers(0, 6))
            if kind == 0:
                q = rng.integers(0, 4, n)
            elif kind == 1:
                q = np.full(n, rng.integers(0, 4))
            elif kind == 2:
                q = np.tile(rng.integers(0, 4, int(rng.integers(1, 6))), n)[:n]
            elif kind == 3:
                q = rng.integers(0, 2, n)
            else:
                q = np.where(rng.random(n) < 0.85, rng.integers(0, 4), rng.integers(0, 4, n))
            q = q.astype(np.int64)
            r = q.copy()
            pos = []
            for _ in range(int(rng.integers(0, 5))):
                if pos and rng.random() < 0.6:
                    p = min(n - 1, pos[-1] + int(rng.integers(1, 5)))
                else:
                    p = 0 if rng.random() < 0.4 else int(rng.integers(0, n))
                pos.append(p)
                u = rng.random()
                if u < 0.15:
                    r[p] = 4
                elif u < 0.25:
                    q[p] = 4
                else:
                    r[p] = (r[p] + 1 + rng.integers(0, 3)) & 3
            u = rng.random()
            if u < 0.1:      # an indel instead
                p = int(rng.integers(0, n))
                r = np.concatenate([r[:p], rng.integers(0, 4, int(rng.integers(1, 12))), r[p:]])
            elif u < 0.2:
                p = int(rng.integers(0, n)); k = int(rng.integers(1, 8))
                r = np.concatenate([r[:p], r[p + k:]])
            extra = int(rng.integers(0, 100)) if rng.random() < 0.9 else 0
            v = rng.random()
            tail = rng.integers(0, 5, extra) if v < 0.5 else (np.tile(q, 3)[:extra] if v < 0.8 else np.concatenate([rng.integers(0, 4, extra // 2), q])[:extra + n])
            sides.append((q.tolist(), np.concatenate([r, tail]).astype(np.int64).tolist()))
        h0 = int(rng.integers(1, 16)) if rng.random() < 0.15 else int(rng.integers(19, 150))
        if rng.random() < 0.1:
            sides[0] = ([], [])
        tasks.append((sides[0][0], sides[0][1], sides[1][0], sides[1][1], h0, len(sides[0][0])))
    soa = _manual_tasks(tasks)
    for (od, ed, oi, ei), w in (((6, 1, 6, 1), 100), ((0, 1, 0, 1), 100), ((3, 2, 7, 1), 30), ((4, 2, 2, 3), 100), ((1, 1, 1, 1), 5),
                                ((10, 1, 9, 2), 100), ((2, 3, 8, 1), 3), ((6, 1, 6, 1), 2)):
        soa.o_del, soa.e_del, soa.o_ins, soa.e_ins, soa.w = od, ed, oi, ei, w
        wire = bpsw_hip.wire_pack(soa)
        for zmode, zdrop in ((0, 100), (1, 100), (1, 7), (0, 0)):
            ctx.set_ext_scoring(po.default_mat(), zdrop, zmode)
            got = ctx.extend_batch(wire).reshape(-1, 10)
            want, _ = orc.wire_extend(wire, po.default_mat(), zdrop, zmode)
            want = want.reshape(-1, 10)
            bad = np.nonzero((got != want).any(axis=1))[0]
            total += soa.n
            if bad.size:
                bad_total += bad.size
                print("ROUND", rd, (od, ed, oi, ei), w, zmode, zdrop, "bad", bad.size, "first", int(bad[0]), got[bad[0]], want[bad[0]], flush=True)
    print("round", rd, "tasks so far", total, "bad", bad_total, flush=True)
print("SOAK", {"task_runs": total, "bad": bad_total})
