#!/usr/bin/env python3
"""Which flanks of the bench batch still run the DP (the exact shortcuts of DESIGN.md 4.1 refused them), by what is in them: the
flank's defects against its target are found with a small banded alignment (unit costs), and the sides are bucketed by
(substitutions, indel events) -- weighted by query length, which is what the DP costs.  Usage on a GPU box:
python tools/dp_side_census.py [config]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import bpsw_hip  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W = bench.WORKLOADS[cfg]
soa = bench.make_ext_soa(W, cfg, 0, 0)
wire = bpsw_hip.wire_pack(soa)
ctx = bpsw_hip.Context(0)
_, how = ctx.extend_batch_classify(wire)


def defects(q, t):
    """(subs, indel events) of the cheapest unit-cost alignment of q against a prefix of t, band 8"""
    n, B = len(q), 8
    INF = 1 << 20
    prev = {d: (abs(d), 0, 1 if d else 0) for d in range(-B, B + 1)}   # cost, subs, indel events (row 0: gaps at the start)
    prev = {d: (INF, 0, 0) for d in range(-B, B + 1)}
    prev[0] = (0, 0, 0)
    state = {0: (0, 0, 0, 0)}  # diagonal offset -> (cost, subs, events, last_was_gap)
    for i in range(n):
        new = {}
        for d, (c, s, e, g) in state.items():
            j = i + d
            if 0 <= j < len(t):   # diagonal step
                m = 0 if q[i] == t[j] else 1
                cand = (c + m, s + m, e, 0)
                if d not in new or cand < new[d]:
                    new[d] = cand
            if d - 1 >= -B:       # q[i] inserted (target does not advance)
                cand = (c + 1, s, e + (0 if g == 1 else 1), 1)
                if d - 1 not in new or cand < new[d - 1]:
                    new[d - 1] = cand
        # deletions (target advances, query does not): relax within the row
        for d in sorted(new):
            c, s, e, g = new[d]
            if d + 1 <= B:
                cand = (c + 1, s, e + (0 if g == 2 else 1), 2)
                if d + 1 not in new or cand < new[d + 1]:
                    new[d + 1] = cand
        state = new
        if not state:
            return (99, 99)
    c, s, e, g = min(state.values())
    return (s, e)


buckets = {}
tot = {"all": 0, "dp": 0}
pool = soa.pool
rng = np.random.default_rng(1)
sel = rng.choice(soa.n, min(soa.n, 6000), replace=False)
for t in sel:
    for side, (ql, rl, qo, ro) in enumerate(((soa.left_qlen, soa.left_rlen, soa.left_q_off, soa.left_r_off),
                                             (soa.right_qlen, soa.right_rlen, soa.right_q_off, soa.right_r_off))):
        n = int(ql[t])
        if n == 0:
            continue
        tot["all"] += n
        if how[t, side] != 2:
            continue
        tot["dp"] += n
        q = pool[int(qo[t]): int(qo[t]) + n].tolist()
        tg = pool[int(ro[t]): int(ro[t]) + int(rl[t])].tolist()
        s, e = defects(q, tg)
        key = (min(s, 6), min(e, 3))
        b = buckets.setdefault(key, [0, 0])
        b[0] += 1
        b[1] += n
print("config", cfg, "tasks sampled", len(sel), "query bases", tot["all"], "of which on DP-run sides", tot["dp"], f"({100 * tot['dp'] / max(tot['all'], 1):.1f} %)")
print("(subs, indel events) -> sides, share of the DP-run query bases")
for k in sorted(buckets, key=lambda k: -buckets[k][1]):
    print(k, buckets[k][0], f"{100 * buckets[k][1] / max(tot['dp'], 1):.1f} %")
