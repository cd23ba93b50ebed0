#!/usr/bin/env python3
"""Soak test of the rescue SW kernels against the oracle: mates of 1..256 bases (every columns-per-lane variant of the packed
kernel), ragged windows, low-complexity and repeated sequence, N in mates and windows, partial copies and decoys, several
scorings (default, other gap costs, general matrices, one the packed kernel must refuse) and flag sets (two passes, forward only,
early stop).  Usage on a GPU box: python tools/soak_sw.py [rounds] [jobs_per_round]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402
from test_swalign_gpu import _jobs_from  # noqa: E402



def seq(rng, n):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        s = rng.integers(0, 4, n)
    elif kind == 1:
        s = np.full(n, rng.integers(0, 4))
    elif kind == 2:
        s = np.tile(rng.integers(0, 4, int(rng.integers(1, 7))), n + 7)[:n]
    elif kind == 3:
        s = rng.integers(0, 2, n)
    else:
        s = np.where(rng.random(n) < 0.8, rng.integers(0, 4), rng.integers(0, 4, n))
    return s.astype(np.int64)


def mutate(rng, s, sub, indel):
    out = []
    for b in s:
        u = rng.random()
        if u < indel / 2:
            continue
        if u < indel:
            out.extend(rng.integers(0, 4, int(rng.integers(1, 6))).tolist())
        out.append(int((b + 1 + rng.integers(0, 3)) & 3) if rng.random() < sub else int(b))
    return np.array(out, np.int64)


def scorings():
    yield "default", (1, 4, 6, 1, 6, 1), None
    yield "gaps", (1, 4, 3, 2, 7, 1), None
    yield "a2b3", (2, 3, 5, 2, 4, 1), None
    yield "free-open", (1, 4, 0, 1, 0, 2), None
    yield "general", (2, 4, 6, 1, 6, 1), [1, -2, -3, -4, -1, -2, 2, -4, -3, 0, -3, -4, 1, -2, -1, -4, -3, -2, 2, -2, -1, 0, -1, -2, -1]
    yield "a5b2 (32-bit kernels)", (5, 2, 6, 1, 6, 1), None


def run(rounds=20, per=1500, time_limit=None, log=print):
    """`rounds` rounds of `per` jobs each (six scorings x six flag sets per round); stops early after `time_limit` seconds.
    Returns (job runs compared, differences).  tests/test_soak_gpu.py runs a one-minute slice of this."""
    import time
    ctx, orc = bpsw_hip.Context(0), po.Oracle()
    t_start = time.time()
    total = bad_total = 0
    for rd in range(rounds):
        if time_limit is not None and time.time() - t_start > time_limit:
            break
        rng = np.random.default_rng(7000 + rd)
        top = [57, 114, 171, 228, 256, 150, 100, 250][rd % 8]
        pairs = []
        for _ in range(per):
            n = int(rng.integers(1, top + 1)) if rng.random() < 0.3 else int(rng.integers(max(1, top - 20), top + 1))
            q = seq(rng, n)
            u = rng.random()
            if u < 0.6:       # a (mutated) copy somewhere in the window
                core = mutate(rng, q, float(rng.choice([0.0, 0.02, 0.1, 0.25])), float(rng.choice([0.0, 0.01, 0.05])))
                t = np.concatenate([seq(rng, int(rng.integers(0, 300))), core, seq(rng, int(rng.integers(0, 500)))])
            elif u < 0.75:    # two copies: second-best logic
                a = mutate(rng, q, 0.03, 0.0)
                b = mutate(rng, q[int(rng.integers(0, max(1, n // 2))):], 0.05, 0.01)
                t = np.concatenate([seq(rng, int(rng.integers(0, 100))), a, seq(rng, int(rng.integers(0, 200))), b, seq(rng, int(rng.integers(0, 100)))])
                if rng.random() < 0.5:
                    t = t[::-1].copy()
            elif u < 0.9:     # unrelated
                t = seq(rng, int(rng.integers(0, 900)))
            else:             # short / empty window
                t = q[: int(rng.integers(0, n + 1))]
            if rng.random() < 0.15 and len(t):
                t = t.copy(); t[rng.integers(0, len(t), max(1, len(t) // 60))] = 4
            if rng.random() < 0.1:
                q = q.copy(); q[rng.integers(0, n, max(1, n // 40))] = 4
            pairs.append((q.tolist(), t.tolist(), int(rng.integers(0, 2))))
        if rng.random() < 0.5:
            pairs = pairs[:-1]            # odd and even job counts
        jobs = _jobs_from(pairs)
        for name, (a, b, od, ed, oi, ei), mat in scorings():
            oo, op = orc.default_opt(), bpsw_hip.default_opt()
            for o in (oo, op):
                o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins = a, b, od, ed, oi, ei
                m = po.default_mat(a, b) if mat is None else mat
                for k in range(25):
                    o.mat[k] = int(m[k])
            for xtra in (po.KSW_XSUBO | po.KSW_XSTART | po.KSW_XBYTE | 19 * a, po.KSW_XSTART | po.KSW_XSUBO | 1, po.KSW_XSUBO | 30 * a,
                         po.KSW_XSTOP | 40 * a, po.KSW_XSTART | po.KSW_XSTOP | po.KSW_XSUBO | 25 * a, 0):
                got = ctx.swalign2_batch(op, xtra, **jobs)
                want, _ = orc.sw_align2_jobs(oo, xtra, **jobs)
                bad = np.nonzero((got != want).any(axis=1))[0]
                total += len(want)
                if bad.size:
                    bad_total += bad.size
                    log(f"round {rd} top {top} scoring {name} xtra {xtra:#x}: {bad.size} differ; first job {bad[0]} got {got[bad[0]]} want {want[bad[0]]}")
        log(f"round {rd} (mates <= {top}): {total} job runs so far, {bad_total} differences")
    ctx.close()
    return total, bad_total


if __name__ == "__main__":
    total, bad_total = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1500)
    print(f"TOTAL {total} job runs, {bad_total} differences")
    sys.exit(1 if bad_total else 0)
