#!/usr/bin/env python3
"""Cost of one DP row of the extension kernel by flank length: a batch of right-side-only tasks whose query is exactly L bases
(target = the query with substitutions / indels plus a random tail), run with every exact shortcut off (bpsw_set_ext_shortcuts 0)
so that every row is swept; prints the rows and cells the oracle counts for the batch.  Under
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace ... -- python3 tools/ext_row_cost.py L [sub] [indel] [mask]
the counters of ext_kernel divided by `rows` are the instructions per row of the sweep that serves that length.
Usage: python tools/ext_row_cost.py L [sub_rate] [indel_rate] [shortcut_mask] [n_tasks] [reps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sub = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
indel = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
mask = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n = int(sys.argv[5]) if len(sys.argv) > 5 else 8192
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5

rng = np.random.default_rng(100 + L)
pool, q_off, t_off, t_len = [], [], [], []
at = 0
for _ in range(n):
    q = rng.integers(0, 4, L).astype(np.uint8)
    t = []
    for b in q:
        u = rng.random()
        if u < indel / 2:
            continue
        if u < indel:
            t.append(int(rng.integers(0, 4)))
        t.append(int((b + 1 + rng.integers(0, 3)) & 3) if rng.random() < sub else int(b))
    t = np.array(t + rng.integers(0, 4, 100).tolist(), np.uint8)
    q_off.append(at); pool.append(q); at += L
    t_off.append(at); pool.append(t); at += len(t); t_len.append(len(t))
z32, z64 = np.zeros(n, np.int32), np.zeros(n, np.int64)
h0 = np.full(n, int(os.environ.get("H0", "30")), np.int32)
soa = bpsw_hip.ExtTaskSoA(pool=np.concatenate(pool + [np.zeros(16, np.uint8)]), left_qlen=z32, left_rlen=z32, left_q_off=z64, left_r_off=z64,
                          right_qlen=np.full(n, L, np.int32), right_rlen=np.array(t_len, np.int32), right_q_off=np.array(q_off, np.int64),
                          right_r_off=np.array(t_off, np.int64), reg_score=h0, h0=h0, q_beg=z32, idx=np.arange(n, dtype=np.int32))
wire = bpsw_hip.wire_pack(soa)
orc = po.Oracle()
orc.lib.orc_diag_ext_rows.restype = C.c_int64
orc.lib.orc_diag_ext_rows(1)
h = (C.c_int64 * 33)()
orc.lib.orc_diag_ext_widths(h, 1)
want, cells = orc.wire_extend(wire)
rows = orc.lib.orc_diag_ext_rows(0)
orc.lib.orc_diag_ext_widths(h, 0)
hist = np.array(list(h))
ctx = bpsw_hip.Context(0)
ctx.set_ext_shortcuts(mask)
got = ctx.extend_batch(wire)
assert np.array_equal(got, np.asarray(want).reshape(-1)), "kernel differs from the oracle"
s0 = ctx.stats()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.extend_batch(wire)
dt = (time.perf_counter() - t0) / reps
s1 = ctx.stats()
k_ms = (s1.ext_kernel_ms - s0.ext_kernel_ms) / reps
print({"L": L, "sub": sub, "indel": indel, "mask": mask, "tasks": n, "rows": int(rows), "cells": int(cells), "cells_per_row": round(cells / rows, 1),
       "rows_le_64_wide": round(float(hist[:8].sum() / hist.sum()), 3), "kernel_ms": round(k_ms, 4), "ns_per_row": round(1e6 * k_ms / rows, 3),
       "launches": reps + 1})
