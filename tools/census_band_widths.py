"""Census of the extension's DP rows on the bench workloads, from the oracle's own counters (oracle/bpsw_oracle.c:22,101): for every
SWExtend row the band width end - beg + 1, in buckets of eight columns -- what the layout of the row sweep should be sized for (round-5
review, item 3).  Also: rows per task, rows per DP side, tasks with a DP side.  CPU only.

    python tools/census_band_widths.py [config ...]      (default: 3 5 = BASELINE.json configs[2] and configs[4])"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cloud-scale-bwamem_amd"), os.path.join(ROOT, "oracle"), ROOT):
    sys.path.insert(0, p)
import bench  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402


def census(cfg):
    W = bench.WORKLOADS[cfg]
    orc = po.Oracle()
    lib = orc.lib
    lib.orc_diag_ext_widths.argtypes = [C.POINTER(C.c_int64), C.c_int]
    lib.orc_diag_ext_rows.restype = C.c_int64
    lib.orc_diag_ext_calls.restype = C.c_int64
    hist = (C.c_int64 * 33)()
    lib.orc_diag_ext_widths(hist, 1); lib.orc_diag_ext_rows(1); lib.orc_diag_ext_calls(1)
    soa = bench.make_ext_soa(W, cfg, 0, 0)
    n = min(soa.n, 8192)
    soa = soa.subset(np.arange(n))
    wire = bpsw_hip.wire_pack(soa)
    orc.wire_extend(wire)
    lib.orc_diag_ext_widths(hist, 1)
    rows, calls = lib.orc_diag_ext_rows(1), lib.orc_diag_ext_calls(1)
    h = np.array(list(hist), np.int64)
    tot = int(h.sum())
    print(f"config {cfg} ({W['read_len']} bp, {W['sub']:.0%} sub / {W['indel']:.1%} indel): {n} tasks, {calls} SWExtend calls (the oracle computes every side by DP: "
          f"no shortcut), {rows} rows = {rows / n:.1f} per task, {rows / max(calls, 1):.1f} per call")
    cum = 0
    for k in range(33):
        if h[k] == 0:
            continue
        cum += int(h[k])
        lab = f"{8 * k:3d}-{8 * k + 7:3d}" if k < 32 else "256+   "
        print(f"  width {lab}: {int(h[k]):9d} rows  {h[k] / tot:7.2%}   cumulative {cum / tot:7.2%}")
    for cut in (4, 8):
        print(f"  rows with a band of at most {8 * cut - 1} columns: {h[:cut].sum() / tot:.1%}")


if __name__ == "__main__":
    for c in ([int(a) for a in sys.argv[1:]] or [3, 5]):
        census(c)
