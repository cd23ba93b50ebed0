#!/usr/bin/env python3
"""Which DP rows a bench-shaped extension batch sweeps, by what the row loops of csrc/bpsw_extend_rows.h care about: CPU only.
The host build of the sift arithmetic (tests/sift_host) says which sides the exact shortcuts resolve; every other side is swept
row by row here (the recurrences of SWUtil.scala:61-230 in the kernels' parallel form, checked against the oracle's result) and
every row is binned by: columns per lane the band needs (1: <= 63 columns, 2), phase of h1 (live / dead), band end at the query
end, zero cell in the band (the trimming's slow path), row improved the maximum, row at or past the query end (tail).
Usage: python tools/row_mix_census.py [config] [n_tasks]"""
import collections
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle", "tests", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import bench  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_want = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
W = bench.WORKLOADS[cfg]
soa = bench.make_ext_soa(W, cfg, 0, 0)
wire = bpsw_hip.wire_pack(soa)
orc = po.Oracle()
mat = po.default_mat().reshape(5, 5).astype(np.int64)

# the host sift
here = os.path.join(ROOT, "tests", "sift_host")
so = os.path.join(here, "_build", "libsift_host.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
hdr = os.path.join(ROOT, "cloud-scale-bwamem_amd", "csrc")
subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + hdr, "-o", so, os.path.join(here, "sift_host.cpp")], check=True)
lib = C.CDLL(so)
n = soa.n
out = np.zeros(10 * n, np.int16); flag = np.zeros(n, np.uint8); kinds = np.zeros(2 * n, np.uint8)
w32 = np.ascontiguousarray(wire).view(np.uint32)
lib.sift_host_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_int] * 1 + [C.c_int] * 5 + [C.c_void_p] * 3
qmax = 127 if W["read_len"] <= 150 else 0   # the sift kernel is not launched for batches with longer flanks
rc = lib.sift_host_batch(w32.ctypes.data, w32.size, n, 100, 3, 1, 5, qmax, out.ctypes.data, flag.ctypes.data, kinds.ctypes.data)
assert rc == 0
SIFT_FORM = 2  # bpsw_extend_sift_core.h: SIFT_UNSEEN 0, SIFT_FAIL 1, SIFT_FORM 2


def sweep(q, t, h0, w, stats, end_bonus):
    global span_hist
    """one SWExtend call; returns (max, qle, tle, gtle, gscore, max_off) and bins its rows"""
    oD, eD, oI, eI = soa.o_del, soa.e_del, soa.o_ins, soa.e_ins
    qlen, tlen = len(q), len(t)
    amax = 1
    max_ins = max(1, int((qlen * amax + end_bonus - oI) / eI + 1.0)); w = min(w, max_ins)
    max_del = max(1, int((qlen * amax + end_bonus - oD) / eD + 1.0)); w = min(w, max_del)
    H = np.zeros(qlen + 2, np.int64); E = np.zeros(qlen + 2, np.int64)
    H[0] = h0
    if qlen >= 1:
        H[1] = h0 - (oI + eI) if h0 > oI + eI else 0
    j = 2
    while j <= qlen and H[j - 1] > eI:
        H[j] = H[j - 1] - eI; j += 1
    mx, max_i, max_j, max_ie, gscore, max_off = h0, -1, -1, -1, -1, 0
    beg, end = 0, qlen
    sc = mat[:, q] if qlen else np.zeros((5, 0), np.int64)
    for i in range(tlen):
        h1 = max(0, h0 - (oD + eD * (i + 1)))
        beg = max(beg, i - w); end = min(end, i + w + 1, qlen)
        # tail-row bound (bpsw_extend_core.h tail_row_bound): the kernels stop here
        if i >= qlen:
            U = max(h0 + qlen * amax - oD - (i - qlen + 1) * eD, qlen * amax)
            if U <= mx and U < gscore:
                break
        span = end - beg
        span_hist[min(span, 255) // 8] += 1
        key = ("cols2" if span > 63 else "cols1", "live" if h1 > 0 else "dead", "atend" if end == qlen else "inner", "tail" if i >= qlen else "body")
        if span > 0:
            js = np.arange(beg, end)
            a = np.maximum(H[js] + sc[t[i], js], E[js])
            g = a + js * eI - (oI + eI)
            pre = np.maximum.accumulate(g)
            F = np.concatenate(([0], np.maximum(0, pre[:-1] - (js[1:] - 1) * eI)))
            Hn = np.maximum(a, F)
            m = int(Hn.max()); mj = int(js[len(js) - 1 - int(np.argmax(Hn[::-1]))])
            E[js] = np.maximum(np.maximum(E[js] - eD, Hn - (oD + eD)), 0)
            H[beg] = h1; H[beg + 1:end + 1] = Hn
            E[end] = 0
            hlast = int(Hn[-1])
        else:
            m, mj, hlast = 0, -1, h1
            H[end] = h1; E[end] = 0
        if (end if span > 0 else beg) == qlen and gscore <= hlast:
            max_ie, gscore = i, hlast
        zero = span > 0 and bool((Hn == 0).any())
        imp = m > mx
        stats[key + ("zero" if zero else "nozero", "imp" if imp else "noimp")] += 1
        if m == 0:
            break
        if imp:
            mx, max_i, max_j = m, i, mj; max_off = max(max_off, abs(mj - i))
        else:
            k = (i - max_i) - (mj - max_j)
            if k > 0 and (mx - m) + k * eI > 100:   # Scala parse (bpsw_extend_rows.h)
                break
        j = mj
        while j >= beg and H[j] > 0: j -= 1
        beg = j + 1
        j = mj + 2
        while j <= end and H[j] > 0: j += 1
        end = j
    return np.array([mx, max_j + 1, max_i + 1, max_ie + 1, gscore, max_off])


stats = collections.Counter()
span_hist = collections.Counter()
sides_dp = sides_all = rows = 0
rng = np.random.default_rng(1)
pick = rng.permutation(n)[:n_want]
for tsk in pick:
    reg = int(soa.reg_score[tsk])
    h = int(soa.h0[tsk])
    for side in (0, 1):
        ql = int((soa.right_qlen if side else soa.left_qlen)[tsk]); rl = int((soa.right_rlen if side else soa.left_rlen)[tsk])
        if ql <= 0:
            continue
        qo = int((soa.right_q_off if side else soa.left_q_off)[tsk]); ro = int((soa.right_r_off if side else soa.left_r_off)[tsk])
        q = soa.pool[qo:qo + ql].astype(np.int64); t = soa.pool[ro:ro + rl].astype(np.int64)
        hinit = reg if side else h
        want, _ = orc.sw_extend(q.astype(np.uint8), t.astype(np.uint8), po.default_mat(), soa.o_del, soa.e_del, soa.o_ins, soa.e_ins, soa.w, 5, 100, hinit)
        sides_all += 1
        if kinds[2 * tsk + side] != SIFT_FORM:  # the DP (a side the sift did not examine may still be resolved by ext_kernel's own forms: slight overcount)
            st = collections.Counter()
            got = sweep(q, t, hinit, soa.w, st, 5)
            # (the tail-row bound ends the sweep early without changing the result)
            assert np.array_equal(got, want), (tsk, side, got, want)
            stats.update(st); sides_dp += 1; rows += sum(st.values())
        reg = int(want[0])
print(f"config {cfg}: {len(pick)} tasks, {sides_all} sides, {sides_dp} on the DP ({100 * sides_dp / sides_all:.1f} %), {rows} DP rows = {rows / max(sides_dp, 1):.1f} per DP side, {rows / len(pick):.1f} per task")
tot = sum(stats.values())
for k, v in stats.most_common(24):
    print(f"  {100 * v / tot:5.1f} %  {' '.join(k)}")
for dim, names in ((0, ("cols1", "cols2")), (1, ("live", "dead")), (2, ("atend", "inner")), (3, ("tail", "body")), (4, ("zero", "nozero")), (5, ("imp", "noimp"))):
    print("  ", {nm: round(100 * sum(v for k, v in stats.items() if k[dim] == nm) / tot, 1) for nm in names})

tot_s = sum(span_hist.values())
acc = 0
print("band width of the swept rows (columns, cumulative share): " + " ".join(f"<={8 * k + 7}:{(acc := acc + span_hist[k]) / tot_s:.3f}" for k in sorted(span_hist)))
