"""The arithmetic of the sift kernel (csrc/bpsw_extend_sift_core.h: word-parallel closed form, the single-gap certificate without
scans, two-gap-open tests, start-gap form, the chaining of the two sides) compiled for the HOST (tests/sift_host/sift_host.cpp) and
held against the oracle's full DP (SWUtil.scala:61-230 / MemChainToAlignBatched.scala:789-883 as oracle/bpsw_oracle.c restates
them): every task the sift resolves must come out as the DP says.  No GPU: the same header is what ext_sift_kernel compiles, and
the -m gpu tests compare the kernel's verdicts with ext_kernel's own wave-wide evaluation of the same forms."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth
import pyoracle as po
import exhaustive_flanks as ef

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def sift():
    out = os.path.join(HERE, "sift_host", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libsift_host.so")
    src = os.path.join(HERE, "sift_host", "sift_host.cpp")
    hdr = os.path.join(ROOT, "cloud-scale-bwamem_amd", "csrc", "bpsw_extend_sift_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.dirname(hdr), "-o", so, src], check=True)
    lib = C.CDLL(so)
    lib.sift_host_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def _levels(mat):
    """exact_match_score / sift_uniform_dm / certify_level of csrc/bpsw_runtime.cpp"""
    m = np.asarray(mat, np.int64).reshape(5, 5)
    a = int(m[0, 0])
    diag = all(m[i, i] == a for i in range(4))
    others = [m[r, c] for r in range(5) for c in range(5) if not (r == c and r < 4)]
    exact_a = a if (a > 0 and diag and all(v < a for v in others)) else 0
    mm = int(m[0, 1])
    uniform = all(m[i, j] == mm for i in range(4) for j in range(4) if i != j)
    dm = exact_a - mm if (exact_a > 0 and uniform and exact_a - mm > 0) else 0
    level = 0
    if exact_a > 0:
        level = 3 if (exact_a == 1 and all(v <= -1 for v in others)) else 1
    return exact_a, dm, level


def _run(lib, orc, wire, mat, zdrop, zmode, certify_cap=3):
    exact_a, dm, level = _levels(mat)
    n = int(np.frombuffer(wire[8:12].tobytes(), "<i4")[0])
    out = np.zeros(10 * max(n, 1), np.int16)
    flag = np.zeros(max(n, 1), np.uint8)
    kinds = np.zeros(2 * max(n, 1), np.uint8)
    w32 = np.ascontiguousarray(wire).view(np.uint32)
    rc = lib.sift_host_batch(w32.ctypes.data, w32.size, n, zdrop, min(level, certify_cap), exact_a, dm, 127, out.ctypes.data, flag.ctypes.data, kinds.ctypes.data)
    assert rc == 0
    want, _ = orc.wire_extend(wire, mat, zdrop, zmode)
    want = np.asarray(want).reshape(-1, 10)
    got = out.reshape(-1, 10)[:n]
    done = flag[:n] == 1
    bad = np.nonzero(done & (got != want).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of {int(done.sum())} resolved tasks differ from the DP; first {bad[:3]}: got {got[bad[:3]]} want {want[bad[:3]]}"
    return int(done.sum()), n


def test_resolved_tasks_equal_the_dp_on_read_like_batches(sift, orc):
    tot = res = 0
    for sub, indel, n_rate in ((0.005, 0.0005, 0.0), (0.01, 0.001, 0.001), (0.03, 0.005, 0.0), (0.06, 0.01, 0.002)):
        soa = synth.ext_tasks(6000, read_len=150, sub_rate=sub, indel_rate=indel, seed=8100 + int(sub * 1e4))
        if n_rate:
            pool = soa.pool.copy()
            pool[np.random.default_rng(5).random(pool.size) < n_rate] = 4
            soa.pool = pool
        for (o, e, w) in ((6, 1, 100), (4, 2, 30), (5, 1, 3), (1, 1, 100)):
            soa.o_del = soa.o_ins = o
            soa.e_del = soa.e_ins = e
            soa.w = w
            wire = bpsw_hip.wire_pack(soa)
            for zmode, zdrop in ((po.ZDROP_SCALA, 100), (po.ZDROP_BWA, 10)):
                for cap in (1, 3):
                    d, n = _run(sift, orc, wire, po.default_mat(), zdrop, zmode, cap)
                    res += d
                    tot += n
    assert res > 0.1 * tot          # the forms do resolve a good part of these batches (a third at low error rates, few at 6 %)


def test_resolved_tasks_equal_the_dp_on_adversarial_flanks(sift, orc):
    """tools/soak_cert2.py's generator: deficits at the forms' limits, low-complexity and periodic sequence, shifted diagonals that
    match more often than by chance, one-base gaps at the start; several matrices of the family and one mild mismatch score"""
    import soak_cert2 as sc
    from test_extend_gpu import _manual_tasks
    mats = [po.default_mat(), sc._mat(1, -1), sc._mat(2, -3), sc._mat(6, -1)]
    res = tot = 0
    for rd in range(6):
        rng = np.random.default_rng(51000 + rd)
        tasks = []
        for t in range(3000):
            l, r = sc.side(rng), sc.side(rng)
            h0 = int(rng.integers(16, 60)) if rng.random() < 0.5 else int(rng.integers(16, 150))
            if rng.random() < 0.1:
                l = ([], [])
            tasks.append((l[0], l[1], r[0], r[1], h0, len(l[0])))
        soa = _manual_tasks(tasks)
        for (od, ed, oi, ei), w in (((6, 1, 6, 1), 100), ((6, 1, 6, 1), 3), ((1, 1, 1, 1), 100), ((3, 1, 3, 1), 100), ((2, 1, 2, 1), 7), ((3, 2, 7, 1), 2)):
            soa.o_del, soa.e_del, soa.o_ins, soa.e_ins, soa.w = od, ed, oi, ei, w
            wire = bpsw_hip.wire_pack(soa)
            for zmode, zdrop in ((0, 100), (1, 16), (0, 0)):
                d, n = _run(sift, orc, wire, mats[(rd + zmode + zdrop) % len(mats)], zdrop, zmode)
                res += d
                tot += n
    assert res > 0.05 * tot


def test_resolved_tasks_equal_the_dp_on_every_short_flank(sift, orc):
    """every query of up to 6 bases against every target variant (tests/exhaustive_flanks.py), as left and as right flank"""
    F = ef.enumerate_flanks(max_q=6, extra_t=3, n_variants_up_to=5)
    res = tot = 0
    for mat, o in ((po.default_mat(), 6), (np.asarray(po.default_mat(1, 1)), 1)):
        for h0 in (5, 19, 40):
            for left in (False, True):
                soa = ef.flank_tasks(*F, h0=h0, left=left)
                soa.o_del = soa.o_ins = o
                soa.e_del = soa.e_ins = 1
                soa.w = 100
                d, n = _run(sift, orc, bpsw_hip.wire_pack(soa), mat, 100, po.ZDROP_SCALA)
                res += d
                tot += n
    assert res > 0 and tot > 300_000
