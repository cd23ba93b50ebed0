#!/usr/bin/env python3
"""CPU soak of the sift kernel's arithmetic (csrc/bpsw_extend_sift_core.h compiled for the host, tests/sift_host) against the
oracle's full DP: tools/soak_cert2.py's adversarial flanks under rotating gap costs, bands, z-drop settings and matrices; every task
the sift resolves must equal the DP.  No GPU.  Usage: python tools/soak_sift_host.py [seconds] [first_seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402
import soak_cert2 as sc  # noqa: E402
import test_sift_host as tsh  # noqa: E402
from test_extend_gpu import _manual_tasks  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = tsh.sift.__wrapped__() if hasattr(tsh.sift, "__wrapped__") else None
if lib is None:   # the fixture's body without pytest
    import ctypes as C
    import subprocess
    out = os.path.join(ROOT, "tests", "sift_host", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libsift_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "cloud-scale-bwamem_amd", "csrc"), "-o", so,
                    os.path.join(ROOT, "tests", "sift_host", "sift_host.cpp")], check=True)
    lib = C.CDLL(so)
    lib.sift_host_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
orc = po.Oracle()
mats = [po.default_mat(), po.default_mat(), sc._mat(1, -1), sc._mat(2, -3), sc._mat(6, -1)]
t0 = time.time()
rd = seed0
tot = res = 0
while time.time() - t0 < seconds:
    rng = np.random.default_rng(910000 + rd)
    sc.SHORT = bool(rd & 1)
    tasks = []
    for t in range(3000):
        l, r = sc.side(rng), sc.side(rng)
        h0 = int(rng.integers(16, 60)) if rng.random() < 0.5 else int(rng.integers(16, 150))
        if sc.SHORT and rng.random() < 0.3:
            h0 = int(rng.integers(1, 24))
        if rng.random() < 0.1:
            l = ([], [])
        tasks.append((l[0], l[1], r[0], r[1], h0, len(l[0])))
    soa = _manual_tasks(tasks)
    for (od, ed, oi, ei), w in (((6, 1, 6, 1), 100), ((6, 1, 6, 1), 3), ((6, 1, 6, 1), 2), ((1, 1, 1, 1), 100), ((3, 1, 3, 1), 100), ((2, 1, 2, 1), 7),
                                ((3, 2, 7, 1), 2), ((4, 2, 2, 3), 5)):
        soa.o_del, soa.e_del, soa.o_ins, soa.e_ins, soa.w = od, ed, oi, ei, w
        wire = bpsw_hip.wire_pack(soa)
        for zmode, zdrop in ((0, 100), (1, 100), (1, 16), (0, 0)):
            d, n = tsh._run(lib, orc, wire, mats[(rd + zmode + zdrop) % len(mats)], zdrop, zmode)   # asserts on a difference
            res += d
            tot += n
    rd += 1
    print(f"round {rd - seed0} task runs {tot} resolved by the sift {res} differences 0", flush=True)
print("SOAK_SIFT_HOST", {"task_runs": tot, "resolved": res, "bad": 0})
