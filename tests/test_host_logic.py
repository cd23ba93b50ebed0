"""The host-side logic of the product library against the oracle and, where oracle/_ref exists, the reference's own C -- on
the CPU: memMarkPrimarySe, memPair, memApproxMapqSe (bpsw_tail.cpp) and memSortAndDedup (bpsw_rescue.cpp).  No device."""
import numpy as np
import pytest

import bpsw_hip
import pyoracle as po
from tail_util import synthetic_group


@pytest.mark.parametrize("flavour", [bpsw_hip.TAIL_SCALA, bpsw_hip.TAIL_C])
def test_tail_host_pieces_vs_oracle(orc, flavour):
    opt, oopt, otopt = bpsw_hip.default_opt(), orc.default_opt(), orc.default_tail_opt()
    topt = bpsw_hip.default_tail_opt(flavour)
    n_mp = n_pair = 0
    for seed, kw in ((41, dict(sub_rate=0.02, indel_rate=0.005)), (42, dict(sub_rate=0.05, indel_rate=0.02, p_dup=0.4))):
        pac, g = synthetic_group(orc, 400, 500 + seed, **kw)
        at = 0
        marked = []
        for r in range(2 * g.group_size):
            c = int(g.reg_cnt[r])
            rid = 2 * (g.id0 + r // 2) | (r & 1)
            got = bpsw_hip.mark_primary_se(opt, topt, g.regs[at:at + c], rid)
            want = orc.mark_primary(oopt, otopt, g.regs[at:at + c], rid, flavour)
            assert got.tobytes() == want.tobytes()
            for reg in got[:3]:
                assert bpsw_hip.approx_mapq_se(opt, topt, reg) == orc.approx_mapq(oopt, otopt, reg, flavour)
            marked.append(got)
            n_mp += c > 1
            at += c
        for p in range(g.group_size):
            a0, a1 = marked[2 * p], marked[2 * p + 1]
            if len(a0) and len(a1):
                assert bpsw_hip.mem_pair(opt, topt, g.l_pac, g.pes, a0, a1, g.id0 + p) == orc.mem_pair(oopt, g.l_pac, g.pes, a0, a1, g.id0 + p, flavour)
                n_pair += 1
    assert n_mp > 50 and n_pair > 500


def test_tail_host_pieces_vs_reference(orc, ref):
    opt, oopt, otopt = bpsw_hip.default_opt(), orc.default_opt(), orc.default_tail_opt()
    topt = bpsw_hip.default_tail_opt(bpsw_hip.TAIL_C)
    pac, g = synthetic_group(orc, 500, 777, zdrop_mode=po.ZDROP_BWA, sub_rate=0.03, indel_rate=0.01, p_dup=0.4)
    at = 0
    marked = []
    for r in range(2 * g.group_size):
        c = int(g.reg_cnt[r])
        rid = 2 * (g.id0 + r // 2) | (r & 1)
        got = bpsw_hip.mark_primary_se(opt, topt, g.regs[at:at + c], rid)
        assert got.tobytes() == ref.mark_primary(oopt, otopt, g.regs[at:at + c], rid).tobytes()
        for reg in got[:2]:
            assert bpsw_hip.approx_mapq_se(opt, topt, reg) == ref.approx_mapq(oopt, otopt, reg)
        marked.append(got)
        at += c
    for p in range(g.group_size):
        a0, a1 = marked[2 * p], marked[2 * p + 1]
        if len(a0) and len(a1):
            assert bpsw_hip.mem_pair(opt, topt, g.l_pac, g.pes, a0, a1, g.id0 + p) == ref.mem_pair(oopt, g.l_pac, g.pes, a0, a1, g.id0 + p)


def test_sort_dedup_vs_oracle_and_golden(orc):
    import os
    from tail_util import G
    z = np.load(os.path.join(G, "mem_sort_and_dedup.npz"))
    names = list(z.keys())
    rng = np.random.default_rng(9)
    for mode in (bpsw_hip.RESCUE_C, bpsw_hip.RESCUE_SCALA):
        for _ in range(300):
            n = int(rng.integers(0, 40))
            regs = np.zeros(n, bpsw_hip.ALNREG_DTYPE)
            base = int(rng.integers(1000, 5000))
            regs["rb"] = base + rng.integers(0, 60, n) * rng.integers(1, 4)
            regs["re"] = regs["rb"] + rng.integers(20, 150, n)
            regs["qb"] = rng.integers(0, 60, n)
            regs["qe"] = regs["qb"] + rng.integers(20, 90, n)
            regs["score"] = rng.integers(20, 150, n)
            got = bpsw_hip.sort_dedup(regs, 0.95, mode)
            want = orc.sort_dedup(regs, 0.95, mode)
            assert got.tobytes() == want.tobytes()
    # the reference's own mem_sort_and_dedup (klib introsort order included), straight from the committed golden file
    from conftest import region_fields_equal
    io, oo = z["in_off"], z["out_off"]
    for i in range(len(io) - 1):
        got = bpsw_hip.sort_dedup(z["regs_in"][io[i]:io[i + 1]].astype(bpsw_hip.ALNREG_DTYPE), 0.95, bpsw_hip.RESCUE_C)
        region_fields_equal(got, z["regs_out"][oo[i]:oo[i + 1]])
    assert names


def test_pe_stat_vs_oracle_and_reference(orc, ref):
    """memPeStat: product == oracle in both flavours (and == the reference's mem_pestat in the C flavour), on libraries with one
    and with several supported orientations, and on a batch too small for any."""
    opt, oopt, otopt = bpsw_hip.default_opt(), orc.default_opt(), orc.default_tail_opt()
    for seed, n in ((61, 600), (62, 300), (63, 8)):
        pac, g = synthetic_group(orc, n, 800 + seed, zdrop_mode=po.ZDROP_BWA, sub_rate=0.02, indel_rate=0.004, p_far=0.1)
        for flavour in (bpsw_hip.TAIL_SCALA, bpsw_hip.TAIL_C):
            got = bpsw_hip.pe_stat(opt, bpsw_hip.default_tail_opt(flavour), g.l_pac, g.reg_cnt, g.regs)
            assert got == orc.pe_stat(oopt, otopt, g.l_pac, g.reg_cnt, g.regs, flavour)
        got_c = bpsw_hip.pe_stat(opt, bpsw_hip.default_tail_opt(bpsw_hip.TAIL_C), g.l_pac, g.reg_cnt, g.regs)
        assert got_c == ref.pestat(oopt, otopt, g.l_pac, g.reg_cnt, g.regs)
        if n >= 300:
            assert got_c[1][2] == 0 and 350 < got_c[1][3] < 450 and got_c[0][2] == 1      # FR supported, mean ~ 400; FF not


def test_wait_estimate_is_not_poisoned_by_an_outlier():
    """The estimate the waits nap by (csrc/bpsw_runtime.cpp, wait_est_update): one wait that a descheduled thread stretched to 48 ms must
    not make calls whose work takes 0.1 ms sleep for milliseconds.  (Rounds 4-5 had no bound on a single measurement: the outlier made
    the estimate 12 ms, and since a wait that ends with its nap is as long as the nap, some thirty calls slept 8.4, 7.8, 7.2 ... ms.)"""
    import ctypes as C
    lib = bpsw_hip.load_library()
    upd, naps = lib.bpsw_diag_wait_est_update, lib.bpsw_diag_wait_naps
    upd.restype = C.c_double
    upd.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    naps.argtypes = [C.c_double]
    work = 0.1
    est = 0.0
    for _ in range(20):                      # steady state: no nap below 0.15 ms, every wait is measured
        est = upd(est, work, 3, naps(est))
    assert abs(est - work) < 1e-6 and not naps(est)
    est = upd(est, 48.0, 500, naps(est))     # the outlier
    assert est <= 0.75 * work + 0.25 * (3 * work + 0.2) + 1e-9
    overslept, calls = 0.0, 0
    while naps(est):
        nap = 0.7 * est - 0.06               # wait_nap's sleep (BPSW_WAIT_MODE=0)
        overslept += max(nap - work, 0.0)
        est = upd(est, max(nap, work), 0 if nap >= work else 2, 1)
        calls += 1
        assert calls < 50
    assert overslept < 0.5, (overslept, calls)
    # a real change of the work's length is followed within a dozen calls
    est = 0.1
    for _ in range(12):
        est = upd(est, 2.0, 5, naps(est))
    assert 1.5 < est <= 2.0
