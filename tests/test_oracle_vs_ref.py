"""Cross-checks the oracle against the reference's own C compiled in place (oracle/_ref), live, on larger
seeded samples than the committed fixtures.  Skipped where oracle/_ref/libbwaref.so is absent."""
import numpy as np

import pyoracle as po
from bpsw_hip import synth
from conftest import region_fields_equal

MAT = po.default_mat()


def test_sw_extend_on_synthetic_tasks(orc, ref):
    for L, sub, ind, tail in ((150, 0.01, 0.001, 0.0), (250, 0.1, 0.02, 0.2)):
        soa = synth.ext_tasks(600, read_len=L, sub_rate=sub, indel_rate=ind, tail_frac=tail, seed=31 + L)
        calls = 0
        for i in range(soa.n):
            for side in ("left", "right"):
                ql, rl = int(getattr(soa, side + "_qlen")[i]), int(getattr(soa, side + "_rlen")[i])
                if ql == 0:
                    continue
                q = soa.pool[getattr(soa, side + "_q_off")[i]:][:ql]
                t = soa.pool[getattr(soa, side + "_r_off")[i]:][:rl]
                for w in (100, 200):
                    got, _ = orc.sw_extend(q, t, MAT, 6, 1, 6, 1, w, 5, 100, int(soa.h0[i]), po.ZDROP_BWA)
                    assert np.array_equal(got, ref.ksw_extend2(q, t, MAT, 6, 1, 6, 1, w, 5, 100, int(soa.h0[i])))
                    calls += 1
        assert calls > 1000


def test_sw_align2_on_synthetic_jobs(orc, ref):
    opt = orc.default_opt()
    for L in (100, 150, 250):
        jobs = synth.sw_jobs(150, read_len=L, seed=9 + L)
        xtra = po.KSW_XSUBO | po.KSW_XSTART | (po.KSW_XBYTE if L < 250 else 0) | 19
        got, _ = orc.sw_align2_jobs(opt, xtra, **jobs)
        for i in range(len(got)):
            q = jobs["q_pool"][jobs["q_off"][i]:][:jobs["q_len"][i]]
            if jobs["q_rev"][i]:
                q = np.where(q[::-1] < 4, 3 - q[::-1], 4).astype(np.uint8)
            want = ref.ksw_align2(q, jobs["t_pool"][jobs["t_off"][i]:][:jobs["t_len"][i]], MAT, 6, 1, 6, 1, xtra)
            assert np.array_equal(got[i][[0, 1, 2, 5, 6]], want[[0, 1, 2, 5, 6]])


def test_group_rescue_c_mode(orc, ref):
    opt = orc.default_opt()
    for allo, n, p in ((False, 200, 0.4), (True, 60, 0.5)):
        g = synth.rescue_group(n, seed=55 + n, p_resc=p, all_orientations=allo, p_multi_anchor=0.4)
        cnt, regs, n_sw, _ = orc.matesw_group(opt, g, po.RESCUE_C)
        rcnt, rregs = ref.matesw_group(opt, g)
        assert n_sw > 0 and np.array_equal(cnt, rcnt)
        region_fields_equal(regs, rregs, skip=("csub",))


def test_bns_get_seq_live(orc, ref):
    rng = np.random.default_rng(5)
    l_pac = 200_003
    pac = rng.integers(0, 256, (l_pac + 3) // 4, dtype=np.uint8)
    for t in range(2000):
        b = int(rng.integers(-100, 2 * l_pac))
        e = b + int(rng.integers(0, 1200))
        if t % 5 == 0:
            b = l_pac - int(rng.integers(0, 600)); e = b + int(rng.integers(0, 1200))
        if t % 9 == 0:
            b, e = e, b
        assert np.array_equal(orc.bns_get_seq(l_pac, pac, b, e), ref.bns_get_seq(l_pac, pac, b, e))


def test_chain2aln_live(orc, ref):
    l_pac = 400_009
    pac, bases = synth.random_pac(l_pac, seed=61)
    for L, es, ei, tail in ((150, 0.01, 0.001, 0.0), (250, 0.08, 0.02, 0.05), (100, 0.04, 0.01, 0.2)):
        b = synth.read_chains(1500, bases, l_pac, read_len=L, sub_rate=es, indel_rate=ei, tail_frac=tail, seed=62 + L)
        cnt, regs, n_ext, _ = orc.chain2aln_batch(orc.default_opt(), pac, b, po.ZDROP_BWA)
        rcnt, rregs = ref.chain2aln_batch(orc.default_opt(), pac, b)
        assert np.array_equal(cnt, rcnt)
        for f in regs.dtype.names:
            assert np.array_equal(regs[f], rregs[f]), (L, f)
        assert n_ext > 1000
