"""Parity of the whole boundary-1 call (speculate -> GPU SW -> replay) with the oracle's sequential walk
(native/bwamem_pair.c:115-228 semantics, and the pure-Scala MemSamPe.scala:1111-1369 flavour)."""
import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth
import pyoracle as po
from conftest import region_fields_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [po.RESCUE_C, po.RESCUE_SCALA])
@pytest.mark.parametrize("allo,n,p", [(False, 400, 0.3), (True, 150, 0.5), (False, 64, 0.0)])
def test_group_rescue_matches_oracle(ctx, orc, mode, allo, n, p):
    g = synth.rescue_group(n, seed=900 + n, p_resc=p, all_orientations=allo)
    want_cnt, want, n_sw, _ = orc.matesw_group(orc.default_opt(), g, mode)
    got_cnt, got = ctx.matesw_group(bpsw_hip.default_opt(), g, mode)
    assert np.array_equal(got_cnt, want_cnt)
    region_fields_equal(got, want)
    if p > 0:
        assert n_sw > 0 and got.shape[0] > g.regs.shape[0]   # something was rescued


# BASELINE.json configs[4]'s half of boundary 1: 2x250 bp mates at 8 % / 2 % error, a quarter (and more) of the pairs to rescue.  l_ms * a >= 250
# (the reference leaves KSW_XBYTE off there, native/bwamem_pair.c:179 / MemSamPe.scala:1187-1190), five query columns per lane in the
# packed kernel, the second class of the submission ring, windows 250 bases wider.
@pytest.mark.parametrize("mode", [po.RESCUE_C, po.RESCUE_SCALA])
@pytest.mark.parametrize("allo,n,p,decoy", [(False, 300, 0.25, 0.0), (True, 120, 0.5, 0.0), (False, 200, 0.4, 0.5)])
def test_group_rescue_250bp_matches_oracle(ctx, orc, mode, allo, n, p, decoy):
    g = synth.rescue_group(n, read_len=250, seed=2500 + n, p_resc=p, all_orientations=allo, sub_rate=0.08, indel_rate=0.02,
                           p_multi_anchor=0.3, p_decoy_anchor=decoy)
    assert int(g.seq_len.max()) == 250
    want_cnt, want, n_sw, _ = orc.matesw_group(orc.default_opt(), g, mode)
    s0 = ctx.stats()
    got_cnt, got = ctx.matesw_group(bpsw_hip.default_opt(), g, mode)
    s1 = ctx.stats()
    assert np.array_equal(got_cnt, want_cnt)
    region_fields_equal(got, want)
    assert n_sw > 0 and got.shape[0] > g.regs.shape[0]
    if decoy > 0:
        assert s1.sw_replayed_rounds > s0.sw_replayed_rounds   # a decoy's rescue failed: the true hit's job came in a second round


def test_no_rescue_flag_and_empty_group(ctx, orc):
    g = synth.rescue_group(50, seed=1, p_resc=0.5)
    opt = bpsw_hip.default_opt(); opt.flag = 0x20  # MEM_F_NO_RESCUE
    cnt, regs = ctx.matesw_group(opt, g)
    assert np.array_equal(cnt, g.reg_cnt) and np.array_equal(regs, g.regs)
    g0 = synth.rescue_group(0, seed=1)
    cnt, regs = ctx.matesw_group(bpsw_hip.default_opt(), g0)
    assert cnt.size == 0 and regs.size == 0


def test_speculation_stats(ctx):
    g = synth.rescue_group(300, seed=31, p_resc=0.3, p_multi_anchor=0.5)
    before = ctx.stats()
    ctx.matesw_group(bpsw_hip.default_opt(), g)
    after = ctx.stats()
    assert after.sw_speculated > before.sw_speculated
    assert after.sw_jobs - before.sw_jobs >= after.sw_speculated - before.sw_speculated


@pytest.mark.parametrize("mode", [po.RESCUE_C, po.RESCUE_SCALA])
def test_lean_speculation_second_round(ctx, orc, mode):
    """csrc/bpsw_rescue.cpp launches only the first anchor of an end that has a job and lets the replay ask for the later anchors'
    jobs when they turn out to be needed: with near-duplicate anchors (the first rescue succeeds) nothing is wasted and no second
    round runs; with a decoy in front of the true hit (its rescue finds nothing) the second round must run -- same results as the
    reference's sequential walk (native/bwamem_pair.c:115-156) either way"""
    opt = bpsw_hip.default_opt()
    g = synth.rescue_group(300, seed=41, p_resc=0.4, p_multi_anchor=0.6)
    s0 = ctx.stats()
    got_cnt, got = ctx.matesw_group(opt, g, mode)
    s1 = ctx.stats()
    want_cnt, want, _, _ = orc.matesw_group(orc.default_opt(), g, mode)
    assert np.array_equal(got_cnt, want_cnt)
    region_fields_equal(got, want)
    assert s1.sw_wasted == s0.sw_wasted and s1.sw_replayed_rounds == s0.sw_replayed_rounds
    g = synth.rescue_group(300, seed=42, p_resc=0.5, p_multi_anchor=0.3, p_decoy_anchor=0.5)
    got_cnt, got = ctx.matesw_group(opt, g, mode)
    s2 = ctx.stats()
    want_cnt, want, n_sw, _ = orc.matesw_group(orc.default_opt(), g, mode)
    assert np.array_equal(got_cnt, want_cnt)
    region_fields_equal(got, want)
    assert s2.sw_replayed_rounds > s1.sw_replayed_rounds          # the decoys' rescues failed: the true hits' jobs came second
    assert s2.sw_jobs - s1.sw_jobs == n_sw                         # and exactly the jobs the sequential walk runs were computed
    assert got.shape[0] > g.regs.shape[0]
