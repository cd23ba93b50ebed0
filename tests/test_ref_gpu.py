"""the 2-bit reference resident in HBM.  bpsw_ref_fetch is bnsGetSeq (util/BNTSeqUtil.scala:37-79)
on the device; SW jobs and rescue groups that name their windows by coordinates must give exactly the results of the
same jobs shipped as bytes."""
import os

import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth
import pyoracle as po
from conftest import region_fields_equal

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
XTRA = po.KSW_XSUBO | po.KSW_XSTART | po.KSW_XBYTE | 19


@pytest.fixture()
def refctx():
    c = bpsw_hip.Context(0)
    yield c
    c.close()


def test_ref_fetch_matches_reference_golden(refctx):
    z = np.load(os.path.join(G, "bns_get_seq.npz"))
    l_pac, pac = int(z["l_pac"]), z["pac"]
    refctx.ref_load(pac, l_pac)
    assert refctx.ref_length() == l_pac
    seqs, lens = refctx.ref_fetch(z["beg"], z["end"])
    off, pool = z["seq_off"], z["seq_pool"]
    for i, s in enumerate(seqs):
        assert np.array_equal(s, pool[off[i]:off[i + 1]]), i
    assert np.array_equal(lens, np.diff(off))


def test_ref_fetch_matches_oracle_large(refctx, orc):
    l_pac = 3_000_017
    pac, _ = synth.random_pac(l_pac, seed=11)
    refctx.ref_load(pac, l_pac)
    rng = np.random.default_rng(12)
    beg = rng.integers(-300, 2 * l_pac, 3000)
    end = beg + rng.integers(0, 2000, 3000)
    beg[::7] = l_pac - rng.integers(0, 900, len(beg[::7]))   # around the strand boundary
    end[::7] = beg[::7] + rng.integers(0, 1800, len(beg[::7]))
    beg[::13], end[::13] = end[::13].copy(), beg[::13].copy()  # swapped
    seqs, _ = refctx.ref_fetch(beg, end)
    for b, e, s in zip(beg, end, seqs):
        assert np.array_equal(s, orc.bns_get_seq(l_pac, pac, int(b), int(e)))


def test_swalign2_by_coordinates_equals_bytes(refctx, orc):
    l_pac = 400_009
    pac, bases = synth.random_pac(l_pac, seed=21)
    refctx.ref_load(pac, l_pac)
    rng = np.random.default_rng(22)
    n, L = 300, 150
    qs, t_off, t_len, q_rev, wins = [], [], [], [], []
    for j in range(n):
        tl = int(rng.integers(300, 900))
        rev = j % 2
        rb = int(rng.integers(l_pac, 2 * l_pac - tl)) if rev else int(rng.integers(0, l_pac - tl))
        win = synth.window_bases(bases, l_pac, rb, rb + tl)
        assert len(win) == tl
        p = int(rng.integers(0, tl - L))
        q = win[p:p + L].copy()
        mut = rng.random(L) < 0.04
        q[mut] = (q[mut] + 1 + rng.integers(0, 3, int(mut.sum()))) & 3
        qr = j % 3 == 0
        if qr:  # the kernel reverse-complements the mate on the fly (MemSamPe.scala:1175-1184)
            q = (3 - q)[::-1].copy()
        qs.append(q); t_off.append(rb); t_len.append(tl); q_rev.append(1 if qr else 0); wins.append(win)
    q_off = np.arange(n, dtype=np.int64) * 160
    q_pool = np.zeros(n * 160, np.uint8)
    for j, q in enumerate(qs):
        q_pool[q_off[j]:q_off[j] + L] = q
    opt = bpsw_hip.default_opt()
    got = refctx.swalign2_batch(opt, XTRA, [L] * n, t_len, q_off, t_off, q_rev, q_pool, None)
    # the same jobs with the windows shipped as bytes
    b_off = np.zeros(n, np.int64)
    b_off[1:] = np.cumsum([(len(w) + 15) & ~15 for w in wins])[:-1]
    t_pool = np.zeros(int(b_off[-1]) + 1024, np.uint8)
    for j, w in enumerate(wins):
        t_pool[b_off[j]:b_off[j] + len(w)] = w
    want = refctx.swalign2_batch(opt, XTRA, [L] * n, t_len, q_off, b_off, q_rev, q_pool, t_pool)
    assert np.array_equal(got, want)
    ora, _ = orc.sw_align2_jobs(orc.default_opt(), XTRA, np.array([L] * n), np.array(t_len), q_off, b_off, np.array(q_rev, np.uint8),
                                q_pool, t_pool)
    assert np.array_equal(got, ora)
    assert (got[:, 0] > 100).mean() > 0.9


@pytest.mark.parametrize("mode", [po.RESCUE_C, po.RESCUE_SCALA])
def test_group_rescue_by_coordinates_equals_bytes(refctx, orc, mode):
    l_pac = 600_011
    pac, bases = synth.random_pac(l_pac, seed=31)
    refctx.ref_load(pac, l_pac)
    g = synth.rescue_group(500, seed=33, l_pac=l_pac, p_resc=0.4, ref_bases=bases)
    want_cnt, want, n_sw, _ = orc.matesw_group(orc.default_opt(), g, mode)
    byte_cnt, byte_regs = refctx.matesw_group(bpsw_hip.default_opt(), g, mode)
    assert np.array_equal(byte_cnt, want_cnt)
    region_fields_equal(byte_regs, want)
    h2d0 = refctx.stats().sw_h2d_ms
    import dataclasses
    gc = dataclasses.replace(g, ref_pool=None, ref_len=None, ref_off=None)   # coordinates only
    got_cnt, got = refctx.matesw_group(bpsw_hip.default_opt(), gc, mode)
    assert np.array_equal(got_cnt, want_cnt)
    region_fields_equal(got, want)
    assert n_sw > 50 and got.shape[0] > g.regs.shape[0]   # mates were rescued from the reference itself


def test_coordinate_mode_errors(refctx):
    g = synth.rescue_group(20, seed=3, l_pac=100_003, p_resc=0.5, ref_bases=synth.random_pac(100_003, seed=4)[1])
    import dataclasses
    gc = dataclasses.replace(g, ref_pool=None, ref_len=None, ref_off=None)
    with pytest.raises(bpsw_hip.BpswError):          # no reference loaded
        refctx.matesw_group(bpsw_hip.default_opt(), gc)
    pac, _ = synth.random_pac(50_021, seed=5)
    refctx.ref_load(pac, 50_021)
    with pytest.raises(bpsw_hip.BpswError):          # l_pac of the group differs from the loaded reference
        refctx.matesw_group(bpsw_hip.default_opt(), gc)
    q_pool = np.zeros(160, np.uint8)
    with pytest.raises(bpsw_hip.BpswError):          # a window bridging the strands is not a bnsGetSeq window
        refctx.swalign2_batch(bpsw_hip.default_opt(), XTRA, [150], [400], [0], [50_021 - 200], [0], q_pool, None)
    with pytest.raises(bpsw_hip.BpswError):          # beyond the doubled coordinate space
        refctx.swalign2_batch(bpsw_hip.default_opt(), XTRA, [150], [400], [0], [2 * 50_021 - 100], [0], q_pool, None)
    refctx.ref_unload()
    assert refctx.ref_length() == 0
