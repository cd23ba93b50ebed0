"""BASELINE.json configs[3]-sized coordinate space: a reference of 3 100 000 019 bases (GRCh38-sized; 0.78 GB as 2-bit .pac,
resident in HBM), so that forward coordinates pass 2^31 and reverse-strand coordinates -- [l_pac, 2 l_pac) = [3.1e9, 6.2e9) --
pass 2^32 (util/BNTSeqUtil.scala:37-79, worker2/MemSamPe.scala:1834-1850 work in that doubled space with Long arithmetic).
Every device path that takes reference coordinates is compared with the oracle on reads placed beyond 2^31, beyond 2^32 and
at the ends of both strands: bpsw_ref_fetch (bnsGetSeq), the coordinate-mode group rescue and the device round loop."""
import dataclasses

import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth
import pyoracle as po
from conftest import region_fields_equal

pytestmark = pytest.mark.gpu
L_PAC = 3_100_000_019


@pytest.fixture(scope="module")
def big():
    rng = np.random.default_rng(20261101)
    pac = rng.integers(0, 256, (L_PAC + 3) // 4, dtype=np.uint8)
    ctx = bpsw_hip.Context(0)
    ctx.ref_load(pac, L_PAC)
    yield ctx, pac, synth.PacBases(pac, L_PAC)
    ctx.ref_unload()
    ctx.close()


def test_ref_fetch_beyond_2_31_and_at_the_strand_ends(big):
    ctx, pac, _ = big
    assert ctx.ref_length() == L_PAC
    orc = po.Oracle()
    rng = np.random.default_rng(7)
    edges = [0, 2 ** 31, 2 ** 32, L_PAC, 2 * L_PAC]
    beg, end = [], []
    for e in edges:                               # windows before, across and after every interesting coordinate
        for d in (-700, -300, -1, 0, 1, 250):
            b = e + d
            beg.append(b); end.append(b + int(rng.integers(1, 600)))
    for _ in range(200):                          # and anywhere in the doubled space, sometimes swapped (bnsGetSeq swaps)
        b = int(rng.integers(0, 2 * L_PAC - 700))
        e = b + int(rng.integers(1, 700))
        if rng.random() < 0.2:
            b, e = e, b
        beg.append(b); end.append(e)
    got, _ = ctx.ref_fetch(np.array(beg, np.int64), np.array(end, np.int64))
    n_beyond = 0
    for b, e, g in zip(beg, end, got):
        want = orc.bns_get_seq(L_PAC, pac, b, e)
        assert np.array_equal(g, want), (b, e)
        n_beyond += int(len(want) > 0 and max(b, e) > 2 ** 32)
    assert n_beyond > 20


def test_group_rescue_by_coordinates_beyond_2_31(big):
    ctx, pac, bases = big
    orc = po.Oracle()
    # forward starts: right behind the start of the strand, either side of 2^31, right before the end of the strand (whose
    # mates then sit at the START of the reverse strand, coordinates just above l_pac); the mates of the first ones sit at
    # the END of the doubled space, just below 2 l_pac = 6.2e9
    pos = [2000, 2100, 2 ** 31 - 700, 2 ** 31 - 100, 2 ** 31 + 5, L_PAC - 3100, L_PAC - 3001] * 6
    g = synth.rescue_group(240, seed=20261102, l_pac=L_PAC, p_resc=0.5, ref_bases=bases, positions=pos)
    assert int((g.regs["rb"] > 2 ** 32).sum()) > 50 and int(((g.regs["rb"] > 2 ** 31) & (g.regs["rb"] < L_PAC)).sum()) > 20
    for mode in (bpsw_hip.RESCUE_C, bpsw_hip.RESCUE_SCALA):
        want_cnt, want, n_sw, _ = orc.matesw_group(orc.default_opt(), g, mode)           # windows as bytes
        gc = dataclasses.replace(g, ref_pool=None, ref_len=None, ref_off=None)            # coordinates only
        got_cnt, got = ctx.matesw_group(bpsw_hip.default_opt(), gc, mode)
        assert np.array_equal(got_cnt, want_cnt)
        region_fields_equal(got, want)
        assert n_sw > 80 and got.shape[0] > g.regs.shape[0]                               # mates were rescued from the resident reference
    assert int((got["rb"] > 2 ** 32).sum()) > int((g.regs["rb"] > 2 ** 32).sum())       # ... also beyond 2^32


def test_round_loop_on_reads_beyond_2_31(big):
    ctx, pac, bases = big
    orc = po.Oracle()
    L = 150
    pos = [600, 2 ** 31 - 80, 2 ** 31 + 3, L_PAC - L - 661, L_PAC + 600, 2 ** 32 - 75, 2 ** 32 + 9, 2 * L_PAC - L - 661] * 5
    b = synth.read_chains(400, bases, L_PAC, read_len=L, sub_rate=0.02, indel_rate=0.004, seed=20261103, positions=pos)
    assert int((b.seed_rbeg > 2 ** 32).sum()) > 100
    for zmode in (po.ZDROP_SCALA, po.ZDROP_BWA):
        want_cnt, want, _, _ = orc.chain2aln_batch(orc.default_opt(), pac, b, zmode)
        got_cnt, got = ctx.chain2aln_batch(bpsw_hip.default_opt(), b, zdrop_mode=zmode)
        assert np.array_equal(got_cnt, want_cnt)
        region_fields_equal(got, want)
    assert int((got["rb"] > 2 ** 32).sum()) > 50
