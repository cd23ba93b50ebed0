"""the memChainToAlnBatched round loop on the device (bpsw_chain2aln_batch) against the oracle's
sequential walk (MemChainToAlignBatched.scala:380-616) and against mem_chain2aln outputs of the reference C."""
import os

import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth
import pyoracle as po

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def refdata():
    l_pac = 700_001
    pac, bases = synth.random_pac(l_pac, seed=71)
    return l_pac, pac, bases


def _same(got_cnt, got, want_cnt, want):
    assert np.array_equal(got_cnt, want_cnt)
    assert got.shape == want.shape
    for f in want.dtype.names:
        assert np.array_equal(got[f], want[f]), f


@pytest.mark.parametrize("zmode", [po.ZDROP_SCALA, po.ZDROP_BWA])
@pytest.mark.parametrize("L,es,ei,tail,n", [(150, 0.01, 0.001, 0.0, 3000), (250, 0.08, 0.02, 0.05, 1500), (100, 0.04, 0.01, 0.2, 1500),
                                            (37, 0.02, 0.0, 0.0, 300)])
def test_round_loop_matches_oracle(ctx, orc, refdata, zmode, L, es, ei, tail, n):
    l_pac, pac, bases = refdata
    ctx.ref_load(pac, l_pac)
    b = synth.read_chains(n, bases, l_pac, read_len=L, sub_rate=es, indel_rate=ei, tail_frac=tail, seed=72 + L)
    want_cnt, want, n_ext, _ = orc.chain2aln_batch(orc.default_opt(), pac, b, zmode)
    got_cnt, got = ctx.chain2aln_batch(bpsw_hip.default_opt(), b, zmode)
    _same(got_cnt, got, want_cnt, want)
    assert n_ext > n // 2 and len(want) < len(b.seed_len)


def test_reference_golden(ctx):
    z = np.load(os.path.join(G, "mem_chain2aln.npz"))
    b = bpsw_hip.ChainBatchSoA(l_pac=int(z["l_pac"]), read_len=z["read_len"], read_off=z["read_off"], read_pool=z["read_pool"],
                               chain_cnt=z["chain_cnt"], seed_cnt=z["seed_cnt"], seed_rbeg=z["seed_rbeg"], seed_qbeg=z["seed_qbeg"],
                               seed_len=z["seed_len"])
    ctx.ref_load(z["pac"], b.l_pac)
    got_cnt, got = ctx.chain2aln_batch(bpsw_hip.default_opt(), b, po.ZDROP_BWA)
    _same(got_cnt, got, z["out_cnt"], z["out_regs"])


@pytest.mark.parametrize("flags,mode", [(bpsw_hip.C2A_SORT_DEDUP, po.RESCUE_C), (bpsw_hip.C2A_SORT_DEDUP | bpsw_hip.C2A_DEDUP_SCALA, po.RESCUE_SCALA)])
def test_sort_dedup_flag(ctx, orc, refdata, flags, mode):
    l_pac, pac, bases = refdata
    ctx.ref_load(pac, l_pac)
    b = synth.read_chains(800, bases, l_pac, read_len=150, sub_rate=0.03, indel_rate=0.005, p_subseed=0.6, p_shifted=0.5, seed=5)
    cnt, regs, _, _ = orc.chain2aln_batch(orc.default_opt(), pac, b, po.ZDROP_SCALA)
    want_cnt, want, at = [], [], 0
    for c in cnt:   # memSortAndDedup per read (BWAMemWorker1Batched.scala:128-133)
        d = orc.sort_dedup(regs[at:at + c].copy(), 0.95, mode)
        want_cnt.append(len(d)); want.append(d); at += c
    got_cnt, got = ctx.chain2aln_batch(bpsw_hip.default_opt(), b, po.ZDROP_SCALA, flags)
    _same(got_cnt, got, np.array(want_cnt, np.int32), np.concatenate(want))
    assert got.shape[0] <= regs.shape[0]


def test_edge_cases_and_errors(ctx, orc, refdata):
    l_pac, pac, bases = refdata
    ctx.ref_load(pac, l_pac)
    mk = lambda **kw: bpsw_hip.ChainBatchSoA(l_pac=l_pac, **kw)
    i32, i64, u8 = (lambda *v: np.array(v, np.int32)), (lambda *v: np.array(v, np.int64)), (lambda v: np.array(v, np.uint8))
    read = synth.window_bases(bases, l_pac, 5000, 5100)
    pool = np.zeros(112, np.uint8); pool[:100] = read
    # a seed spanning the whole read (no extension), a read with no chain, a chain with an empty seed list
    b = mk(read_len=i32(100, 100, 100), read_off=i64(0, 0, 0), read_pool=pool, chain_cnt=i32(1, 0, 1), seed_cnt=i32(1, 0),
           seed_rbeg=i64(5000), seed_qbeg=i32(0), seed_len=i32(100))
    want_cnt, want, _, _ = orc.chain2aln_batch(orc.default_opt(), pac, b)
    got_cnt, got = ctx.chain2aln_batch(bpsw_hip.default_opt(), b)
    _same(got_cnt, got, want_cnt, want)
    assert list(got_cnt) == [1, 0, 0] and got[0]["score"] == 100 and got[0]["qe"] == 100
    # a chain near the strand boundary: getMaxSpan crops the window (MemChainToAlignBatched.scala:669-673)
    rd = synth.window_bases(bases, l_pac, l_pac - 80, l_pac)
    pool2 = np.zeros(112, np.uint8); pool2[20:100] = rd; pool2[:20] = 1
    b2 = mk(read_len=i32(100), read_off=i64(0), read_pool=pool2, chain_cnt=i32(1), seed_cnt=i32(1), seed_rbeg=i64(l_pac - 80),
            seed_qbeg=i32(20), seed_len=i32(60))
    want_cnt, want, _, _ = orc.chain2aln_batch(orc.default_opt(), pac, b2)
    got_cnt, got = ctx.chain2aln_batch(bpsw_hip.default_opt(), b2)
    _same(got_cnt, got, want_cnt, want)
    for bad in (dict(seed_qbeg=i32(60), seed_len=i32(60)),            # seed outside the read
                dict(seed_rbeg=i64(2 * l_pac - 10)),                  # seed outside the reference
                dict(read_len=i32(300))):                             # read longer than the kernel limit
        kw = dict(read_len=i32(100), read_off=i64(0), read_pool=np.zeros(320, np.uint8), chain_cnt=i32(1), seed_cnt=i32(1),
                  seed_rbeg=i64(5000), seed_qbeg=i32(0), seed_len=i32(50))
        kw.update(bad)
        with pytest.raises(bpsw_hip.BpswError):
            ctx.chain2aln_batch(bpsw_hip.default_opt(), mk(**kw))
    opt = bpsw_hip.default_opt(); opt.e_ins = 0
    with pytest.raises(bpsw_hip.BpswError):
        ctx.chain2aln_batch(opt, b)
