"""SURVEY.md 8(f).2 for boundary 2: a coordinate batch (wire format 2, include/bpsw.h) names the target flanks of every task by
the seed's reference coordinates; the device reads them from the resident 2-bit reference (bnsGetSeq,
util/BNTSeqUtil.scala:37-79) instead of receiving leftRs / rightRs (MemChainToAlignBatched.scala:363, 511-517, 534-541).
Results must equal the oracle's on the byte tasks the Scala driver would have built for the same seeds."""
import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth
import pyoracle as po
from test_jni_shim import _extend, fake  # noqa: F401  (the fake-JVM fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def refctx():
    c = bpsw_hip.Context(0)
    yield c
    c.close()


def _tasks(l_pac, n_reads, read_len, sub, indel, seed, **kw):
    pac, bases = synth.random_pac(l_pac, seed=seed)
    chains = synth.read_chains(n_reads, bases, l_pac, read_len=read_len, sub_rate=sub, indel_rate=indel, seed=seed + 1, **kw)
    co, by = synth.coord_ext_tasks(chains, bases, seed=seed + 2)
    return pac, bases, co, by


@pytest.mark.parametrize("read_len,sub,indel", [(150, 0.01, 0.001), (150, 0.05, 0.01), (250, 0.08, 0.02)])
def test_coordinate_batch_equals_byte_batch_and_oracle(refctx, orc, read_len, sub, indel):
    l_pac = 1_000_003
    pac, bases, co, by = _tasks(l_pac, 700, read_len, sub, indel, seed=3100 + read_len, tail_frac=0.05)
    assert co.n > 700
    refctx.ref_load(pac, l_pac)
    wire_b, wire_c = bpsw_hip.wire_pack(by), bpsw_hip.wire_coords_pack(co)
    assert wire_c.size < 0.75 * wire_b.size           # what the coordinates save on the wire
    want, _ = orc.wire_extend(wire_b)
    want = np.asarray(want).reshape(-1)
    got_b = refctx.extend_batch(wire_b)
    got_c = refctx.extend_batch(wire_c)
    assert np.array_equal(got_b, want)
    assert np.array_equal(got_c, want)


def test_windows_used_for_the_byte_tasks_are_bns_get_seq(orc):
    """the generator's window_bases is the oracle's bnsGetSeq: pins what "the same task as bytes" means"""
    l_pac = 200_003
    pac, bases = synth.random_pac(l_pac, seed=41)
    rng = np.random.default_rng(42)
    for _ in range(300):
        rb = int(rng.integers(0, 2 * l_pac - 400))
        re = rb + int(rng.integers(1, 300))
        if rb < l_pac < re:
            continue
        assert np.array_equal(synth.window_bases(bases, l_pac, rb, re), orc.bns_get_seq(l_pac, pac, rb, re))


def test_both_strands_strand_ends_and_scorings(refctx, orc):
    l_pac = 300_007
    pac, bases = synth.random_pac(l_pac, seed=51)
    L = 150
    # reads that start right at the strand ends: the flanks are clipped to the strand, as getMaxSpan clips rmax (:667-675)
    pos = [0, 3, l_pac - L - 61, l_pac - L - 70, l_pac + 1, l_pac, 2 * l_pac - L - 61, 2 * l_pac - L - 90]
    chains = synth.read_chains(400, bases, l_pac, read_len=L, sub_rate=0.03, indel_rate=0.005, seed=52, positions=pos)
    co, by = synth.coord_ext_tasks(chains, bases, seed=53)
    refctx.ref_load(pac, l_pac)
    for (o_del, e_del, o_ins, e_ins, w, zmode) in ((6, 1, 6, 1, 100, po.ZDROP_SCALA), (4, 2, 7, 1, 30, po.ZDROP_BWA), (6, 1, 6, 1, 3, po.ZDROP_SCALA)):
        for t in (co, by):
            t.o_del, t.e_del, t.o_ins, t.e_ins, t.w = o_del, e_del, o_ins, e_ins, w
        refctx.set_ext_scoring(None, 100, zmode)
        want, _ = orc.wire_extend(bpsw_hip.wire_pack(by), zdrop_mode=zmode)
        got = refctx.extend_batch(bpsw_hip.wire_coords_pack(co))
        assert np.array_equal(got, np.asarray(want).reshape(-1)), (o_del, e_del, o_ins, e_ins, w)


def test_coordinate_batch_shortcuts_off_equals_on(refctx):
    l_pac = 500_009
    pac, bases, co, by = _tasks(l_pac, 500, 150, 0.01, 0.001, seed=61)
    refctx.ref_load(pac, l_pac)
    wire_c = bpsw_hip.wire_coords_pack(co)
    on = refctx.extend_batch(wire_c)
    refctx.set_ext_shortcuts(0)
    off = refctx.extend_batch(wire_c)
    refctx.set_ext_shortcuts(-1)
    assert np.array_equal(on, off)


def test_rejects_what_it_cannot_run(refctx):
    l_pac = 100_003
    pac, bases, co, by = _tasks(l_pac, 50, 150, 0.01, 0.001, seed=71)
    wire_c = bpsw_hip.wire_coords_pack(co)
    other, _ = synth.random_pac(5_003, seed=72)
    refctx.ref_load(other, 5_003)
    with pytest.raises(bpsw_hip.BpswError):     # windows outside the (too short) loaded reference
        refctx.extend_batch(wire_c)
    refctx.ref_load(pac, l_pac)
    refctx.extend_batch(wire_c)
    bad = wire_c.copy()
    bad[7] = 9
    with pytest.raises(bpsw_hip.BpswError):     # unknown format byte
        refctx.extend_batch(bad)
    bad = wire_c.copy()
    rb = np.array([l_pac - 5], np.int64)         # first task: flanks bridge the two strands
    bad[32 + 32: 32 + 40] = rb.view(np.uint8)
    with pytest.raises(bpsw_hip.BpswError):
        refctx.extend_batch(bad)
    # a seed coordinate near INT64_MAX / INT64_MIN: rb + len + rr used to wrap and pass the window tests (the kernel would
    # then have read the reference at a garbage offset); it is range-checked before any arithmetic now
    for evil in (np.iinfo(np.int64).max - 10, np.iinfo(np.int64).max, np.iinfo(np.int64).min, np.iinfo(np.int64).min + 70_000,
                 -1, 2 * l_pac + 1):
        bad = wire_c.copy()
        bad[32 + 32: 32 + 40] = np.array([evil], np.int64).view(np.uint8)
        with pytest.raises(bpsw_hip.BpswError):
            refctx.extend_batch(bad)
    assert np.array_equal(refctx.extend_batch(wire_c), refctx.extend_batch(bpsw_hip.wire_pack(by)))  # the context still works


def test_coordinate_batch_through_the_jni_symbol(refctx, fake):
    """the same swExtendFPGAJNI(n, bytes) carries a coordinate batch: no new native method on the Scala side"""
    l_pac = 400_009
    pac, bases, co, by = _tasks(l_pac, 300, 150, 0.02, 0.002, seed=81)
    refctx.ref_load(pac, l_pac)               # the reference is per device: the shim's thread context sees it
    wire_c = bpsw_hip.wire_coords_pack(co)
    want = refctx.extend_batch(bpsw_hip.wire_pack(by))
    rc, got, msg = _extend(fake, wire_c, co.n, 2)
    assert rc == 0, msg
    assert np.array_equal(got, want)


def test_reference_backed_c_generator_both_forms_agree(refctx, orc):
    """csrc/bpsw_synth.cpp bpsw_synth_ext_tasks_ref (what `bench.py` ships as coordinate batches): the byte form and the coordinate
    form of the same seeds give the same results, equal to the oracle; the target flanks of the byte form are bnsGetSeq's windows"""
    l_pac = 3_000_017
    pac = synth.hash_pac(l_pac, seed=91)
    refctx.ref_load(pac, l_pac)
    by, co = synth.ext_tasks_ref(3000, pac, l_pac, read_len=150, sub_rate=0.02, indel_rate=0.002, seed=92)
    assert by.n == co.n and by.n > 2000 and (co.seed_rbeg >= l_pac).any() and (co.seed_rbeg < l_pac).any()   # both strands
    wire_b, wire_c = bpsw_hip.wire_pack(by), bpsw_hip.wire_coords_pack(co)
    want, _ = orc.wire_extend(wire_b)
    assert np.array_equal(refctx.extend_batch(wire_b), want)
    assert np.array_equal(refctx.extend_batch(wire_c), want)
    for t in range(0, by.n, 97):
        rb, ln, lr, rr = int(co.seed_rbeg[t]), int(co.seed_len[t]), int(by.left_rlen[t]), int(by.right_rlen[t])
        if rr:
            assert np.array_equal(np.asarray(orc.bns_get_seq(l_pac, pac, rb + ln, rb + ln + rr)), by.pool[int(by.right_r_off[t]): int(by.right_r_off[t]) + rr])
        if lr:
            assert np.array_equal(np.asarray(orc.bns_get_seq(l_pac, pac, rb - lr, rb))[::-1], by.pool[int(by.left_r_off[t]): int(by.left_r_off[t]) + lr])
    assert wire_c.size < 0.7 * wire_b.size


@pytest.mark.parametrize("sub,indel", [(0.01, 0.001), (0.03, 0.006)])
def test_sift_kernel_on_coordinate_batches(refctx, orc, sub, indel):
    """csrc/bpsw_extend_sift.hip reads the target flanks of a coordinate batch from the 2-bit reference (both strands, the left
    flank backwards): same results and the same per-side verdicts as ext_kernel alone, equal to the oracle on the byte form"""
    l_pac = 2_000_003
    pac = synth.hash_pac(l_pac, seed=191)
    refctx.ref_load(pac, l_pac)
    by, co = synth.ext_tasks_ref(5000, pac, l_pac, read_len=150, sub_rate=sub, indel_rate=indel, seed=192 + int(sub * 1000))
    assert (co.seed_rbeg >= l_pac).any() and (co.seed_rbeg < l_pac).any()
    wire_c = bpsw_hip.wire_coords_pack(co)
    want, _ = orc.wire_extend(bpsw_hip.wire_pack(by))
    try:
        for level in (1, 1 | 2 | 4, 31):
            refctx.set_ext_shortcuts(level)
            out0, how0 = refctx.extend_batch_classify(wire_c)
            refctx.set_ext_shortcuts(level | 32)
            out1, how1 = refctx.extend_batch_classify(wire_c)
            assert np.array_equal(out0, want) and np.array_equal(out1, want), level
            assert np.array_equal(how0, how1), level
        assert (how1 == 1).sum() > 0.2 * (how1 != 0).sum()
    finally:
        refctx.set_ext_shortcuts(-1)
