"""The host-buffer entry points as round 2 left them (csrc/bpsw_runtime.cpp, bpsw_sw_runtime.cpp, bpsw_rescue.cpp): pooled device
streams, sleeping waits, zero-copy small transfers, the single-pass rescue host layer, the native feeder bench.py drives them with.
Every variant must give the oracle's bytes."""
import os
import subprocess
import sys

import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth
import pyoracle as po
from conftest import region_fields_equal

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_feeder_threads_through_both_boundaries(orc):
    """bench.py's timed region in small: 12 native threads, one context each, 6 wire batches + 24 rescue groups, dynamic assignment;
    every output equals the oracle's"""
    from bpsw_hip import feeder as fd
    soas = [synth.ext_tasks(3000, seed=700 + b) for b in range(6)]
    wires = [bpsw_hip.wire_pack(s) for s in soas]
    groups = [synth.rescue_group_fast(256, seed=800 + g, p_resc=0.3) for g in range(24)]
    ext_outs = [np.zeros(10 * s.n, np.int16) for s in soas]
    structs = [g.as_struct() for g in groups]
    cnts = [np.zeros(2 * g.group_size, np.int32) for g in groups]
    regs = [np.empty(int(g.regs.shape[0] + g.ref_rb.shape[0] + 16), bpsw_hip.ALNREG_DTYPE) for g in groups]
    items, order = fd.make_items(wires, ext_outs, groups, structs, cnts, regs)
    F = fd.Feeder(12, 0, bpsw_hip.default_opt())
    for _ in range(3):
        F.run(items)
    totals = {i: it.out_total for it, (kind, i) in zip(items, order) if kind == 1}
    st = F.stats_sum()
    F.close()
    for w, out in zip(wires, ext_outs):
        want, _ = orc.wire_extend(w)
        assert np.array_equal(out, want)
    for i, g in enumerate(groups):
        wcnt, wregs, _, _ = orc.matesw_group(orc.default_opt(), g, po.RESCUE_C)
        assert np.array_equal(cnts[i], wcnt)
        region_fields_equal(regs[i][: totals[i]], wregs)
    assert st["ext_calls"] == 18 and st["grp_calls"] == 72 and st["grp_pairs"] == 72 * 256


def test_rescue_group_edge_cases(ctx, orc):
    """empty group, a group without any rescue job, MEM_F_NO_RESCUE, a group whose every pair needs a job, tiny groups"""
    opt, oopt = bpsw_hip.default_opt(), orc.default_opt()
    for g in (synth.rescue_group_fast(64, seed=11, p_resc=0.0, p_multi_anchor=0.0, p_wrong_mate=0.0), synth.rescue_group_fast(64, seed=12, p_resc=1.0),
              synth.rescue_group(1, seed=13, p_resc=1.0), synth.rescue_group(3, seed=14, p_resc=0.5, all_orientations=True)):
        for mode in (bpsw_hip.RESCUE_C, bpsw_hip.RESCUE_SCALA):
            cnt, regs = ctx.matesw_group(opt, g, mode)
            wcnt, wregs, _, _ = orc.matesw_group(oopt, g, mode)
            assert np.array_equal(cnt, wcnt)
            region_fields_equal(regs, wregs)
    g = synth.rescue_group_fast(200, seed=15, p_resc=0.5)
    opt.flag, oopt.flag = 0x20, 0x20          # MEM_F_NO_RESCUE: the lists come back as they went in
    cnt, regs = ctx.matesw_group(opt, g)
    assert np.array_equal(cnt, g.reg_cnt) and np.array_equal(regs, g.regs)
    wcnt, wregs, n_sw, _ = orc.matesw_group(oopt, g, po.RESCUE_C)
    assert n_sw == 0 and np.array_equal(wcnt, cnt)
    import dataclasses
    empty = dataclasses.replace(g, group_size=0)
    cnt, regs = ctx.matesw_group(bpsw_hip.default_opt(), empty)
    assert cnt.shape[0] == 0 and regs.shape[0] == 0


def test_classify_reports_how_each_side_was_produced(ctx, orc):
    soa = synth.ext_tasks(4000, seed=21)
    wire = bpsw_hip.wire_pack(soa)
    out, how = ctx.extend_batch_classify(wire)
    want, _, side_cells = orc.wire_extend_sides(wire)
    assert np.array_equal(out, want)
    assert set(np.unique(how)) <= {0, 1, 2}
    empty_side = np.stack([soa.left_qlen == 0, soa.right_qlen == 0], axis=1)
    assert np.array_equal(how == 0, empty_side)               # 0 exactly for the empty sides
    assert (how == 1).sum() > 0.3 * (how != 0).sum()          # 2x150 bp at 1 %: a large share of the sides is closed form ...
    assert (how == 2).sum() > 0                               # ... and some run the DP
    assert int(side_cells[how == 0].sum()) == 0
    ctx.set_ext_shortcuts(0)
    try:
        out0, how0 = ctx.extend_batch_classify(wire)
    finally:
        ctx.set_ext_shortcuts(-1)
    assert np.array_equal(out0, want) and not (how0 == 1).any()


def test_negative_band_width_is_rejected(ctx):
    soa = synth.ext_tasks(64, seed=22)
    soa.w = 200                                # does not fit the signed byte of the wire header (MemChainToAlignBatched.scala:78-84)
    wire = bpsw_hip.wire_pack(soa)
    assert np.frombuffer(wire[6:7].tobytes(), np.int8)[0] < 0
    with pytest.raises(bpsw_hip.BpswError):
        ctx.extend_batch(wire)


@pytest.mark.parametrize("env", [{"BPSW_ZEROCOPY": "0"}, {"BPSW_STREAM_POOL": "0", "BPSW_SPIN_WAIT": "1"}, {"BPSW_STREAM_POOL": "3", "BPSW_STREAM_SHARE": "2"},
                                 {"BPSW_EXT_CHUNK": "1", "BPSW_SW_KEYS_LDS": "0"}])
def test_parity_with_the_round_2_mechanisms_switched_off(env):
    """the switches are read once per process: re-run the parity tests of both boundaries in a subprocess with copies instead of
    zero-copy / a private stream per context and the runtime's busy wait / a tiny shared pool / one task per dequeue and HBM row keys"""
    if any(os.environ.get(k) == v for k, v in env.items()):
        pytest.skip("already running with this setting")
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(HERE, "test_extend_gpu.py"),
                        os.path.join(HERE, "test_swalign_gpu.py"), os.path.join(HERE, "test_rescue_gpu.py"),
                        os.path.join(HERE, "test_concurrency_gpu.py"), os.path.join(HERE, "test_host_path_gpu.py"), "-k", "not switched_off and not large_batch"],   # (test_large_batch: 11 s of oracle time per child, and nothing a switch changes)
                       env=e, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
