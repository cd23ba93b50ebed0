"""The extension ring (csrc/bpsw_ring.h, class RING_CLASS_EXT, round 5): SMALL extension batches -- no flank above 255 bases, too few
tasks for the sift kernel -- are descriptors of one resident kernel per device instead of a launch each.  What the reference has to
compare with is the call's contract (one blocking call per batch: jni_fpga/sw_extend_fpga.c:116-193) and the sizes it sends
(-FPGASWExtThreshold 64: run_test.sh:7; the later rounds of memChainToAlnBatched, worker1/MemChainToAlignBatched.scala:471-615, send
a few dozen tasks): the results must not depend on the path.  The switches are read once per process, so these tests run children: at the library's own
defaults (also when this process is the forced-paths pass, tests/test_forced_paths_gpu.py, whose BPSW_EXT_SIFT_MIN=0 keeps every batch
OFF this ring), with the ring switched off, with a tiny ring, with a ring that cannot launch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys, threading
import numpy as np
sys.path.insert(0, {pkg!r}); sys.path.insert(0, {orc!r}); sys.path.insert(0, {tests!r})
import bpsw_hip
from bpsw_hip import synth
import pyoracle as po
orc = po.Oracle()
ring_expected = os.environ.get("BPSW_EXT_RING", "1") != "0" and os.environ.get("BPSW_RING", "1") != "0" and not os.environ.get("BPSW_RING_TEST_FAIL_LAUNCH")
ctx = bpsw_hip.Context(0)
cases = []
for k, (n, rl) in enumerate(((1, 150), (7, 150), (61, 150), (64, 100), (253, 150), (1019, 150), (3000, 150), (200, 250), (2500, 250))):
    soa = synth.ext_tasks(n + 8, read_len=rl, seed=7100 + k, sub_rate=0.06 if rl == 250 else 0.01, indel_rate=0.015 if rl == 250 else 0.001)
    soa = soa.subset(np.arange(min(n, soa.n)))     # exactly n tasks
    wire = bpsw_hip.wire_pack(soa)
    want, _ = orc.wire_extend(wire)
    cases.append((wire, np.asarray(want).reshape(-1)))
before = ctx.stats().ext_ring_calls
for wire, want in cases:
    for _ in range(2):   # the context's staging buffers are reused: the second call reads what the first one's batch left behind them
        got = np.asarray(ctx.extend_batch(wire)).reshape(-1)
        assert np.array_equal(got, want), ("lone call", wire.size)
taken = ctx.stats().ext_ring_calls - before
assert (taken > 0) == ring_expected and taken <= 2 * len(cases), (taken, ring_expected)   # (a batch without tasks, or with a target flank beyond the resident kernel's LDS, is not the ring's)
# the reference's own symbol, swExtendFPGAJNI, through the fake JNIEnv: its stage / commit pair takes the same path (the shim's
# per-thread context: its handle from bpsw_jni_thread_info, its statistics through the C ABI)
import ctypes as C
from bpsw_hip import jnishim
fake, lib = jnishim.load_fake()
wire_j, want_j = cases[2]
n_j = int(np.frombuffer(wire_j[8:12].tobytes(), "<i4")[0])
for _ in range(2):
    rc, out_j, msg = jnishim.extend(fake, wire_j, n_j)
    assert rc == 0, msg
    assert np.array_equal(np.asarray(out_j).reshape(-1), want_j), "swExtendFPGAJNI"
lib.bpsw_jni_thread_info.restype = C.c_uint64
three = (C.c_int32 * 3)()
h = lib.bpsw_jni_thread_info(three)
st = bpsw_hip.Stats()
assert h and lib.bpsw_get_stats(C.c_void_p(h), C.byref(st)) == 0
assert (st.ext_ring_calls >= 2) == ring_expected, (st.ext_ring_calls, ring_expected)
# coordinate batches (wire format 2: the target flanks come from the reference on the device) are descriptors like the others
pac_c, bases_c, off_c, ln_c, names_c, dups_c = synth.contig_reference([60_000, 40_000], seed=7400)
l_pac_c = int(off_c[-1] + ln_c[-1])
ctx.ref_load(pac_c, l_pac_c)
bases_fwd = np.asarray(bases_c[:l_pac_c], np.uint8)
r_before = ctx.stats().ext_ring_calls
for k, n_reads in enumerate((20, 64, 200)):
    chains = synth.read_chains(n_reads, bases_fwd, l_pac_c, read_len=150, seed=7410 + k)
    co, by = synth.coord_ext_tasks(chains, bases_fwd, seed=7420 + k)
    want_c = np.asarray(orc.wire_extend(bpsw_hip.wire_pack(by))[0]).reshape(-1)
    wire_c = bpsw_hip.wire_coords_pack(co)
    for _ in range(2):
        assert np.array_equal(np.asarray(ctx.extend_batch(wire_c)).reshape(-1), want_c), ("coordinate batch", n_reads)
assert (ctx.stats().ext_ring_calls - r_before > 0) == ring_expected
ctx.ref_unload()
# a custom matrix and other gap costs ride in the descriptor
soa = synth.ext_tasks(300, read_len=150, seed=7200)
for zmode in (po.ZDROP_SCALA, po.ZDROP_BWA):
    mat = po.default_mat().copy().reshape(5, 5)
    mat[0, 1] = mat[1, 0] = -2
    ctx.set_ext_scoring(mat.reshape(-1), zdrop=60, zdrop_mode=zmode)
    soa.o_del, soa.e_del, soa.o_ins, soa.e_ins = 5, 2, 7, 1
    wire = bpsw_hip.wire_pack(soa)
    want, _ = orc.wire_extend(wire, mat=mat.reshape(-1), zdrop=60, zdrop_mode=zmode)
    assert np.array_equal(np.asarray(ctx.extend_batch(wire)).reshape(-1), np.asarray(want).reshape(-1)), ("scoring", zmode)
# sixteen task threads of small calls, and rescue groups through the OTHER ring meanwhile
errors = []
groups = [synth.rescue_group(40, seed=7300 + j, p_resc=0.4) for j in range(3)]
gwant = [orc.matesw_group(orc.default_opt(), g, po.RESCUE_C)[:2] for g in groups]
def worker(t):
    try:
        c = bpsw_hip.Context(0)
        for it in range(30):
            wire, want = cases[(t + it) % len(cases)]
            assert np.array_equal(np.asarray(c.extend_batch(wire)).reshape(-1), want), ("thread", t, it)
            if it % 5 == 0:
                cnt, regs = c.matesw_group(bpsw_hip.default_opt(), groups[t % 3])
                assert np.array_equal(cnt, gwant[t % 3][0]) and regs.tobytes() == gwant[t % 3][1].tobytes(), ("group", t, it)
        c.close()
    except BaseException as e:
        errors.append(repr(e))
ts = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
[t.start() for t in ts]; [t.join(600) for t in ts]
assert not any(t.is_alive() for t in ts), "a caller is still waiting"
assert not errors, errors[:2]
e, s, carried = ctx.ring_stats()
on, checked, faults = ctx.ring_integrity()   # the tripwire of csrc/bpsw_ring.cpp: every ring record poisoned before, looked at after
assert on and faults == 0 and (checked > 0 or not ring_expected), (on, checked, faults)   # (the rescue groups' ring counts too: `checked` can be > 0 with the extension ring off)
print("EXTRING", taken, e, s, carried)
ctx.close()
"""


def _run(extra_env):
    env = dict(os.environ, **extra_env)
    for k in ("BPSW_TEST_FORCED_PATHS", "BPSW_EXT_SIFT_MIN", "BPSW_RING_LONE_LAUNCH"):   # the library's own defaults, whatever this process runs with
        env.pop(k, None)
    src = _CHILD.format(pkg=os.path.join(ROOT, "cloud-scale-bwamem_amd"), orc=os.path.join(ROOT, "oracle"), tests=os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", src], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2500:]
    return [ln for ln in r.stdout.splitlines() if ln.startswith("EXTRING")][0].split() + [r.stderr]


def test_small_extension_batches_through_the_ring_are_bit_exact():
    """1 ... 3 000 tasks per call at 100 / 150 / 250 bp, twice each on one context; a custom matrix and gap costs; sixteen threads of
    small calls with rescue groups through the rescue ring in between -- every record equal to the oracle's, every eligible call taken
    by the ring."""
    if os.environ.get("BPSW_RING", "1") == "0":
        pytest.skip("BPSW_RING=0")
    line = _run({})
    assert int(line[1]) >= 10      # of the eighteen lone calls: those of up to 256 tasks (BPSW_EXT_RING_MAX_TASKS)


def test_the_same_calls_with_the_extension_ring_switched_off():
    line = _run({"BPSW_EXT_RING": "0"})
    assert int(line[1]) == 0


def test_a_tiny_extension_ring_rolls_over():
    """64 descriptors per epoch: the sixteen threads' few hundred calls cross several epoch boundaries; every size up to 8 192 tasks through
    the ring, and nothing zero-copy (the batches are copied into device memory first)"""
    if os.environ.get("BPSW_RING", "1") == "0":
        pytest.skip("BPSW_RING=0")
    line = _run({"BPSW_RING_CAPACITY": "64", "BPSW_EXT_RING_MAX_TASKS": "8192", "BPSW_EXT_RING_ZC_BYTES": "0"})
    assert int(line[1]) == 18
    assert int(line[2]) >= 5   # epochs (both rings)


def test_an_extension_ring_that_cannot_launch_sends_its_callers_back_to_launches():
    """BPSW_RING_TEST_FAIL_LAUNCH=1: the first epoch launch of every ring of the device "fails".  The call that met the failure and every
    later one take kernel launches of their own -- the same records, one line on stderr per ring, no error to any caller."""
    if os.environ.get("BPSW_RING", "1") == "0":
        pytest.skip("BPSW_RING=0")
    line = _run({"BPSW_RING_TEST_FAIL_LAUNCH": "1"})
    assert int(line[1]) == 0
    assert "the extension ring of device 0 failed" in line[-1] and "the submission ring of device 0 failed" in line[-1]
