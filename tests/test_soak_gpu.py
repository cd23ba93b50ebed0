"""Half-minute slices of the adversarial soak tools, so that the driver's GPU run witnesses them (they used to live only in
tools/ and profiles/*.txt): tools/soak_cert2.py aims at the extension kernel's exact shortcuts (flanks whose deficit sits at the
boundaries of the closed forms, in low-complexity and periodic sequence, six gap-cost / band settings, both z-drop parses);
tools/soak_sw.py at the rescue SW kernels (mates of 1..256 bases, every columns-per-lane variant of the packed kernel, repeats,
N, decoys, six scorings x six flag sets); tools/soak_long.py at long extension flanks.  Every result is compared with the oracle's full DP."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(TOOLS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_soak_slice_extension_shortcuts():
    lines = []
    total, bad = _load("soak_cert2").run(rounds=1000, per=4000, time_limit=25, log=lines.append)
    assert bad == 0, "\n".join(lines[-5:])
    assert total >= 4000      # at least one whole round (how many more is the box's speed, not the test's business)


def test_soak_slice_long_flanks_sliding_window():
    """tools/soak_long.py: flanks of 64..255 bases (the sliding-window sweep of csrc/bpsw_extend_core.h, its hand-over from the slot
    sweep, its window moves and its overflow fallback), wide bands, the doubled band of the retry"""
    lines = []
    total, bad = _load("soak_long").run(rounds=1000, per=1000, time_limit=20, log=lines.append)
    assert bad == 0, "\n".join(lines[-5:])
    assert total >= 1000


def test_soak_slice_rescue_sw():
    lines = []
    total, bad = _load("soak_sw").run(rounds=1000, per=1500, time_limit=25, log=lines.append)
    assert bad == 0, "\n".join(lines[-5:])
    assert total >= 1500


def test_no_ring_integrity_fault_in_the_whole_run(ctx):
    """the last test of the suite (tests/conftest.py orders the files): the process-wide count of ring records that were not in host
    memory when their batch's completion word was (csrc/bpsw_ring.cpp, bpsw_ring_integrity) -- over every ring batch of this process"""
    on, checked, faults = ctx.ring_integrity()
    assert on and faults == 0, (on, checked, faults)
