"""The extension parity tests once more with the library's PRODUCTION defaults: the rest of the suite runs with BPSW_EXT_SIFT_MIN=0
(tests/conftest.py), which puts the sift kernel in front of every batch; here the variable is unset, so batches below 8 192 tasks go
straight to the extension kernel and larger ones take the sift kernel, as in an executor (csrc/bpsw_runtime.cpp).  One child process,
run to its end before this one continues."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_extension_suite_with_production_sift_threshold():
    if os.environ.get("BPSW_TEST_PRODUCTION_DEFAULTS"):
        pytest.skip("already running with the production defaults")
    env = dict(os.environ, BPSW_TEST_PRODUCTION_DEFAULTS="1")
    env.pop("BPSW_EXT_SIFT_MIN", None)
    env.pop("BPSW_RING_LONE_LAUNCH", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(HERE, "test_extend_gpu.py"),
                        os.path.join(HERE, "test_golden_gpu.py"), os.path.join(HERE, "test_extend_coords_gpu.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_boundary_1_suites_with_production_defaults():
    """... and boundary 1, the JNI shim and the multi-threaded tests: with the production defaults a lone caller's sizeable SW
    batch takes a launch of its own (BPSW_RING_LONE_LAUNCH, csrc/bpsw_sw_runtime.cpp; the suite pins the ring), small extension batches
    go through the extension ring, and large ones meet the sift kernel only from 8 192 tasks on.  (tests/test_ring_gpu.py asserts ring
    counters and stays with the pinned settings.)"""
    if os.environ.get("BPSW_TEST_PRODUCTION_DEFAULTS"):
        pytest.skip("already running with the production defaults")
    env = dict(os.environ, BPSW_TEST_PRODUCTION_DEFAULTS="1")
    for k in ("BPSW_EXT_SIFT_MIN", "BPSW_RING_LONE_LAUNCH"):
        env.pop(k, None)
    files = ["test_rescue_gpu.py", "test_host_path_gpu.py", "test_concurrency_gpu.py", "test_jni_shim.py"]   # (the whole suite must stay well inside the driver's 900 s)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu"] + [os.path.join(HERE, f) for f in files],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
