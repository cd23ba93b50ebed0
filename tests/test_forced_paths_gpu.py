"""The parity tests once more with the paths the library's defaults reserve for large batches and busy devices FORCED on every batch:
BPSW_EXT_SIFT_MIN=0 (the sift kernel, csrc/bpsw_extend_sift.hip, in front of every extension batch instead of those of 8 192 tasks and
more) and BPSW_RING_LONE_LAUNCH=0 (every SW batch through the submission ring, also a lone caller's sizeable one).  The suite itself
runs at the library's own defaults (tests/conftest.py); the switches are read once per process, hence the child processes, each run to
its end before this one continues.  (The `switched_off` re-runs of test_host_path_gpu.py are children of their own and are not nested
here: the whole suite must stay well inside the driver's limit.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _forced(files, extra_k=""):
    if os.environ.get("BPSW_TEST_FORCED_PATHS") == "1":
        pytest.skip("already running with the forced paths")
    env = dict(os.environ, BPSW_TEST_FORCED_PATHS="1")
    k = "not switched_off" + (" and " + extra_k if extra_k else "")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k", k] + [os.path.join(HERE, f) for f in files],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and " failed" not in r.stdout, r.stdout[-500:]


def test_extension_suites_with_the_sift_kernel_on_every_batch():
    _forced(["test_golden_gpu.py", "test_extend_gpu.py", "test_extend_coords_gpu.py", "test_chain2aln_gpu.py"])


def test_boundary_1_suites_with_every_sw_batch_on_the_ring():
    _forced(["test_swalign_gpu.py", "test_rescue_gpu.py", "test_tail_gpu.py", "test_ring_gpu.py", "test_jni_shim.py", "test_concurrency_gpu.py",
             "test_host_path_gpu.py"], extra_k="not large_batch")
