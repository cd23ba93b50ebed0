"""The quad kernel (bpsw_extend_quad.hip: four extension flanks per wavefront, opt-in with BPSW_EXT_QUAD=1) against the oracle:
the extension parity tests, the golden vectors and the exhaustive short-flank test are run again, in a child process, with the
hand-over switched on -- the DP of every flank the exact shortcuts refuse (or, with the shortcuts off, of every flank) then goes
through that kernel."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_parity_suite_with_the_quad_kernel():
    if os.environ.get("BPSW_EXT_QUAD") == "1":
        pytest.skip("already running with BPSW_EXT_QUAD=1")
    env = dict(os.environ, BPSW_EXT_QUAD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(HERE, "test_extend_gpu.py"), os.path.join(HERE, "test_golden_gpu.py"),
                        os.path.join(HERE, "test_extend_coords_gpu.py"), os.path.join(HERE, "test_extend_exhaustive_gpu.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
