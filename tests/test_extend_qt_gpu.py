"""The opt-in quad-task extension kernels (BPSW_EXT_QT=1, csrc/bpsw_extend_qt.hip) must stay bit-exact too: the
extension and golden parity tests are re-run in a subprocess with the switch on (the switch is read once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
# the experimental kernels are not in the default library: `make EXPERIMENTAL=1` / tools/build_variant.sh exp -DBPSW_EXPERIMENTAL_KERNELS
# (__graft_entry__.build() builds the variant) puts them in lib_exp/libbPSW_hip_exp.so
EXP_LIB = os.path.join(os.path.dirname(HERE), "cloud-scale-bwamem_amd", "lib_exp", "libbPSW_hip_exp.so")


def test_extension_parity_with_quad_task_kernels():
    if os.environ.get("BPSW_EXT_QT") == "1":
        pytest.skip("already running with BPSW_EXT_QT=1")
    if not os.path.exists(EXP_LIB):
        pytest.skip("experimental-kernel build of the library not present")
    env = dict(os.environ, BPSW_LIB=EXP_LIB, BPSW_EXT_QT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(HERE, "test_extend_gpu.py"),
                        os.path.join(HERE, "test_golden_gpu.py"), os.path.join(HERE, "test_jni_shim.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
