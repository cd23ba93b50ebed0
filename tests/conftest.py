import os
import subprocess
import sys

import numpy as np
import pytest

# the library's stream pool is clamped to the process's HIP hardware queues (default 4): the multi-threaded tests want what an
# executor is told to set; must be in the environment before the HIP runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
# The suite runs with the LIBRARY'S OWN DEFAULTS -- what an executor gets: the sift kernel (csrc/bpsw_extend_sift.hip) in front of
# batches of 8 192 and more tasks only, a lone caller's sizeable SW batch on a launch of its own (BPSW_RING_LONE_LAUNCH,
# csrc/bpsw_sw_runtime.cpp), small extension batches through the extension ring.  (Rounds 3-5 ran it the other way round: the suite
# pinned BPSW_EXT_SIFT_MIN=0 and BPSW_RING_LONE_LAUNCH=0 and a slice of it ran once more at the defaults.)  The FORCED pass --
# tests/test_forced_paths_gpu.py: a child process with BPSW_TEST_FORCED_PATHS=1 -- puts the sift kernel in front of every batch and
# every SW batch on the ring, so that the small batches of the parity tests meet those paths too.
FORCED_PATHS = os.environ.get("BPSW_TEST_FORCED_PATHS") == "1"
if FORCED_PATHS:
    os.environ.setdefault("BPSW_EXT_SIFT_MIN", "0")
    os.environ.setdefault("BPSW_RING_LONE_LAUNCH", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cloud-scale-bwamem_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The driver runs the GPU suite with -x: whatever comes first is what a failure cannot hide.  So the suite is ordered by evidence value,
# not by file name: kernels against the reference's own outputs (tests/golden) and against the oracle first, then the boundary (JNI
# shim, host path, coordinates, large genome), then the policy of the submission rings and the multi-threaded / multi-process tests, and
# last the production-defaults re-runs, the full-size property runs and the soaks.  Files not listed keep their place between the
# boundary and the policy tier.
_ORDER = [
    # tier 1: parity of every kernel
    "test_golden_gpu", "test_extend_gpu", "test_swalign_gpu", "test_rescue_gpu", "test_global_gpu", "test_tail_gpu",
    "test_chain2aln_gpu", "test_ref_gpu", "test_extend_coords_gpu", "test_extend_exhaustive_gpu",
    # tier 2: the boundary
    "test_jni_shim", "test_host_path_gpu", "test_large_genome_gpu", "test_async_entries_gpu",
    # tier 3: ring policy, threads, processes
    None,
    "test_concurrency_gpu", "test_ring_gpu", "test_ext_ring_gpu", "test_two_processes_gpu", "test_bench_ranks_gpu",
    # tier 4: re-runs at other defaults, full-size properties, soaks
    "test_forced_paths_gpu", "test_scale_properties_gpu", "test_soak_gpu",
]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_ORDER) if name}
    unlisted = _ORDER.index(None)

    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(mod, unlisted)

    items.sort(key=key)   # stable: the order inside a file stays


def _have(path):
    return os.path.exists(path)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build what can be built here: the oracle always, oracle/_ref and the HIP library when missing."""
    import pyoracle
    if not _have(pyoracle.ORACLE_SO):
        pyoracle.build(ref=False)
    if not _have(pyoracle.REF_SO) and os.path.isdir("/root/reference/src/main/native"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"], check=False)
    import bpsw_hip
    if not (_have(bpsw_hip.LIB_PATH) and _have(bpsw_hip.SYNTH_PATH)):
        subprocess.run(["make", "-C", PKG, "-s", "-j4"], check=True)


@pytest.fixture(scope="session")
def orc():
    import pyoracle
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def ref():
    import pyoracle
    if not pyoracle.Ref.available():
        pytest.skip("oracle/_ref/libbwaref.so not built (reference tree absent)")
    return pyoracle.Ref()


@pytest.fixture(scope="session")
def ctx():
    """A device context; only -m gpu tests ask for it.  No fallback: failure to create one is an error."""
    import bpsw_hip
    c = bpsw_hip.Context(0)
    yield c
    c.close()


def region_fields_equal(a: np.ndarray, b: np.ndarray, skip=()):
    assert a.shape == b.shape, (a.shape, b.shape)
    for f in a.dtype.names:
        if f in skip:
            continue
        assert np.array_equal(a[f], b[f]), f"field {f}: {int((a[f] != b[f]).sum())} of {a.shape[0]} differ"
