"""The library's HOST layer under sanitizers, on the CPU box (round-5 review, item 6: "no sanitizer has ever run" on ~7 k lines of
threaded host C++).  tests/host_san builds csrc/bpsw_runtime.cpp, bpsw_sw_runtime.cpp, bpsw_ring.cpp, bpsw_rescue.cpp, bpsw_pack.cpp,
bpsw_tail.cpp, bpsw_tail_pool.cpp and bpsw_jni.cpp UNCHANGED with g++ -fsanitize=address,undefined (and =thread) over a fake device:
the HIP entry points over ordinary memory, the kernels played by the oracle, a ring's resident kernel played by C++ threads that follow
csrc/bpsw_ring_dev.h (tests/host_san/fake_device.cpp).  Then the ordinary parity tests of the boundary run against THAT build in a child
process (BPSW_LIB, the sanitizer runtime preloaded).  The results are the oracle's by construction; what is checked is everything around
the kernels -- the JNI marshalling (eager, lazy and flat entries through tests/fake_jvm), the rescue planner's speculation and replay, the
packers, the staging arithmetic, both rings' host halves, the integrity tripwire -- for out-of-bounds accesses, use after free, undefined
behaviour (address,undefined) and data races (thread: 32 caller threads of mixed batch sizes, ring roll-overs, a ring that fails to
launch; worker2's tail with its pool of tail workers and the concurrent-contexts test: memRegToAln, the chain round loop and the reference
fetch are played by the oracle too).  Left out because the fake device does not play them: the extension's coordinate batches, SWGlobal,
the sift kernel's classification, the device-resident entries.  First finding (fixed): `&mates->rb` on a null `mates` in bpsw_rescue.cpp."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SAN_DIR = os.path.join(HERE, "host_san")
NOT_PLAYED = ("not switched_off and not large_batch and not classify and not idle_epoch and not device_resident "
              "and not sift_kernel_changes and not unexpected_band and not lone_caller and not scale_properties")


def _gcc_file(name):
    r = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


def _run(san, runtime_lib, files, k, timeout=900):
    rt, stdcpp = _gcc_file(runtime_lib), _gcc_file("libstdc++.so")
    if not rt or not stdcpp:
        pytest.skip(f"{runtime_lib} is not installed with this gcc")
    tag = san.replace(",", "_")
    r = subprocess.run(["make", "-C", SAN_DIR, "-s", "-j8", f"SAN={san}"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lib = os.path.join(SAN_DIR, "_build", f"libbPSW_hostsan_{tag}.so")
    # (libstdc++ rides along: the interpreter does not link it, and the sanitizer runtime resolves __cxa_throw when IT is loaded)
    env = dict(os.environ, LD_PRELOAD=f"{rt}:{stdcpp}", BPSW_LIB=lib, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS=f"halt_on_error=0 ignore_noninstrumented_modules=1 second_deadlock_stack=1 report_thread_leaks=0 log_path={os.path.join(SAN_DIR, '_build', 'tsan_report')}", BPSW_TEST_THREAD_ITERS="4")
    for v in ("BPSW_TEST_FORCED_PATHS", "BPSW_EXT_SIFT_MIN", "BPSW_RING_LONE_LAUNCH", "BPSW_RING"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k", k] + [os.path.join(HERE, f) for f in files],
                       env=env, capture_output=True, text=True, timeout=timeout)
    out = r.stdout + r.stderr
    import glob
    for rep in glob.glob(os.path.join(SAN_DIR, "_build", "tsan_report*")):   # (the child's pytest captures the tests' stderr: TSan writes its reports to files)
        out += open(rep).read()
        os.remove(rep)
    for mark in ("runtime error:", "ERROR: AddressSanitizer", "WARNING: ThreadSanitizer", "AddressSanitizer CHECK failed"):
        assert mark not in out, out[max(0, out.index(mark) - 200):][:6000]
    assert r.returncode == 0, out[-4000:]
    assert " passed" in r.stdout and " failed" not in r.stdout, r.stdout[-1500:]
    return r.stdout


def test_host_layer_under_address_and_ub_sanitizers():
    out = _run("address,undefined", "libasan.so",
               ["test_rescue_gpu.py", "test_jni_shim.py", "test_swalign_gpu.py", "test_ring_gpu.py", "test_host_path_gpu.py", "test_extend_gpu.py",
                "test_tail_gpu.py", "test_chain2aln_gpu.py", "test_concurrency_gpu.py", "test_ref_gpu.py"],
               NOT_PLAYED + " and not 32_threads")
    assert int(out.strip().splitlines()[-1].split()[0]) >= 120    # (tests that ran against the sanitizer build)


def test_host_layer_under_thread_sanitizer():
    # (no test that starts a child process: a fork from a multi-threaded process under the thread sanitizer does not come back)
    _run("thread", "libtsan.so", ["test_ring_gpu.py", "test_rescue_gpu.py", "test_tail_gpu.py", "test_concurrency_gpu.py"],
         "32_threads or batches_go or (group_rescue and not 250bp) or tail_pool or concurrent", timeout=800)
