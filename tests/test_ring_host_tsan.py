"""The host half of the submission ring under sanitizers, on the CPU box (tests/ring_host/): csrc/bpsw_ring.cpp is compiled UNCHANGED by
g++ with -fsanitize=thread (and address,undefined) against C++ threads that play the resident kernel -- a poller and workers that follow
csrc/bpsw_ring_dev.h step by step, with the hardware's orderings written as the C++ orderings they amount to (the visibility argument of
DESIGN.md 4.2a as checkable code).  Eight caller threads on three ring classes, a 64-descriptor ring that idles out after 0.3 ms, a thread
that pauses and resumes the rings every 3 ms (bpsw_ref_load's path): every call must return with its own results -- none lost across
idle closes, used-up rings, carried descriptors and pauses --, the integrity tripwire (ring_poison / ring_check) must count no fault, and
the sanitizer must have nothing to report.  The reference has no such structure to compare with (its boundary is one synchronous C
call per group, native/jni_mate_sw.c:534); what is pinned here is the contract: a call returns exactly its own batch's results."""
import os
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ring_host")


def _run(san, threads, calls, workers, extra_env=None):
    tag = san.replace(",", "_")
    r = subprocess.run(["make", "-C", HERE, "-s", f"SAN={san}"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1")
    for k in ("BPSW_RING_CAPACITY", "BPSW_RING_IDLE_US", "BPSW_RING_INTEGRITY", "BPSW_RING"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([os.path.join(HERE, "_build", f"ring_host_{tag}"), str(threads), str(calls), str(workers)], env=env,
                       capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "WARNING: ThreadSanitizer" not in out and "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RINGHOST")][0].split()
    f = dict(zip(line[1::2], line[2::2]))
    assert int(f["calls"]) == threads * calls
    assert int(f["submitted"]) == threads * calls or (extra_env and "BPSW_RING_TEST_FAIL_CARRY_LAUNCH" in extra_env)
    assert int(f["wrong"]) == 0 and int(f["integrity_faults"]) == 0 and int(f["failures"]) == 0
    if not (extra_env and "BPSW_RING_TEST_FAIL_CARRY_LAUNCH" in extra_env):
        assert int(f["integrity_checked"]) == int(f["units"])
        assert int(f["epochs"]) >= int(f["calls"]) // 64          # a 64-descriptor ring: at least that many roll-overs
    return f


@pytest.mark.parametrize("threads,calls,workers", [(8, 300, 6), (3, 500, 2), (16, 120, 3)])
def test_ring_host_under_thread_sanitizer(threads, calls, workers):
    _run("thread", threads, calls, workers)


def test_ring_host_under_address_and_ub_sanitizers():
    _run("address,undefined", 8, 300, 6)


def test_descriptors_a_closing_epoch_did_not_consume_are_carried_over():
    """a poller that is slow to pick descriptors up (FAKE_POLLER_LAG_US): the pauses every 3 ms find published descriptors unconsumed, the
    next epoch carries them -- hundreds of times in a run --, and every call still returns its own results"""
    f = _run("thread", 16, 200, 3, {"FAKE_POLLER_LAG_US": "300"})
    assert int(f["carried"]) > 0 and int(f["relaunched"]) == 0 and int(f["fallback"]) == 0


def test_a_failed_epoch_launch_sends_the_waiters_of_carried_descriptors_to_launches_of_their_own():
    """BPSW_RING_TEST_FAIL_CARRY_LAUNCH=1: the first epoch launch that carries descriptors over fails.  The thread in start_epoch falls back
    to a launch of its own (round 5 did that much); the OTHER threads, whose descriptors were among the carried ones and which are waiting
    for them, used to get BPSW_ERR_DEVICE -- now ring_wait tells them BPSW_RING_RELAUNCH (nothing of their batch has reached a kernel, none
    exists) and they take a launch too (advisor, round 5).  Every call of every thread completes with its own results; afterwards every
    caller goes straight to launches (ring_usable)."""
    f = _run("thread", 16, 200, 3, {"FAKE_POLLER_LAG_US": "300", "BPSW_RING_TEST_FAIL_CARRY_LAUNCH": "1"})
    assert int(f["relaunched"]) > 0 and int(f["relaunched"]) == int(f["carried"]), f
    assert int(f["fallback"]) > 0 and int(f["fallback"]) + int(f["relaunched"]) + int(f["submitted"]) - int(f["carried"]) >= int(f["calls"]) - int(f["carried"]), f
