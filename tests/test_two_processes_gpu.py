"""Two executor PROCESSES on one GPU (Spark runs several executors per node, run_test.sh:1-7): each with a 12-thread feeder over
both boundaries, every output of both equal to the oracle while they share device 0.  Each process gets half of the hardware
queues the device tolerates (GPU_MAX_HW_QUEUES=10, and the library clamps its stream pool to that: DESIGN.md section 6)."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_pair(queues):
    env = dict(os.environ, GPU_MAX_HW_QUEUES=str(queues))
    env.pop("BPSW_STREAM_POOL", None)
    go_at = time.time() + 25.0     # both are past their imports and warm-up by then (a fresh box pages torch in for a while)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "two_proc_worker.py"), str(k), repr(go_at)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(2)]
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads(so.strip().split("\n")[-1]))
    return outs


def test_two_executor_processes_share_device_0():
    outs = _run_pair(10)
    assert [o["bad"] for o in outs] == [0, 0], outs
    print("two processes on device 0:", outs)
