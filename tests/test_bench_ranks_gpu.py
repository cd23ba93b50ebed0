"""bench.py's N-rank path rehearsed on a one-GPU box: `python bench.py --gpus 2` -- no launcher around it -- starts two ranks itself;
BENCH_FORCE_DEVICE=0 puts both on device 0 and BENCH_BACKEND=gloo stands in for RCCL (two ranks cannot share one GPU under RCCL).  Each
rank streams its own shard through both boundaries with the real feeder and the real kernels (a sixteenth of the workload:
BENCH_SHRINK); the one JSON line says two ranks took part, and its value is the two shards over the slower rank's time
(the reference's counterpart is mapPartitions over Spark partitions, FastMap.scala:266-293)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_on_one_device():
    env = dict(os.environ, BENCH_FORCE_DEVICE="0", BENCH_BACKEND="gloo", BENCH_SHRINK="16", BENCH_THREADS="8", GPU_MAX_HW_QUEUES="10")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_STUB"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["shrink"] == 16
    assert d["verified"] and d["value"] > 0
    # whole-job aggregate: both ranks' reads over the slower rank's time
    assert abs(d["value"] - 2 * d["config"]["reads_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
