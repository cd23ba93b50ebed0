"""The N>1 path of bench.py on CPU: two processes, gloo backend.  The hot path has no data-path collective
(Spark partition -> device), so what multi-rank runs rely on is (1) every rank deriving a disjoint,
deterministic shard of the synthetic pairs from its rank -- bench.build_inputs itself, on a cut-down workload -- and
(2) bench.py's own barrier / MAX-over-ranks / ranks-seen reduction and whole-job aggregation (bench.reduce_over_ranks,
bench.whole_job_rate) around a really timed region.  No GPU is needed; the kernels are covered by -m gpu, and the
partition -> context -> device mapping of the JNI shim by tests/test_jni_shim.py."""
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _small_workload(bench):
    W = dict(bench.WORKLOADS[3])
    W["ext_batches"], W["groups"] = 2, 2
    bench.READS_PER_EXT_BATCH, bench.PAIRS_PER_GROUP = 512, 64   # the generators take the sizes from the module
    return W


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd"))
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    W = _small_workload(bench)

    # (1) this rank's shard, by bench.py's own generator
    wires, ntasks, groups = bench.build_inputs(W, 3, rank, workers=2)
    digest = torch.tensor([int(sum(int(w.astype(np.int64).sum()) for w in wires)), int(sum(ntasks)),
                           int(sum(int(g.seq_pool.astype(np.int64).sum()) for g in groups))], dtype=torch.int64)
    gathered = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, digest)

    # (2) bench.py's timing protocol around a timed region whose length depends on the rank (rank 1 is the slow one)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))
    dist.barrier()                      # as in bench.py: the timed region ends with a barrier, so every rank waits for the slowest
    elapsed = time.perf_counter() - t0
    elapsed_max, ranks_seen = bench.reduce_over_ranks(elapsed, "cpu")
    reads_per_step = bench.READS_PER_EXT_BATCH * W["ext_batches"]
    value = bench.whole_job_rate(reads_per_step, 3, world, elapsed_max)
    np.save(os.path.join(out_dir, f"r{rank}.npy"),
            np.array([g.tolist() for g in gathered] + [[value, elapsed_max, ranks_seen], [elapsed, reads_per_step, 0]], dtype=np.float64))
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing_protocol(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0[:3], r1[:3])              # every rank sees the same gathered view and the same aggregate
    assert r0[0, 0] != r0[1, 0] and r0[0, 2] != r0[1, 2]   # the ranks' wire batches and rescue groups differ
    value, elapsed_max, ranks_seen = r0[2]
    assert ranks_seen == 2
    assert elapsed_max >= max(r0[3, 0], r1[3, 0]) - 1e-9 and elapsed_max >= 0.1   # MAX over ranks, and it covers the slow rank's sleep
    assert abs(value - r0[3, 1] * 3 * 2 / elapsed_max) < 1e-6 * value       # whole-job reads / max time


def test_rank_shards_are_deterministic():
    sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd"))
    sys.path.insert(0, ROOT)
    import bench
    saved = bench.READS_PER_EXT_BATCH, bench.PAIRS_PER_GROUP
    try:
        W = _small_workload(bench)
        a = bench.build_inputs(W, 3, 1, workers=2)
        b = bench.build_inputs(W, 3, 1, workers=2)
        c = bench.build_inputs(W, 3, 0, workers=2)
    finally:
        bench.READS_PER_EXT_BATCH, bench.PAIRS_PER_GROUP = saved
    assert all(np.array_equal(x, y) for x, y in zip(a[0], b[0])) and a[1] == b[1]
    assert all(np.array_equal(x.seq_pool, y.seq_pool) and np.array_equal(x.regs, y.regs) for x, y in zip(a[2], b[2]))
    assert not all(x.size == y.size and np.array_equal(x, y) for x, y in zip(a[0], c[0]))


def _run_bench(extra_args, env_extra, timeout=300):
    import json
    import subprocess
    env = dict(os.environ, BENCH_STUB="1", BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's N = 1 command): bench.py starts the two ranks
    (torch.distributed.run on 127.0.0.1) before it touches any GPU, rank 0 prints the one line, and the line says two ranks took part.
    BENCH_STUB=1: the launch / rendezvous / reduction skeleton with a sleep for a workload (no GPU here)."""
    r, line = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    assert line is not None and line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["stub"] is True
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1          # ONE line: rank 0's
    assert line["ms_per_step"] >= 2 * 20.0 - 1.0                                          # MAX over ranks: rank 1 sleeps twice as long


def test_bench_refuses_a_world_that_is_not_gpus():
    """under a launcher WORLD_SIZE must equal --gpus: a run that was started as one rank cannot print n_gpus: 2"""
    r, line = _run_bench(["--gpus", "2", "--steps", "1"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2 and line is None
    assert "WORLD_SIZE=1" in r.stderr


def test_bench_single_rank_stub_line():
    r, line = _run_bench(["--gpus", "1", "--steps", "1"], {})
    assert r.returncode == 0 and line["n_gpus"] == 1 and line["ranks_seen"] == 1
