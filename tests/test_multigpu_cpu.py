"""The N>1 path of bench.py on CPU: two processes, gloo backend.  The hot path has no data-path collective
(Spark partition -> device), so what multi-rank runs rely on is (1) every rank deriving a disjoint,
deterministic shard of the synthetic pairs from its rank, and (2) the barrier / MAX-over-ranks timing protocol and
the whole-job aggregation.  Both are exercised here without a GPU; the kernels themselves are covered by -m gpu."""
import os
import socket
import sys

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


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd"))
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from bpsw_hip import synth, wire_pack

    # (1) rank-dependent shard: same generator, seed offset by rank (bench.build_inputs); small sizes here
    seed0 = synth.CONFIG_SEED_BASE + 3 + 1000 * rank
    soa = synth.ext_tasks(256, read_len=bench.READ_LEN, seed=seed0)
    wire = wire_pack(soa)
    digest = torch.tensor([int(np.frombuffer(wire.tobytes(), np.uint8).astype(np.int64).sum()), soa.n], dtype=torch.int64)
    gathered = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, digest)

    # (2) timing protocol of bench.py: barrier, local elapsed, MAX over ranks, whole-job aggregate
    dist.barrier()
    elapsed = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)   # rank 1 is the slow one
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    reads_total = 2 * bench.PAIRS_PER_STEP * 3 * world
    value = reads_total / float(elapsed.item())
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([g.tolist() for g in gathered] + [[value, elapsed.item()]], dtype=np.float64))
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing_protocol(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0, r1)                      # every rank sees the same gathered view and the same aggregate
    assert r0[0, 0] != r0[1, 0]                        # shards differ between ranks
    import bench
    assert abs(r0[2, 1] - 0.020) < 1e-12               # MAX over ranks
    assert abs(r0[2, 0] - 2 * bench.PAIRS_PER_STEP * 3 * 2 / 0.020) < 1e-3   # whole-job reads / max time


def test_rank_shards_are_deterministic():
    sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd"))
    from bpsw_hip import synth, wire_pack
    a = wire_pack(synth.ext_tasks(128, seed=synth.CONFIG_SEED_BASE + 3 + 1000))
    b = wire_pack(synth.ext_tasks(128, seed=synth.CONFIG_SEED_BASE + 3 + 1000))
    c = wire_pack(synth.ext_tasks(128, seed=synth.CONFIG_SEED_BASE + 3))
    assert np.array_equal(a, b) and not np.array_equal(a[: min(a.size, c.size)], c[: min(a.size, c.size)])
