"""The device-resident entry points are asynchronous: table scan, main launch and the scan's read-back are enqueued back to
back.  A malformed batch is reported at the next synchronising call on the context (bpsw_last_kernel_ms or the next call),
a batch that outgrows the speculative launch geometry is launched again there, and results are the ones of the host path."""
import numpy as np
import pytest
import torch

import bpsw_hip
from bpsw_hip import synth
import pyoracle as po

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_extend_device_entry_defers_errors_and_recovers(ctx, orc):
    soa = synth.ext_tasks(2000, read_len=150, seed=77)
    wire = bpsw_hip.wire_pack(soa)
    want = ctx.extend_batch(wire)
    d_out = torch.zeros(10 * soa.n, dtype=torch.int16, device=DEV)
    bad = wire.copy()
    bad[32 + 8: 32 + 12] = np.frombuffer(np.int32(10 ** 8).tobytes(), np.uint8)     # a sequence offset outside the buffer
    d_bad = _dev(bad)
    torch.cuda.synchronize()
    ctx.extend_batch_device(d_bad.data_ptr(), bad.size, soa.n, d_out.data_ptr())    # accepted: nothing has been read back yet
    with pytest.raises(bpsw_hip.BpswError):
        ctx.last_kernel_ms()                                                        # ... and reported here
    assert not d_out.cpu().numpy().any()                                            # the kernel left the batch untouched
    d_wire = _dev(wire)
    torch.cuda.synchronize()
    ctx.extend_batch_device(d_bad.data_ptr(), bad.size, soa.n, d_out.data_ptr())
    with pytest.raises(bpsw_hip.BpswError):                                         # or by the next call on the context
        ctx.extend_batch_device(d_wire.data_ptr(), wire.size, soa.n, d_out.data_ptr())
    ctx.extend_batch_device(d_wire.data_ptr(), wire.size, soa.n, d_out.data_ptr())  # the context is usable again
    ms, _ = ctx.last_kernel_ms()
    assert ms > 0 and np.array_equal(d_out.cpu().numpy(), want)


def test_extend_device_entry_relaunches_batches_beyond_the_async_geometry(ctx, orc):
    """sides of 300..400 bases need the LDS row of sw_extend_wave: more than the fixed geometry of the asynchronous launch"""
    rng = np.random.default_rng(5)
    from test_extend_gpu import _manual_tasks
    tasks = []
    for _ in range(40):
        n = int(rng.integers(300, 400))
        q = rng.integers(0, 4, n)
        r = q.copy()
        for p in rng.integers(0, n, 12):
            r[p] = (r[p] + 1) & 3
        tasks.append((q.tolist(), np.concatenate([r, rng.integers(0, 4, 80)]).tolist(), [], [], 60, n))
    soa = _manual_tasks(tasks)
    wire = bpsw_hip.wire_pack(soa)
    want, _ = orc.wire_extend(wire)
    d_wire, d_out = _dev(wire), torch.zeros(10 * soa.n, dtype=torch.int16, device=DEV)
    torch.cuda.synchronize()
    ctx.extend_batch_device(d_wire.data_ptr(), wire.size, soa.n, d_out.data_ptr())
    ctx.last_kernel_ms()
    assert np.array_equal(d_out.cpu().numpy(), want)


def test_swalign_device_entry_speculates_the_previous_geometry(ctx, orc):
    opt = bpsw_hip.default_opt()
    xtra = bpsw_hip.KSW_XSUBO | bpsw_hip.KSW_XSTART | bpsw_hip.KSW_XBYTE | 19

    def run(jobs):
        d = {k: _dev(v) for k, v in jobs.items()}
        n = int(jobs["q_len"].shape[0])
        out = torch.zeros((n, 7), dtype=torch.int32, device=DEV)
        sj = bpsw_hip.SwJobs()
        sj.n, sj.xtra = n, xtra
        for k in ("q_len", "t_len", "q_off", "t_off", "q_rev", "q_pool", "t_pool"):
            setattr(sj, k, d[k].data_ptr())
        sj.q_pool_bytes, sj.t_pool_bytes = d["q_pool"].numel(), d["t_pool"].numel()
        torch.cuda.synchronize()
        ctx.swalign2_batch_device(opt, sj, out.data_ptr(), 0)
        return d, sj, out

    small = synth.sw_jobs(300, read_len=100, win_min=300, win_max=400, seed=31)
    big = synth.sw_jobs(200, read_len=250, win_min=700, win_max=900, seed=32)
    for jobs in (small, small, big, big, small):   # same geometry (async), larger (re-launched at the sync), smaller (fits)
        d, sj, out = run(jobs)
        ctx.last_kernel_ms()
        want, _ = orc.sw_align2_jobs(orc.default_opt(), xtra, **jobs)
        assert np.array_equal(out.cpu().numpy(), want)
    # a job table that points outside its pool: deferred error
    d, sj, out = run(small)
    ctx.last_kernel_ms()
    broken = dict(small)
    broken["q_off"] = small["q_off"].copy()
    broken["q_off"][5] = 10 ** 9
    d, sj, out = run(broken)
    with pytest.raises(bpsw_hip.BpswError):
        ctx.last_kernel_ms()
