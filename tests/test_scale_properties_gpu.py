"""Size-independent properties at BASELINE.json scale (config 2: 1 M single-end 150 bp reads; config 5 shape: 2x250 bp,
high error).  The oracle is too slow to re-run a million tasks in a test, so the full-size run is checked through
properties the path must have, plus the oracle on a random sample:
  * determinism / idempotence: the same batch twice -> identical bytes;
  * batching invariance: a task's result does not depend on which batch it travels in or on its position
    (the reference's round structure re-batches tasks arbitrarily, MemChainToAlignBatched.scala:471-615);
  * permutation equivariance: shuffling the tasks of a batch permutes the results (idx travels with the task);
  * a checksum of per-batch checksums is stable across the two batchings;
  * oracle agreement on a uniform sample of the million tasks.
"""
import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth

pytestmark = pytest.mark.gpu
BATCH = 32768


def _run_batches(ctx, soa, order, batch):
    """results indexed by task position in `soa` (idx is rewritten per batch to the local position)"""
    out = np.zeros((soa.n, 10), np.int16)
    sums = []
    for s in range(0, len(order), batch):
        sel = order[s:s + batch]
        sub = soa.subset(sel)
        sub.idx = np.arange(len(sel), dtype=np.int32)
        res = ctx.extend_batch(bpsw_hip.wire_pack(sub)).reshape(-1, 10)
        idx = (res[:, 0].astype(np.int64) & 0xffff) | (res[:, 1].astype(np.int64) << 16)
        assert np.array_equal(idx, np.arange(len(sel)))          # results come back in task order with their idx
        out[sel] = res
        sums.append(int(res[:, 2:9].astype(np.int64).sum()))
    return out, sums


@pytest.mark.parametrize("n_reads,read_len,sub,indel,tail", [(1_000_000, 150, 0.01, 0.001, 0.0), (200_000, 250, 0.08, 0.02, 0.05)])
def test_full_size_properties(ctx, orc, n_reads, read_len, sub, indel, tail):
    soa = synth.ext_tasks(n_reads, read_len=read_len, sub_rate=sub, indel_rate=indel, tail_frac=tail,
                          seed=synth.CONFIG_SEED_BASE + (2 if read_len == 150 else 5))
    assert soa.n > 0.8 * n_reads
    rng = np.random.default_rng(0)
    natural = np.arange(soa.n)
    a, sums_a = _run_batches(ctx, soa, natural, BATCH)
    b, _ = _run_batches(ctx, soa, natural, BATCH)
    assert np.array_equal(a, b)                                   # deterministic / idempotent
    perm = rng.permutation(soa.n)
    c, sums_c = _run_batches(ctx, soa, perm, 20000)               # other batch size, other neighbours, other order
    assert np.array_equal(a[:, 2:9], c[:, 2:9])                   # batching invariance + permutation equivariance
    assert sum(sums_a) == sum(sums_c)                             # checksum of checksums
    # sanity of the values themselves: scores never below the seed score, coordinates inside the read
    assert (a[:, 6] >= soa.h0.astype(np.int16)).all() and (a[:, 2] >= 0).all() and (a[:, 3] >= 0).all()
    # oracle on a uniform sample
    pick = np.sort(rng.choice(soa.n, size=3000, replace=False))
    sub_soa = soa.subset(pick)
    sub_soa.idx = np.arange(len(pick), dtype=np.int32)
    want, _ = orc.wire_extend(bpsw_hip.wire_pack(sub_soa))
    assert np.array_equal(want.reshape(-1, 10)[:, 2:9], a[pick][:, 2:9])


def test_swalign_full_size_properties(ctx, orc):
    """100 k rescue jobs (config 3 has ~1 M; the kernel is ~1e5 cells per job): determinism, order independence, sample."""
    jobs = synth.sw_jobs(100_000, read_len=150, seed=synth.CONFIG_SEED_BASE + 3)
    xtra = bpsw_hip.KSW_XSUBO | bpsw_hip.KSW_XSTART | bpsw_hip.KSW_XBYTE | 19
    opt = bpsw_hip.default_opt()
    a = ctx.swalign2_batch(opt, xtra, **jobs)
    assert np.array_equal(a, ctx.swalign2_batch(opt, xtra, **jobs))
    perm = np.random.default_rng(1).permutation(len(a))
    shuffled = dict(jobs)
    for k in ("q_len", "t_len", "q_off", "t_off", "q_rev"):
        shuffled[k] = jobs[k][perm]
    assert np.array_equal(ctx.swalign2_batch(opt, xtra, **shuffled), a[perm])
    # invariants of SWAlign2: start <= end, second best below best, begin set iff the score reached the threshold
    ok = a[:, 0] >= 19
    assert (a[ok, 5] <= a[ok, 1]).all() and (a[ok, 6] <= a[ok, 2]).all() and (a[ok, 5] >= 0).all()
    assert (a[:, 3] <= a[:, 0]).all() and (a[~ok, 5] == -1).all()
    pick = np.sort(np.random.default_rng(2).choice(len(a), 400, replace=False))
    sub = dict(jobs)
    for k in ("q_len", "t_len", "q_off", "t_off", "q_rev"):
        sub[k] = jobs[k][pick]
    want, _ = orc.sw_align2_jobs(orc.default_opt(), xtra, **sub)
    assert np.array_equal(want, a[pick])


def test_chain2aln_full_size_properties(ctx, orc):
    """The on-device round loop at 100 k reads: independence of how the reads are batched, determinism,
    structural invariants of every region, and oracle parity on a sample."""
    l_pac = 4_000_037
    pac, bases = synth.random_pac(l_pac, seed=101)
    ctx.ref_load(pac, l_pac)
    n = 100_000
    b = synth.read_chains(n, bases, l_pac, read_len=150, sub_rate=0.01, indel_rate=0.001, seed=102)
    opt = bpsw_hip.default_opt()
    cnt, regs = ctx.chain2aln_batch(opt, b)
    cnt2, regs2 = ctx.chain2aln_batch(opt, b)
    assert np.array_equal(cnt, cnt2) and np.array_equal(regs, regs2)                     # deterministic
    parts = [ctx.chain2aln_batch(opt, b.slice(lo, min(lo + 32768, n))) for lo in range(0, n, 32768)]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), cnt)                    # batching does not matter
    assert np.array_equal(np.concatenate([p[1] for p in parts]), regs)
    assert int(cnt.sum()) == regs.shape[0] <= b.seed_len.shape[0]
    assert (regs["qb"] >= 0).all() and (regs["qe"] <= 150).all() and (regs["qb"] < regs["qe"]).all()
    assert (regs["rb"] < regs["re"]).all() and (regs["rb"] >= 0).all() and (regs["re"] <= 2 * l_pac).all()
    assert ((regs["rb"] < l_pac) == (regs["re"] <= l_pac)).all()                          # a region stays on one strand
    assert (regs["score"] >= 19).all() and (regs["truesc"] <= regs["score"] + 5).all()    # seed score; to-end bonus pen_clip
    assert (regs["seedcov"] >= 19).all()
    lo = 41_000
    sub = b.slice(lo, lo + 3000)
    want_cnt, want, _, _ = orc.chain2aln_batch(orc.default_opt(), pac, sub)
    at = int(cnt[:lo].sum())
    assert np.array_equal(cnt[lo:lo + 3000], want_cnt)
    got = regs[at:at + int(want_cnt.sum())]
    for f in want.dtype.names:
        assert np.array_equal(got[f], want[f]), f
