"""worker2's tail on the GPU (bpsw_reg2aln_batch, bpsw_sam_pe_batch) against the reference's golden vectors and the oracle,
in both flavours, through the C ABI."""
import os

import numpy as np
import pytest

import bpsw_hip
import pyoracle as po
import copy
from bpsw_hip import synth

from tail_util import G, load_sam_pe_golden, rescue_group_of, synthetic_group, synthetic_group_with_bases

pytestmark = pytest.mark.gpu


def _load_ref(ctx, pac, g):
    ctx.ref_load(pac, g.l_pac)
    n = g.ann_off.shape[0]
    names = [bytes(g.ann_name_pool[int(g.ann_name_off[i]):int(g.ann_name_off[i + 1])]).decode() for i in range(n)]
    ctx.bns_load(g.ann_off, g.ann_len, names)


@pytest.mark.parametrize("stem", ["mem_sam_pe", "mem_sam_pe_all", "mem_sam_pe_rg"])
def test_sam_pe_vs_reference_golden_text(ctx, stem):
    pac, g, flag, want = load_sam_pe_golden(stem)
    _load_ref(ctx, pac, g)
    opt = bpsw_hip.default_opt()
    opt.flag = flag
    topt = bpsw_hip.default_tail_opt(bpsw_hip.TAIL_C)
    topt.rg_id = g.rg_id        # "mem_sam_pe_rg": the reference ran with -R "@RG\tID:lane7.A..." (RG:Z:lane7.A behind XS on every line)
    got, _ = ctx.sam_pe_batch(opt, topt, g)
    assert got == want          # the reference's mem_sam_pe output, byte for byte
    ms, n_jobs = ctx.last_tail_kernel()
    assert n_jobs > 0 and ms > 0


def test_reg2aln_vs_reference_golden(ctx):
    z = np.load(os.path.join(G, "mem_reg2aln.npz"))
    ctx.ref_load(z["pac"], int(z["l_pac"]))
    ctx.bns_load(z["ann_off"], z["ann_len"])
    regs = z["regs"].astype(bpsw_hip.ALNREG_DTYPE)
    alns, cig, md = ctx.reg2aln_batch(bpsw_hip.default_opt(), bpsw_hip.default_tail_opt(bpsw_hip.TAIL_C), z["read_len"], z["read_off"],
                                      z["read_pool"], regs, max_cigar=32, max_md=160)
    want = z["alns"]
    assert int((alns["status"] != 0).sum()) == 0
    for f in ("pos", "rid", "flag", "is_rev", "mapq", "NM", "n_cigar", "score", "sub", "md_len"):
        assert np.array_equal(alns[f], want[f]), f
    assert np.array_equal(cig, z["cigar"]) and np.array_equal(md, z["md"])


def test_sam_pe_text_buffer_too_small_reports_the_size_and_a_second_call_gives_the_text(ctx, orc):
    """bpsw_sam_pe_batch writes straight into the caller's buffer: past its capacity it only counts (BPSW_ERR_CAPACITY, *out_needed,
    a complete out_off), and nothing is written beyond text_cap (what is below it is a part of the text, not specified further)"""
    import ctypes as C
    from bpsw_hip import _pairs_struct, _ptr
    pac, g = synthetic_group(orc, 200, 4242, read_len=150, sub_rate=0.02, indel_rate=0.004, p_span=0.05)
    _load_ref(ctx, pac, g)
    opt, topt = bpsw_hip.default_opt(), bpsw_hip.default_tail_opt(bpsw_hip.TAIL_SCALA)
    want, _ = ctx.sam_pe_batch(opt, topt, g)
    full = b"".join(want)
    st, keep, regs = _pairs_struct(g)
    off = np.zeros(2 * g.group_size + 1, np.int64)
    need = C.c_size_t(0)
    for cap in (0, 1, 777, len(full) - 1):
        buf = np.full(cap + 64, 0xAB, np.uint8)
        rc = ctx.lib.bpsw_sam_pe_batch(ctx.h, C.byref(opt), C.byref(topt), C.byref(st), _ptr(buf), cap, _ptr(off), C.byref(need), None)
        assert rc == -3 and need.value == len(full) and int(off[-1]) == len(full)
        assert (buf[cap:] == 0xAB).all()
    buf = np.zeros(len(full), np.uint8)
    rc = ctx.lib.bpsw_sam_pe_batch(ctx.h, C.byref(opt), C.byref(topt), C.byref(st), _ptr(buf), len(full), _ptr(off), C.byref(need), None)
    assert rc == 0 and buf.tobytes() == full


@pytest.mark.parametrize("flavour", [bpsw_hip.TAIL_SCALA, bpsw_hip.TAIL_C])
@pytest.mark.parametrize("L,es,ei,flag", [(150, 0.01, 0.002, 0), (150, 0.05, 0.02, bpsw_hip.MEM_F_ALL), (250, 0.08, 0.02, 0),
                                          (100, 0.02, 0.005, bpsw_hip.MEM_F_NO_MULTI | bpsw_hip.MEM_F_ALL), (150, 0.02, 0.004, bpsw_hip.MEM_F_NOPAIRING)])
def test_sam_pe_vs_oracle(ctx, orc, flavour, L, es, ei, flag):
    pac, g = synthetic_group(orc, 400, 3000 + L + flavour, read_len=L, sub_rate=es, indel_rate=ei, p_span=0.05)
    _load_ref(ctx, pac, g)
    opt, oopt = bpsw_hip.default_opt(), orc.default_opt()
    opt.flag = oopt.flag = flag
    otopt, topt = orc.default_tail_opt(), bpsw_hip.default_tail_opt(flavour)
    if flag & bpsw_hip.MEM_F_ALL:       # two of the cases run with a read group: RG:Z:<id> behind XS on every line (R2S:496-500)
        otopt.rg_id = topt.rg_id = b"run12.lane3"
    want, want_regs, n_jobs = orc.sam_pe_batch(oopt, otopt, pac, g, flavour=flavour)
    got, got_regs = ctx.sam_pe_batch(opt, topt, g)
    assert all((b"\tRG:Z:run12.lane3" in w) == bool(flag & bpsw_hip.MEM_F_ALL) for w in want)
    bad = [i for i in range(len(want)) if want[i] != got[i]]
    assert not bad, (len(bad), want[bad[0]], got[bad[0]])
    assert want_regs.tobytes() == got_regs.tobytes()
    assert 0 < ctx.last_tail_kernel()[1] <= n_jobs      # one job per distinct (read, region)


def test_reg2aln_jobs_vs_oracle_and_edge_cases(ctx, orc):
    pac, g = synthetic_group(orc, 300, 5151, sub_rate=0.04, indel_rate=0.02, p_span=0.1)
    _load_ref(ctx, pac, g)
    rl, ro = [], []
    for r in range(2 * g.group_size):
        rl += [int(g.read_len[r])] * int(g.reg_cnt[r]); ro += [int(g.read_off[r])] * int(g.reg_cnt[r])
    regs = g.regs.copy()
    unm = np.zeros(1, regs.dtype); unm["rb"] = -1; unm["re"] = -1
    regs = np.concatenate([regs, unm]); rl.append(int(g.read_len[0])); ro.append(int(g.read_off[0]))
    for flavour in (bpsw_hip.TAIL_SCALA, bpsw_hip.TAIL_C):
        want, wc, wm = orc.reg2aln_batch(orc.default_opt(), orc.default_tail_opt(), pac, g.l_pac, g.ann_off, g.ann_len, rl, ro, g.read_pool,
                                         regs, flavour=flavour, cigar_cap=48, md_cap=320)
        got, gc, gm = ctx.reg2aln_batch(bpsw_hip.default_opt(), bpsw_hip.default_tail_opt(flavour), rl, ro, g.read_pool, regs, max_cigar=48, max_md=320)
        for f in want.dtype.names:
            assert np.array_equal(want[f], got[f]), (flavour, f)
        assert np.array_equal(wc, gc) and np.array_equal(wm, gm)
    assert ctx.reg2aln_batch(bpsw_hip.default_opt(), bpsw_hip.default_tail_opt(), [], [], g.read_pool, regs[:0])[0].shape[0] == 0
    # a tiny room for the CIGAR: the job reports how many operations it needs and produces none of them
    a, c, _ = ctx.reg2aln_batch(bpsw_hip.default_opt(), bpsw_hip.default_tail_opt(), rl, ro, g.read_pool, regs, max_cigar=1, max_md=320)
    big = want["n_cigar"] > 1
    assert big.any() and np.array_equal(a["n_cigar"], want["n_cigar"]) and not c[big].any()
    bad = regs[:1].copy(); bad["re"] = 2 * g.l_pac + 5
    with pytest.raises(bpsw_hip.BpswError):
        ctx.reg2aln_batch(bpsw_hip.default_opt(), bpsw_hip.default_tail_opt(), rl[:1], ro[:1], g.read_pool, bad)
    bad = regs[:1].copy(); bad["qe"] = 10_000
    with pytest.raises(bpsw_hip.BpswError):
        ctx.reg2aln_batch(bpsw_hip.default_opt(), bpsw_hip.default_tail_opt(), rl[:1], ro[:1], g.read_pool, bad)


@pytest.mark.parametrize("rl", [dict(sub_rate=0.02, indel_rate=0.004), dict(read_len=250, sub_rate=0.08, indel_rate=0.02)], ids=["150bp", "250bp"])
@pytest.mark.parametrize("rescue_mode,flavour,zmode", [(bpsw_hip.RESCUE_C, bpsw_hip.TAIL_C, po.ZDROP_BWA),
                                                        (bpsw_hip.RESCUE_SCALA, bpsw_hip.TAIL_SCALA, po.ZDROP_SCALA)])
def test_worker2_in_one_call_vs_oracle_pipeline(ctx, orc, rescue_mode, flavour, zmode, rl):
    """bpsw_worker2_batch = prepare + rescue (windows from the resident reference) + tail, against the oracle's own pipeline
    (windows cut as bytes -> orc_matesw_group -> orc_sam_pe_batch) on pairs of which a fifth have an end without any seed.
    250bp: BASELINE.json configs[4]'s reads (8 % / 2 % error) through the same call."""
    pac, bases, g = synthetic_group_with_bases(orc, 500 if "read_len" not in rl else 300, 8800 + flavour, zdrop_mode=zmode, dedup_mode=rescue_mode,
                                               p_hard=0.2, p_unmappable=0.03, **rl)
    _load_ref(ctx, pac, g)
    opt, oopt = bpsw_hip.default_opt(), orc.default_opt()
    rg = rescue_group_of(g, bases, oopt)
    cnt_o, regs_o, n_sw, _ = orc.matesw_group(oopt, rg, rescue_mode)
    assert n_sw > 50 and regs_o.shape[0] > g.regs.shape[0]          # the rescue really added regions
    g2 = copy.copy(g)
    g2.reg_cnt, g2.regs = cnt_o, regs_o
    want, want_regs, _ = orc.sam_pe_batch(oopt, orc.default_tail_opt(), pac, g2, flavour=flavour)
    got, cnt, regs = ctx.worker2_batch(opt, bpsw_hip.default_tail_opt(flavour), g, rescue_mode)
    assert np.array_equal(cnt, cnt_o)
    bad = [i for i in range(len(want)) if want[i] != got[i]]
    assert not bad, (len(bad), want[bad[0]], got[bad[0]])
    assert regs.tobytes() == want_regs.tobytes()


def test_worker2_in_one_call_vs_reference_mem_sam_pe(ctx, orc, ref):
    """The same call against the reference's own mem_sam_pe INCLUDING its rescue (mem_matesw over the packed reference).
    Everything must agree except what derives from ksw_align2's second-best score (the SSE2 kernel inflates
    score2 -> csub -> the XS tag and the mapQ cap), so lines may differ in MAPQ / XS only, and only for a few percent."""
    pac, bases, g = synthetic_group_with_bases(orc, 500, 8900, zdrop_mode=po.ZDROP_BWA, sub_rate=0.02, indel_rate=0.004, p_hard=0.2,
                                               p_unmappable=0.03)
    _load_ref(ctx, pac, g)
    want = ref.sam_pe_batch(orc.default_opt(), orc.default_tail_opt(), pac, g, no_rescue=False)
    got, cnt, _ = ctx.worker2_batch(bpsw_hip.default_opt(), bpsw_hip.default_tail_opt(bpsw_hip.TAIL_C), g, bpsw_hip.RESCUE_C)
    assert int(cnt.sum()) > int(g.reg_cnt.sum())

    def strip(line):   # drop MAPQ (field 5) and the XS tag
        f = line.rstrip(b"\n").split(b"\t")
        return [x for k, x in enumerate(f) if k != 4 and not x.startswith(b"XS:i:")]

    diff = [i for i in range(len(want)) if want[i] != got[i]]
    assert len(diff) <= 0.05 * len(want), len(diff)
    for i in diff:
        wl, gl = want[i].splitlines(), got[i].splitlines()
        assert len(wl) == len(gl) and all(strip(a) == strip(b) for a, b in zip(wl, gl)), (want[i], got[i])


def _slice_group(g, lo, hi):
    """pairs lo..hi-1 of a TailGroupSoA as their own group (same pools; id0 moves so that every pair keeps its id)"""
    r0, r1 = int(g.reg_cnt[:2 * lo].sum()), int(g.reg_cnt[:2 * hi].sum())
    s = copy.copy(g)
    s.group_size, s.id0 = hi - lo, g.id0 + lo
    s.read_len, s.read_off = np.ascontiguousarray(g.read_len[2 * lo:2 * hi]), np.ascontiguousarray(g.read_off[2 * lo:2 * hi])
    s.name_off = np.ascontiguousarray(g.name_off[lo:hi + 1])        # offsets stay absolute into the shared name pool
    s.reg_cnt, s.regs = np.ascontiguousarray(g.reg_cnt[2 * lo:2 * hi]), np.ascontiguousarray(g.regs[r0:r1])
    return s


def test_tail_scale_properties(ctx, orc):
    """16 384 pairs (regions from the device's own memChainToAln + memSortAndDedup): the text does not depend on how the pairs
    are grouped into calls, repeated calls agree, every read gets at least one line with its own name and flag bits that are
    consistent between mates, and a slice in the middle matches the oracle."""
    from bpsw_hip import synth
    n_pairs = 16384
    pac, bases, off, ln, names, dups = synth.contig_reference([400_000, 300_000, 200_000, 100_000], seed=4242)
    tb, rn, quals, pes = synth.tail_pairs(n_pairs, bases, off, ln, dups, seed=4243, p_hard=0.05)
    ctx.ref_load(pac, int(off[-1] + ln[-1]))
    ctx.bns_load(off, ln, names)
    opt, topt = bpsw_hip.default_opt(), bpsw_hip.default_tail_opt()
    cnt, regs = ctx.chain2aln_batch(opt, tb, flags=bpsw_hip.C2A_SORT_DEDUP)
    g = bpsw_hip.make_tail_group(tb, rn, quals, pes, cnt, regs, off, ln, names, id0=1_000_000)
    whole, _ = ctx.sam_pe_batch(opt, topt, g)
    again, _ = ctx.sam_pe_batch(opt, topt, g)
    assert whole == again
    parts = []
    for lo, hi in ((0, 1), (1, 4097), (4097, 12000), (12000, n_pairs)):
        t, _ = ctx.sam_pe_batch(opt, topt, _slice_group(g, lo, hi))
        parts += t
    assert parts == whole
    n_proper = 0
    for k in range(n_pairs):
        a, b = whole[2 * k].split(b"\n")[0].split(b"\t"), whole[2 * k + 1].split(b"\n")[0].split(b"\t")
        assert a[0] == b[0] == rn[k].encode()
        fa, fb = int(a[1]), int(b[1])
        assert fa & 0x41 == 0x41 and fb & 0x81 == 0x81                      # paired, first / second in pair
        assert bool(fa & 0x8) == bool(fb & 0x4) and bool(fb & 0x8) == bool(fa & 0x4)   # mate-unmapped mirrors unmapped
        assert bool(fa & 0x2) == bool(fb & 0x2)
        n_proper += bool(fa & 0x2)
    assert n_proper > 0.7 * n_pairs
    lo, hi = 7000, 7400
    sub = _slice_group(g, lo, hi)
    want, _, _ = orc.sam_pe_batch(orc.default_opt(), orc.default_tail_opt(), pac, sub)
    assert whole[2 * lo:2 * hi] == want


def test_chains_to_sam_end_to_end_vs_reference(ctx, orc, ref):
    """Everything the library offers, chained as the driver chains it -- chains -> bpsw_chain2aln_batch (round loop + sort/dedup)
    -> bpsw_pe_stat -> bpsw_worker2_batch (rescue + tail) -> SAM text -- against the reference's own C pipeline on the same
    chains: mem_chain2aln + mem_sort_and_dedup, mem_pestat, mem_sam_pe.  The region lists and the insert-size statistics must be
    identical; the text may differ in MAPQ / XS only, and only for a few percent of the pairs."""
    pac, bases, off, ln, names, dups = synth.contig_reference([70_000, 50_000, 30_000, 50_000], seed=9100)
    tb, rn, quals, _ = synth.tail_pairs(600, bases, off, ln, dups, seed=9101, sub_rate=0.02, indel_rate=0.004, p_hard=0.15,
                                        p_unmappable=0.03)
    opt, oopt = bpsw_hip.default_opt(), orc.default_opt()
    topt, otopt = bpsw_hip.default_tail_opt(bpsw_hip.TAIL_C), orc.default_tail_opt()
    l_pac = int(tb.l_pac)
    ctx.ref_load(pac, l_pac)
    # worker1's second half
    cnt, regs = ctx.chain2aln_batch(opt, tb, po.ZDROP_BWA, bpsw_hip.C2A_SORT_DEDUP)
    rcnt, rregs = ref.chain2aln_batch(oopt, pac, tb)
    want_regs, at = [], 0
    for c in rcnt:
        want_regs.append(ref.sort_dedup(rregs[at:at + c]) if c else rregs[0:0])
        at += c
    assert np.array_equal(cnt, [len(r) for r in want_regs]) and regs.tobytes() == np.concatenate(want_regs).tobytes()
    # the driver's statistics
    pes = bpsw_hip.pe_stat(opt, topt, l_pac, cnt, regs)
    assert pes == ref.pestat(oopt, otopt, l_pac, cnt, regs) and pes[1][2] == 0
    # worker2
    g = bpsw_hip.make_tail_group(tb, rn, quals, pes, cnt, regs, off, ln, names, id0=5)
    _load_ref(ctx, pac, g)
    want = ref.sam_pe_batch(oopt, otopt, pac, g, no_rescue=False)
    got, cnt2, _ = ctx.worker2_batch(opt, topt, g, bpsw_hip.RESCUE_C)
    assert int(cnt2.sum()) > int(cnt.sum())

    def strip(line):
        f = line.rstrip(b"\n").split(b"\t")
        return [x for k, x in enumerate(f) if k != 4 and not x.startswith(b"XS:i:")]

    diff = [i for i in range(len(want)) if want[i] != got[i]]
    assert len(diff) <= 0.05 * len(want), len(diff)
    for i in diff:
        wl, gl = want[i].splitlines(), got[i].splitlines()
        assert len(wl) == len(gl) and all(strip(a) == strip(b) for a, b in zip(wl, gl)), (want[i], got[i])


# ---- the tail off the calling thread: bpsw_tail_pool_* (round 5) ------------------------------------------------------------------------
def test_tail_pool_gives_the_text_of_the_direct_calls(ctx, orc):
    """Twelve groups (150 and 250 bp, both flavours, tail-only and rescue + tail) enqueued on a pool of four workers by ONE thread and
    collected in reverse order: every group's text, counts and regions are byte for byte what the direct call on a context returns, and
    the golden group of the reference's mem_sam_pe comes back as the reference wrote it."""
    pac, bases, g150 = synthetic_group_with_bases(orc, 400, 9100, zdrop_mode=po.ZDROP_BWA, dedup_mode=bpsw_hip.RESCUE_C, p_hard=0.2,
                                                  p_unmappable=0.03, sub_rate=0.02, indel_rate=0.004)
    _load_ref(ctx, pac, g150)
    opt = bpsw_hip.default_opt()
    jobs = []
    for k in range(6):
        gk = copy.copy(g150)
        gk.id0 = 1000 * k       # (the pair ids feed the hash that breaks ties: every group is a different call)
        jobs.append((gk, bpsw_hip.default_tail_opt(bpsw_hip.TAIL_C if k % 2 else bpsw_hip.TAIL_SCALA),
                     bpsw_hip.TAIL_POOL_TAIL_ONLY if k % 3 == 0 else (bpsw_hip.RESCUE_C if k % 3 == 1 else bpsw_hip.RESCUE_SCALA)))
    want = []
    for gk, topt, mode in jobs:
        want.append(ctx.sam_pe_batch(opt, topt, gk) if mode == bpsw_hip.TAIL_POOL_TAIL_ONLY else ctx.worker2_batch(opt, topt, gk, mode))
    pool = bpsw_hip.TailPool(0, workers=4)
    assert pool.workers == 4
    tickets = [pool.submit(opt, topt, gk, mode) for gk, topt, mode in jobs] * 1
    tickets += [pool.submit(opt, topt, gk, mode) for gk, topt, mode in jobs]      # the same groups again: twelve in flight
    for idx in reversed(range(len(tickets))):
        got, w = pool.collect(tickets[idx]), want[idx % len(jobs)]
        assert got[0] == w[0], idx
        for a, b in zip(got[1:], w[1:]):
            assert np.asarray(a).tobytes() == np.asarray(b).tobytes(), idx
    pool.close()


def test_tail_pool_golden_text_and_errors(ctx, orc):
    pac, g, flag, want = load_sam_pe_golden("mem_sam_pe")
    _load_ref(ctx, pac, g)
    opt = bpsw_hip.default_opt()
    opt.flag = flag
    topt = bpsw_hip.default_tail_opt(bpsw_hip.TAIL_C)
    pool = bpsw_hip.TailPool(0, workers=2)
    t_small = pool.submit(opt, topt, g, text_cap=100)     # far too small: the ticket reports BPSW_ERR_CAPACITY and the size, collect() re-submits
    t_ok = pool.submit(opt, topt, g)
    assert pool.collect(t_ok)[0] == want                  # the reference's mem_sam_pe output, byte for byte
    assert pool.collect(t_small)[0] == want
    assert t_small.cap > 100
    with pytest.raises(bpsw_hip.BpswError, match="unknown ticket"):
        pool.collect(t_ok)                                # each ticket is collected once
    bad = copy.copy(g)
    bad.read_len = np.asarray(g.read_len).copy()
    bad.read_len[0] = 10_000_000                          # a read outside its pool: the worker's error reaches the collecting thread
    t_bad = pool.submit(opt, topt, bad)
    with pytest.raises(bpsw_hip.BpswError, match="outside its pool"):
        pool.collect(t_bad)
    t_after = pool.submit(opt, topt, g)                   # the worker that met the error goes on
    assert pool.collect(t_after)[0] == want
    pool.close()


def test_tail_pool_shared_by_many_submitting_threads(ctx, orc):
    """The JNI shim keeps ONE pool per device for all the executor's task threads: six threads submit and collect their own groups on a
    pool of three workers at once; every thread gets its own groups' text back."""
    import threading
    pac, g = synthetic_group(orc, 300, 9300, read_len=150, sub_rate=0.02, indel_rate=0.004, p_span=0.05)
    _load_ref(ctx, pac, g)
    opt, topt = bpsw_hip.default_opt(), bpsw_hip.default_tail_opt(bpsw_hip.TAIL_SCALA)
    groups, want = [], []
    for k in range(6):
        gk = copy.copy(g)
        gk.id0 = 7000 * k
        groups.append(gk)
        want.append(ctx.sam_pe_batch(opt, topt, gk)[0])
    pool = bpsw_hip.TailPool(0, workers=3)
    errors = []

    def task(k):
        try:
            for _ in range(3):
                tickets = [pool.submit(opt, topt, groups[k]) for _ in range(4)]
                for t in tickets:
                    assert pool.collect(t)[0] == want[k]
        except BaseException as e:   # noqa: BLE001 - reported to the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=task, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not any(t.is_alive() for t in threads)
    assert not errors, errors[:2]
    pool.close()
