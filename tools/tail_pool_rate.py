#!/usr/bin/env python3
"""worker2's tail through the library-owned pool (bpsw_tail_pool_*): ONE calling thread enqueues groups of 4 096 pairs and collects them;
reads/s by the number of pool workers, next to the same thread making the calls itself.
Usage on a GPU box:  python tools/tail_pool_rate.py [out.json] [n_pairs] [groups_in_flight]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cloud-scale-bwamem_amd"))
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
from bpsw_hip import synth, _pairs_struct, _ptr  # noqa: E402


def measure(n_pairs=4096, in_flight=32, rounds=4, workers=(1, 2, 4, 8, 12, 16)):
    ctx = bpsw_hip.Context(0)
    lib = ctx.lib
    pac, bases, off, ln, names, dups = synth.contig_reference([400_000, 300_000, 200_000, 100_000], seed=synth.CONFIG_SEED_BASE + 40)
    tb, rn, quals, pes = synth.tail_pairs(n_pairs, bases, off, ln, dups, seed=synth.CONFIG_SEED_BASE + 41)
    ctx.ref_load(pac, int(off[-1] + ln[-1]))
    ctx.bns_load(off, ln, names)
    opt = bpsw_hip.default_opt()
    cnt, regs = ctx.chain2aln_batch(opt, tb, flags=bpsw_hip.C2A_SORT_DEDUP)
    g = bpsw_hip.make_tail_group(tb, rn, quals, pes, cnt, regs, off, ln, names, id0=0)
    topt = bpsw_hip.default_tail_opt()
    texts, _ = ctx.sam_pe_batch(opt, topt, g)
    want = b"".join(texts)
    st, keep, regs_in = _pairs_struct(g)
    cap = len(want) + 4096
    bufs = [np.empty(cap, np.uint8) for _ in range(in_flight)]
    offs = [np.zeros(2 * n_pairs + 1, np.int64) for _ in range(in_flight)]
    need = C.c_size_t(0)
    # the calling thread makes the calls itself
    for _ in range(2):
        lib.bpsw_sam_pe_batch(ctx.h, C.byref(opt), C.byref(topt), C.byref(st), _ptr(bufs[0]), cap, _ptr(offs[0]), C.byref(need), None)
    t0 = time.perf_counter()
    n_direct = 16
    for k in range(n_direct):
        rc = lib.bpsw_sam_pe_batch(ctx.h, C.byref(opt), C.byref(topt), C.byref(st), _ptr(bufs[k % in_flight]), cap, _ptr(offs[k % in_flight]),
                                   C.byref(need), None)
        assert rc == 0
    dt = time.perf_counter() - t0
    out = {"pairs_per_group": n_pairs, "groups_in_flight": in_flight, "sam_bytes_per_group": len(want),
           "direct_calls_one_thread": {"reads_per_s": round(2 * n_pairs * n_direct / dt), "ms_per_group": round(1e3 * dt / n_direct, 3)},
           "pool_one_calling_thread": {}}
    for w in workers:
        pool = C.c_void_p()
        assert lib.bpsw_tail_pool_create(0, w, C.byref(pool)) == 0
        tick = (C.c_int64 * in_flight)()

        def one_round():
            for k in range(in_flight):
                t = C.c_int64(0)
                rc = lib.bpsw_tail_pool_submit(pool, C.byref(opt), C.byref(topt), C.byref(st), -1, _ptr(bufs[k]), cap, _ptr(offs[k]), None, None, 0,
                                               C.byref(t))
                assert rc == 0
                tick[k] = t.value
            for k in range(in_flight):
                assert lib.bpsw_tail_pool_wait(pool, tick[k], C.byref(need), None) == 0
        one_round()
        t0 = time.perf_counter()
        c0 = time.process_time()
        for _ in range(rounds):
            one_round()
        dt = time.perf_counter() - t0
        cpu = time.process_time() - c0
        same = all(bufs[k][: int(offs[k][-1])].tobytes() == want for k in (0, in_flight // 2, in_flight - 1))
        out["pool_one_calling_thread"][str(w)] = {"reads_per_s": round(2 * n_pairs * in_flight * rounds / dt),
                                                  "ms_per_group": round(1e3 * dt / (in_flight * rounds), 3),
                                                  "cpus_busy": round(cpu / dt, 2), "text_identical_to_direct_call": bool(same)}
        lib.bpsw_tail_pool_destroy(pool)
    out["note"] = ("bpsw_sam_pe_batch (memSamPeGroupRest: plan, reg2aln kernel, SAM text) of one synthetic group, host buffers in, SAM text out; the "
                   "calling thread submits `groups_in_flight` tickets and collects them, `rounds` times; cpus_busy = process CPU time / wall time")
    return out


if __name__ == "__main__":
    res = measure(int(sys.argv[2]) if len(sys.argv) > 2 else 4096, int(sys.argv[3]) if len(sys.argv) > 3 else 32)
    txt = json.dumps(res, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")
