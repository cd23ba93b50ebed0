"""The reference's native code is called concurrently from several Spark task threads of one executor JVM (SURVEY.md 8b
"Threading"): one context per thread, all on the same device, sharing the device-resident reference.  Results must not
depend on what the other threads are doing."""
import threading

import numpy as np
import pytest

import bpsw_hip
from bpsw_hip import synth
import pyoracle as po
from conftest import region_fields_equal

pytestmark = pytest.mark.gpu


def test_concurrent_contexts_give_identical_results():
    l_pac = 500_009
    pac, bases = synth.random_pac(l_pac, seed=91)
    soa = synth.ext_tasks(6000, seed=92)
    wire = bpsw_hip.wire_pack(soa)
    g = synth.rescue_group(300, seed=93, l_pac=l_pac, p_resc=0.4, ref_bases=bases)
    import dataclasses
    gc = dataclasses.replace(g, ref_pool=None, ref_len=None, ref_off=None)
    b = synth.read_chains(1500, bases, l_pac, seed=94)
    main = bpsw_hip.Context(0)
    main.ref_load(pac, l_pac)
    opt = bpsw_hip.default_opt()
    want_ext = main.extend_batch(wire)
    want_cnt, want_regs = main.matesw_group(opt, g)
    want_ccnt, want_cregs = main.chain2aln_batch(opt, b)
    errors = []

    def worker(tid):
        try:
            c = bpsw_hip.Context(0)   # the way the JNI shim keeps one context per task thread
            for it in range(6):
                kind = (tid + it) % 4
                if kind == 0:
                    assert np.array_equal(c.extend_batch(wire), want_ext)
                elif kind == 1:
                    cnt, regs = c.matesw_group(opt, g)
                    assert np.array_equal(cnt, want_cnt); region_fields_equal(regs, want_regs)
                elif kind == 2:
                    cnt, regs = c.matesw_group(opt, gc)     # windows read from the shared device reference
                    assert np.array_equal(cnt, want_cnt); region_fields_equal(regs, want_regs)
                else:
                    cnt, regs = c.chain2aln_batch(opt, b)
                    assert np.array_equal(cnt, want_ccnt) and np.array_equal(regs, want_cregs)
            c.close()
        except BaseException as e:   # noqa: BLE001 - reported to the main thread
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    main.close()


def test_concurrent_tail_and_worker2_calls(orc):
    """worker2's tail from several threads at once (one context each), all reading the same device-resident reference and
    contig table: every thread must get the text a lone call gets."""
    from tail_util import synthetic_group
    pac, g = synthetic_group(orc, 300, 6060, sub_rate=0.03, indel_rate=0.008, p_hard=0.15, p_span=0.05)
    names = [bytes(g.ann_name_pool[int(g.ann_name_off[i]):int(g.ann_name_off[i + 1])]).decode() for i in range(g.ann_off.shape[0])]
    main = bpsw_hip.Context(0)
    main.ref_load(pac, g.l_pac)
    main.bns_load(g.ann_off, g.ann_len, names)
    opt, topt = bpsw_hip.default_opt(), bpsw_hip.default_tail_opt()
    want_tail, _ = main.sam_pe_batch(opt, topt, g)
    want_w2, want_cnt, _ = main.worker2_batch(opt, topt, g)
    errors = []

    def worker(tid):
        try:
            c = bpsw_hip.Context(0)
            for it in range(5):
                if (tid + it) % 2:
                    got, _ = c.sam_pe_batch(opt, topt, g)
                    assert got == want_tail
                else:
                    got, cnt, _ = c.worker2_batch(opt, topt, g)
                    assert got == want_w2 and np.array_equal(cnt, want_cnt)
            c.close()
        except BaseException as e:   # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    main.close()
