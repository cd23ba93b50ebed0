#!/usr/bin/env python3
"""Soak of the sift kernel on COORDINATE batches (wire format 2, csrc/bpsw_extend_sift.hip <true>): target flanks expanded from the
2-bit reference, both strands, left flanks read backwards.  Per round: a fresh counter-hash reference, reads at several error
rates, the batch through ext_kernel alone and through sift + ext_kernel -- results equal to the oracle on the byte form of the same
tasks, per-side verdicts equal between the two.  Usage on a GPU box: python tools/soak_sift_coords.py [rounds] [reads_per_round]"""
import os
import sys

os.environ.setdefault("BPSW_EXT_SIFT_MIN", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402
from bpsw_hip import synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
per = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
ctx, ctx_wave, orc = bpsw_hip.Context(0), bpsw_hip.Context(0), po.Oracle()
ctx_wave.set_ext_shortcuts(31)
total = bad = 0
for rd in range(rounds):
    l_pac = 1_000_003 + 37 * rd
    pac = synth.hash_pac(l_pac, seed=7000 + rd)
    ctx.ref_load(pac, l_pac)          # (the reference is per device: both contexts see it)
    for sub, indel in ((0.003, 0.0003), (0.01, 0.001), (0.02, 0.004), (0.04, 0.01)):
        by, co = synth.ext_tasks_ref(per, pac, l_pac, read_len=int(np.random.default_rng(rd).choice([100, 150, 150, 125])), sub_rate=sub,
                                     indel_rate=indel, seed=7100 + 10 * rd + int(sub * 1000))
        wire_c = bpsw_hip.wire_coords_pack(co)
        want, _ = orc.wire_extend(bpsw_hip.wire_pack(by))
        out1, how1 = ctx.extend_batch_classify(wire_c)
        out0, how0 = ctx_wave.extend_batch_classify(wire_c)
        d = int((out1.reshape(-1, 10) != want.reshape(-1, 10)).any(axis=1).sum() + (out0.reshape(-1, 10) != want.reshape(-1, 10)).any(axis=1).sum()
                + (how0 != how1).any(axis=1).sum())
        total += co.n
        bad += d
    print(f"round {rd} tasks so far {total} bad {bad} resolved by a form {(how1 == 1).sum() / max(1, (how1 != 0).sum()):.2f}", flush=True)
print("SOAK_SIFT_COORDS", {"task_runs": total, "bad": bad})
sys.exit(1 if bad else 0)
