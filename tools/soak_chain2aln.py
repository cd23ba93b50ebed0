#!/usr/bin/env python3
"""Soak test of the on-device round loop (bpsw_chain2aln_batch) against the oracle's sequential walk: many seeds, read lengths
and error mixes (including indel-rich reads, whose flanks start with a gap: the closed forms of bpsw_extend_core.h run here on
the staged reference window, not on the wire batch), both z-drop parses.
Usage on a GPU box: python tools/soak_chain2aln.py [rounds] [reads_per_round]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402
import bpsw_hip  # noqa: E402
import pyoracle as po  # noqa: E402
from bpsw_hip import synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
per = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
ctx, orc = bpsw_hip.Context(0), po.Oracle()
l_pac = 500_003
pac, bases = synth.random_pac(l_pac, seed=4711)
ctx.ref_load(pac, l_pac)
total = bad_total = 0
for rd in range(rounds):
    for L, es, ei, tail in ((150, 0.01, 0.001, 0.0), (150, 0.005, 0.01, 0.0), (100, 0.02, 0.02, 0.1), (250, 0.03, 0.005, 0.02), (60, 0.01, 0.02, 0.0)):
        b = synth.read_chains(per, bases, l_pac, read_len=L, sub_rate=es, indel_rate=ei, tail_frac=tail, seed=9000 + 17 * rd + L)
        for zmode in (po.ZDROP_SCALA, po.ZDROP_BWA):
            want_cnt, want, _, _ = orc.chain2aln_batch(orc.default_opt(), pac, b, zmode)
            got_cnt, got = ctx.chain2aln_batch(bpsw_hip.default_opt(), b, zmode)
            ok = np.array_equal(got_cnt, want_cnt) and got.shape == want.shape and got.tobytes() == want.tobytes()
            total += per
            if not ok:
                bad_total += 1
                print(f"round {rd} L {L} es {es} ei {ei} zmode {zmode}: MISMATCH (counts equal: {np.array_equal(got_cnt, want_cnt)})", flush=True)
    print(f"round {rd}: {total} read runs so far, {bad_total} batches differ", flush=True)
print("SOAK_C2A", {"read_runs": total, "bad_batches": bad_total})
sys.exit(1 if bad_total else 0)
