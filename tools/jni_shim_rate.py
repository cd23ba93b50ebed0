#!/usr/bin/env python3
"""What the JNI shim costs around the C ABI at the reference's batch sizes (32 768 reads per swExtendFPGAJNI call, run_test.sh:7;
4 096 pairs per mateSWJNI call), through the fake JNIEnv.  Usage on a GPU box: python tools/jni_shim_rate.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cloud-scale-bwamem_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bpsw_hip  # noqa: E402
from bpsw_hip import jnishim  # noqa: E402

W = bench.WORKLOADS[3]
soa = bench.make_ext_soa(W, 3, 0, 0)
wire = bpsw_hip.wire_pack(soa)
grp = bench.make_group(W, 3, 0, 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
res = jnishim.shim_rate(wire, soa.n, grp, reps=reps)
# the flat entry with the windows named by coordinates: a group drawn from a reference of its own (4 096 pairs as well)
from bpsw_hip import synth  # noqa: E402
l_pac = 4_000_003
pac, bases = synth.random_pac(l_pac, seed=synth.CONFIG_SEED_BASE + 91)
g_ref = synth.rescue_group(4096, seed=synth.CONFIG_SEED_BASE + 92, l_pac=l_pac, p_resc=W["p_resc"], ref_bases=bases)
r2 = jnishim.shim_rate(wire, soa.n, g_ref, reps=reps, pac=pac)
res["reference_backed_group"] = {k: r2[k] for k in ("mateSWJNI", "mateSWFlatJNI", "mateSWFlatJNI_coordinates") if k in r2}
print(json.dumps(res))
