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
print(json.dumps(jnishim.shim_rate(wire, soa.n, grp, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 7)))
