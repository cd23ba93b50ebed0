#!/bin/bash
# On the GPU box: instruction counters of the extension kernel for stand-alone batches (tools/ext_kernel_time.py).
#   tools/pmc_ext.sh [args of ext_kernel_time.py]  -> gpurun_out/pmc_ext/*.csv + one summary line per kernel
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_ext
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/a -o ext -- python tools/ext_kernel_time.py "$@" > $out/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $out/b -o ext -- python tools/ext_kernel_time.py "$@" > $out/b.log 2>&1
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_ext/*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
