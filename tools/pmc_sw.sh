#!/bin/bash
# On the GPU box: instruction counters of the rescue kernel for one stand-alone batch (tools/sw_kernel_time.py).
#   tools/pmc_sw.sh <n_jobs>  -> gpurun_out/pmc_sw/*.csv
set -e
n=${1:-7208}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_sw
mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $out/a -o sw -- python tools/sw_kernel_time.py $n > $out/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY --output-format csv -d $out/b -o sw -- python tools/sw_kernel_time.py $n > $out/b.log 2>&1
python - <<'PY'
import csv, glob, collections
for sub in ("a", "b"):
    for f in glob.glob(f"gpurun_out/pmc_sw/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "sw" in k:
                print(sub, k, {c: round(x) for c, x in v.items()})
PY
