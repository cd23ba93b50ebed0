#!/bin/bash
# What the exact shortcuts' own evaluation costs: one bench batch with the quad kernel taking every DP row (BPSW_EXT_QUAD=1), so that
# ext_kernel's instruction count is the shortcuts + the per-task overhead, at shortcut masks 0 / 1 / 3 / 7 / 15 / 31 (bpsw_set_ext_shortcuts).
#   tools/pmc_shortcut_cost.sh [config]      (on a GPU box)
cfg=${1:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BPSW_EXT_QUAD=1
for m in 0 16 17 19 23 31; do
  out=gpurun_out/pmc_sc/$m; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out -o r -- python3 tools/ext_batch_instr.py $m $cfg > $out/log.txt 2>&1
  python3 - $out $m <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "quad" if "ext_quad" in r["Kernel_Name"] else ("ext" if "ext_kernel" in r["Kernel_Name"] else None)
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("mask", sys.argv[2], {k: {c: round(sum(x) / max(len([y for y in x if y > 1e5]), 1) / 1e6, 2) for c, x in v.items()} for k, v in acc.items()})
PY
done
