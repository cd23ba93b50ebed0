#!/bin/bash
# instructions per DP row of ext_kernel by flank length (run on a GPU box): tools/pmc_rowcost.sh OUTDIR L...
set -e
SUB=${SUB:-0.05}; INDEL=${INDEL:-0.01}; MASK=${MASK:-0}; NTASK=${NTASK:-16384}
out=$1; shift
mkdir -p $out; out=$(cd $out && pwd)
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_$L -o r -- python3 $GRAFT_REPO_ROOT/tools/ext_row_cost.py $L $SUB $INDEL $MASK $NTASK > $out/row_$L.log 2>&1
  python3 - "$out" "$L" <<'PY'
import csv, glob, sys, ast
out, L = sys.argv[1], sys.argv[2]
line = [l for l in open(f"{out}/row_{L}.log") if l.startswith("{")][-1]
d = ast.literal_eval(line)
tot = {}
for f in glob.glob(f"{out}/pmc_{L}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ext_kernel" in r["Kernel_Name"] or "ext_quad_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
per = {k: v / d["launches"] / d["rows"] for k, v in tot.items()}
print("L", L, d, "insts_per_row", {k: round(v, 1) for k, v in per.items()}, "sum", round(sum(per.values()), 1))
PY
done
