#!/bin/bash
# wave-instructions per extension call of configs[4] (2x250 bp), by build / switch (run on a GPU box): tools/pmc_cfg5_instr.sh OUTDIR [label env=val ...]...
set -e
out=$1; shift; mkdir -p $out; out=$(cd $out && pwd)
cd /tmp && export TMPDIR=/tmp
run() {
  label=$1; shift
  ( export BENCH_ONLY=ext "$@"; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $out/$label -o r -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $out/$label.json 2> $out/$label.err )
  python3 - "$out" "$label" <<'PY'
import csv, glob, sys, collections
out, label = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(f"{out}/{label}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "ext_kernel" + k.split("ext_kernel")[1][:10] if "ext_kernel" in k else ("sift" if "sift" in k else None)
        if not k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k in tot:
    print(label, k, "launches", n[k], {c: round(v / max(n[k], 1) / 1e6, 2) for c, v in tot[k].items()}, "(M per launch)")
PY
}
if [ $# -eq 0 ]; then run default BPSW_NOP=1; else for spec in "$@"; do run $spec; done; fi
