#!/bin/bash
# Instructions one bench-shaped extension batch issues, per kernel, for one or more library builds:
#   tools/pmc_batch_instr.sh OUT CONFIG lib...     (lib = path relative to the repo, or "default")
out=$1; cfg=$2; shift 2
mkdir -p $out; out=$(cd $out && pwd)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" = default ]; then unset BPSW_LIB; else export BPSW_LIB=$root/$lib; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/$tag -o r -- python3 $root/tools/ext_batch_instr.py 63 $cfg > $out/$tag.log 2>&1
  python3 - "$out/$tag" "$tag" <<'PY'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
        acc[m.group(1) if m else r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(sys.argv[2], k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in sorted(v.items())}, "M per launch;", len(next(iter(v.values()))), "launches")
PY
done
