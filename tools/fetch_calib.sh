#!/bin/bash
# tools/fetch_calib.sh OUT: FETCH_SIZE / WRITE_SIZE of tools/ubench/fetch_calib per kernel (KB) -> OUT/calib.txt
set -e
out=$1
cd /tmp && export TMPDIR=/tmp
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -o c -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib > $out/$c.log 2>&1
done
python3 - "$out" <<'PY' | tee $out/calib.txt
import csv, glob, sys
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            print(c, r["Kernel_Name"].split("(")[0], "KB", float(r["Counter_Value"]))
PY
