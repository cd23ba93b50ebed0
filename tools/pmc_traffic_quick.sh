#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the two hot kernels under the bench command, quick (2 steps): tools/pmc_traffic_quick.sh OUT [env...]
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  env "$@" true
  ( export "$@" DUMMY_=1; rocprofv3 --pmc $c --output-format csv -d $out/$c -o b -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $out/$c.json 2> $out/$c.err )
done
python3 - "$out" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = ("extend_full" if ", 0>" in r["Kernel_Name"] else "extend") if "ext_kernel" in r["Kernel_Name"] else ("swalign2" if "swp_kernel" in r["Kernel_Name"] else None)
            if k:
                per.setdefault(k, {}).setdefault(c, []).append(float(r["Counter_Value"]))
d = json.loads([l for l in open(f"{out}/FETCH_SIZE.json") if l.startswith("{")][-1])
for k, v in per.items():
    fe, wr = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    if k not in d["kernels"]:   # (the full kernel: launched only for what the host lists)
        continue
    alg = d["kernels"][k]["bytes_per_launch"]
    print(k, "fetch_KB", round(fe, 1), "write_KB", round(wr, 1), "traffic_B", int((2 * fe + wr) * 1024), "algorithmic_B", alg, "ratio", round((2 * fe + wr) * 1024 / alg, 2), "value", d["value"])
PY
