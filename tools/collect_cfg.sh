#!/bin/bash
# bench lines of the other SURVEY.md 8(d) workloads + a kernel trace and instruction counters for config 5 (run on a GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_cfg
mkdir -p $out
python3 bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline > $out/r02_bench_config2.json 2> $out/c2.err
python3 bench.py --config 5 --steps 10 --warmup 3 > $out/r02_bench_config5.json 2> $out/c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats5 -o b -- python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/c5_stats.json 2> $out/c5_stats.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/pmc5 -o b -- python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/c5_pmc.json 2> $out/c5_pmc.err
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/prof_cfg"
st = glob.glob(out + "/stats5/**/*kernel_stats.csv", recursive=True)
with open(out + "/r02_cfg5_kernel_stats.csv", "w", newline="") as f:
    wr = csv.writer(f)
    wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for r in csv.DictReader(open(st[0])):
        wr.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(out + "/pmc5/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/r02_cfg5_pmc_counters.csv", "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            f.write(f"{k},{c},{sum(v) / len(v):.4e},{len(v)}\n")
print(open(out + "/r02_cfg5_pmc_counters.csv").read())
PY
