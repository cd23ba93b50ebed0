#!/bin/bash
# bench lines of the other SURVEY.md 8(d) workloads + a kernel trace and PER-KERNEL instruction / request counters for config 5
# (run on a GPU box):  tools/collect_cfg.sh [tag]   ->  gpurun_out/prof_cfg/<tag>_*
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_cfg
mkdir -p $out
python3 bench.py --config 2 --steps 6 --warmup 2 --no-cpu-baseline > $out/${tag}_bench_config2.json 2> $out/c2.err
python3 bench.py --config 5 --steps 6 --warmup 2 > $out/${tag}_bench_config5.json 2> $out/c5.err
# each boundary of config 5 alone: the GCUPS of each kernel without the other sharing the device
BENCH_ONLY=ext python3 bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $out/${tag}_bench_config5_ext_only.json 2> $out/c5e.err
BENCH_ONLY=grp python3 bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $out/${tag}_bench_config5_grp_only.json 2> $out/c5g.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats5 -o b -- python3 bench.py --config 5 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $out/c5_stats.json 2> $out/c5_stats.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/pmc5 -o b -- python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/c5_pmc.json 2> $out/c5_pmc.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/rdreq5 -o b -- python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/c5_rdreq.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write5 -o b -- python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $out/c5_write.err
python3 - "$tag" <<'PY'
import csv, glob, collections, re, sys
tag = sys.argv[1]
out = "gpurun_out/prof_cfg"
def short(n):   # the kernel's own name with its template arguments (a cut at the first "(" lands inside "(anonymous namespace)")
    m = re.search(r"(\w+_kernel)<([^>]*)>", n)
    if m:
        return m.group(1) + "<" + m.group(2).replace(" ", "") + ">"
    m = re.search(r"(\w+_kernel)\b", n)
    return m.group(1) if m else n[:48]
st = glob.glob(out + "/stats5/**/*kernel_stats.csv", recursive=True)
with open(out + f"/{tag}_cfg5_kernel_stats.csv", "w", newline="") as f:
    wr = csv.writer(f)
    wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for r in csv.DictReader(open(st[0])):
        wr.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc5", "rdreq5", "write5"):
    for fn in glob.glob(out + f"/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + f"/{tag}_cfg5_pmc_counters.csv", "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            f.write(f"\"{k}\",{c},{sum(v) / len(v):.4e},{len(v)}\n")
print(open(out + f"/{tag}_cfg5_pmc_counters.csv").read())
PY
