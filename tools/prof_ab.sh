#!/bin/bash
# Kernel trace + SQ instruction counters of one bench invocation on a GPU box, per kernel, as a compact table:
#   [ENV...] tools/prof_ab.sh OUTDIR [bench args]      (separate --pmc pass, as the pool requires)
set -e
out=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/$out
cd /tmp && export TMPDIR=/tmp && cd $repo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o b -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $out/bench_stats.json 2> $out/stats.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/sq -o b -- python3 bench.py "$@" --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $out/bench_sq.json 2> $out/sq.err
python3 - "$out" <<'PY'
import csv, glob, re, sys, collections, json
out = sys.argv[1]
def short(n):
    m = re.search(r"(\w+_kernel)<([^>]*)>", n)
    return (m.group(1) + "<" + m.group(2).replace(" ", "") + ">") if m else n.split("(")[0][-40:]
rows = []
for fn in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), r["Percentage"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(out + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as f:
    try:
        d = json.loads(open(out + "/bench_stats.json").read().strip().split("\n")[-1])
        f.write("bench under the kernel trace: value %.4g %s, ms_per_step %.2f\n" % (d["value"], d["unit"], d["ms_per_step"]))
    except Exception as e:
        f.write("bench line unreadable: %r\n" % (e,))
    f.write("kernel | calls | avg_us | total_ms | pct || per launch: VALU | SALU | waves | busy_cycles\n")
    for k, c, tot, avg, pct in sorted(rows, key=lambda r: -r[2])[:12]:
        a = acc.get(k, {})
        mean = lambda key: (sum(a[key]) / len(a[key])) if key in a else float("nan")
        f.write("%s | %d | %.1f | %.1f | %s || %.4g | %.4g | %.4g | %.4g\n" % (k, c, avg / 1e3, tot / 1e6, pct, mean("SQ_INSTS_VALU"), mean("SQ_INSTS_SALU"), mean("SQ_WAVES"), mean("SQ_BUSY_CYCLES")))
print(open(out + "/summary.txt").read())
PY
