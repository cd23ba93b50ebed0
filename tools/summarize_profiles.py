#!/usr/bin/env python3
"""Turns the CSVs of tools/collect_profiles.sh into the summaries committed under profiles/:
   python tools/summarize_profiles.py gpurun_out/prof_<tag> <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def short(name):
    for k, v in (("ext_kernel", "extend"), ("ext_prepass", "ext_prepass"), ("swp_kernel", "swalign2"), ("sw4_kernel", "swalign2"), ("sw_kernel", "swalign2"), ("sw_prepass", "sw_prepass"),
                 ("reg2aln", "reg2aln"), ("chain2aln", "chain2aln"), ("ext_qt", "extend_qt"), ("global_kernel", "global")):
        if k in name:
            return v
    return name[:40]


stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(P, f"{tag}_kernel_stats_bench.csv"), "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for r in rows:
            f.write(f"{r['Name']},{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['Percentage']}\n")
    print("kernel stats:", [(short(r["Name"]), r["Calls"], r["AverageNs"]) for r in rows[:4]])

per = defaultdict(lambda: defaultdict(list))
for sub in ("fetch", "write", "sq"):
    for fn in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(P, f"{tag}_pmc_counters.csv"), "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for k in sorted(per):
        for c in sorted(per[k]):
            v = per[k][c]
            f.write(f"{k},{c},{sum(v) / len(v):.4e},{len(v)}\n")
traffic = {}
for k in ("extend", "swalign2"):
    if "FETCH_SIZE" in per[k] and "WRITE_SIZE" in per[k]:
        fe = sum(per[k]["FETCH_SIZE"]) / len(per[k]["FETCH_SIZE"])
        wr = sum(per[k]["WRITE_SIZE"]) / len(per[k]["WRITE_SIZE"])
        traffic[k] = int((2 * fe + wr) * 1024)   # KB -> bytes; x2 on FETCH_SIZE: gfx950 correction, MI355X_MICROARCH.md
traffic["note"] = ("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes "
                   f"(profiles/{tag}_pmc_counters.csv); the x2 on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md")
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(traffic)

issue = {}
for k in ("extend", "swalign2", "ext_prepass", "sw_prepass"):
    if "SQ_INSTS_VALU" in per[k] and "SQ_INSTS_SALU" in per[k]:
        issue[k] = {"valu": int(sum(per[k]["SQ_INSTS_VALU"]) / len(per[k]["SQ_INSTS_VALU"])),
                    "salu": int(sum(per[k]["SQ_INSTS_SALU"]) / len(per[k]["SQ_INSTS_SALU"]))}
issue["note"] = f"wave-instructions per launch (SQ_INSTS_VALU, SQ_INSTS_SALU) from profiles/{tag}_pmc_counters.csv"
json.dump(issue, open(os.path.join(P, "pmc_issue.json"), "w"), indent=1)
print(issue)
