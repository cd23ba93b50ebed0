#!/usr/bin/env python3
"""Turns the CSVs of tools/collect_profiles.sh into the summaries committed under profiles/:
   python tools/summarize_profiles.py gpurun_out/prof_<tag> <tag>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.environ.get("PROFILES_OUT") or os.path.join(ROOT, "profiles")   # PROFILES_OUT: summarise on the GPU box (the raw CSVs of a 12-pass step are too large to pull back)
os.makedirs(P, exist_ok=True)


def short(name):
    m = re.search(r"ext_kernel<\w+, (\d)>", name)   # <COORD, SHORT>: 0 = the full kernel (listed / deferred tasks), 1 = the short kernel
    if m:
        return "extend_full" if m.group(1) == "0" else "extend"
    for k, v in (("swp_resident", "swalign2_resident"), ("ext_sift", "extend_sift"), ("ext_kernel", "extend"), ("ext_prepass", "ext_prepass"), ("swp_kernel", "swalign2"), ("sw4_kernel", "swalign2"), ("sw_kernel", "swalign2"), ("sw_prepass", "sw_prepass"),
                 ("reg2aln", "reg2aln"), ("chain2aln", "chain2aln"), ("global_kernel", "global")):
        if k in name:
            return v
    return name[:40]


stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(P, f"{tag}_kernel_stats_bench.csv"), "w", newline="") as f:
        wr = csv.writer(f)   # kernel names contain commas: quoted
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows:
            wr.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
    print("kernel stats:", [(short(r["Name"]), r["Calls"], r["AverageNs"]) for r in rows[:4]])

per = defaultdict(lambda: defaultdict(list))
for sub in ("fetch", "write", "sq", "sqact", "rdreq", "wrreq", "rdsrc"):
    for fn in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(P, f"{tag}_pmc_counters.csv"), "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for k in sorted(per):
        for c in sorted(per[k]):
            v = per[k][c]
            f.write(f"{k},{c},{sum(v) / len(v):.4e},{len(v)}\n")
# The resident rescue kernel (one launch per epoch of the submission ring) serves many batches per launch: what the other kernels have
# per launch, it has per BATCH -- the epochs' counters summed over the run and divided by the batches the run submitted
# (PROF_SW_BATCHES: tools/collect_profiles.sh passes what its bench commands submit).  "swalign2" below is that per-batch figure.
SW_BATCHES = int(os.environ.get("PROF_SW_BATCHES", "0"))
if "swalign2_resident" in per and SW_BATCHES > 0 and "swalign2" not in per:
    for c, v in per["swalign2_resident"].items():
        per["swalign2"][c] = [sum(v) / SW_BATCHES]
    with open(os.path.join(P, f"{tag}_pmc_counters.csv"), "a") as f:
        for c in sorted(per["swalign2"]):
            f.write(f"swalign2 (per batch: swalign2_resident summed over its {len(per['swalign2_resident'][c])} launches / {SW_BATCHES} batches),{c},{per['swalign2'][c][0]:.4e},{SW_BATCHES}\n")


def mean(k, c):
    v = per[k].get(c)
    return sum(v) / len(v) if v else None


traffic, detail = {}, {}
for k in ("extend", "swalign2"):
    fe, wr = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
    if fe is None or wr is None:
        continue
    d = {"fetch_size_x2_plus_write_size": int((2 * fe + wr) * 1024)}   # the blanket gfx950 correction of MI355X_MICROARCH.md
    r32, r64, r128 = mean(k, "TCC_EA0_RDREQ_32B_sum"), mean(k, "TCC_EA0_RDREQ_64B_sum"), mean(k, "TCC_EA0_RDREQ_128B_sum")
    if r128 is not None:
        # FETCH_SIZE = RDREQ x 64 B (same guide): exact only when every request is 64 B.  By request size instead:
        rd = 32 * r32 + 64 * r64 + 128 * r128
        d.update({"read_requests": {"32B": round(r32, 1), "64B": round(r64, 1), "128B": round(r128, 1)}, "read_bytes": int(rd),
                  "write_bytes": int(wr * 1024)})
        for c, name in (("TCC_EA0_RDREQ_IO_32B_sum", "read_32B_units_from_host_memory"), ("TCC_EA0_RDREQ_DRAM_sum", "read_requests_to_hbm"),
                        ("SQC_TC_INST_REQ", "instruction_cache_requests_to_l2"), ("SQC_ICACHE_MISSES", "instruction_cache_misses"),
                        ("TCC_EA0_WRREQ_WRITE_IO_32B_sum", "write_32B_units_to_host_memory"), ("TCC_EA0_WRREQ_ATOMIC_DRAM_sum", "atomic_requests")):
            if mean(k, c) is not None:
                d[name] = round(mean(k, c), 1)
        traffic[k] = int(rd + wr * 1024)
    else:
        traffic[k] = d["fetch_size_x2_plus_write_size"]
    detail[k] = d
# an extension CALL launches the sift kernel in front of the extension kernel and the full kernel behind it (bpsw_extend.hip): the
# call's traffic is the sum (the sift kernel reads the whole wire batch once more), scaled to launches of the extension kernel
def _bytes(k):
    r32, r64, r128, wr = mean(k, "TCC_EA0_RDREQ_32B_sum"), mean(k, "TCC_EA0_RDREQ_64B_sum"), mean(k, "TCC_EA0_RDREQ_128B_sum"), mean(k, "WRITE_SIZE")
    if None in (r32, r64, r128, wr):
        return None
    return 32 * r32 + 64 * r64 + 128 * r128 + wr * 1024
if "extend" in traffic and _bytes("extend") is not None:
    n_ext = len(per["extend"]["WRITE_SIZE"])
    parts = {"extension_kernel": int(_bytes("extend"))}
    for k, name in (("extend_sift", "sift_kernel"), ("extend_full", "full_kernel")):
        if _bytes(k) is not None:
            parts[name] = int(_bytes(k) * len(per[k]["WRITE_SIZE"]) / n_ext)
    detail["extend"]["per_call_by_kernel"] = parts
    traffic["extend"] = int(sum(parts.values()))
traffic["detail"] = detail
traffic["note"] = ("fabric-side bytes per launch from separate rocprofv3 --pmc passes under the bench command "
                   f"(profiles/{tag}_pmc_counters.csv).  Reads by request size, 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B: FETCH_SIZE "
                   "counts every request as 64 B (MI355X_MICROARCH.md), so doubling it is exact only for all-128-byte request streams; "
                   "these kernels issue 45-55 % 64-byte requests.  Writes = WRITE_SIZE.  The blanket 2*FETCH_SIZE + WRITE_SIZE figure is "
                   "kept in detail.  Reads of pinned host memory (zero-copy inputs) and instruction fetch are included in both.")
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(traffic)

issue = {}
for k in ("extend", "swalign2", "ext_prepass", "sw_prepass"):
    if "SQ_INSTS_VALU" in per[k] and "SQ_INSTS_SALU" in per[k]:
        issue[k] = {"valu": int(sum(per[k]["SQ_INSTS_VALU"]) / len(per[k]["SQ_INSTS_VALU"])),
                    "salu": int(sum(per[k]["SQ_INSTS_SALU"]) / len(per[k]["SQ_INSTS_SALU"]))}
# per bpsw_extend_batch CALL: the 48-VGPR launch plus the sift kernel in front of it and the full-kernel launch behind it (listed /
# deferred tasks), when there are any
if "extend" in issue and ("SQ_INSTS_VALU" in per["extend_full"] or "SQ_INSTS_VALU" in per["extend_sift"]):
    calls = len(per["extend"]["SQ_INSTS_VALU"])
    tot = lambda c: sum(sum(per[k].get(c, [])) for k in ("extend", "extend_full", "extend_sift"))
    issue["extend_per_call"] = {"valu": int(tot("SQ_INSTS_VALU") / calls), "salu": int(tot("SQ_INSTS_SALU") / calls)}
    if "SQ_INSTS_VALU" in per["extend_sift"]:
        issue["extend_sift"] = {"valu": int(sum(per["extend_sift"]["SQ_INSTS_VALU"]) / len(per["extend_sift"]["SQ_INSTS_VALU"])),
                                "salu": int(sum(per["extend_sift"]["SQ_INSTS_SALU"]) / len(per["extend_sift"]["SQ_INSTS_SALU"]))}
issue["note"] = f"wave-instructions per launch (SQ_INSTS_VALU, SQ_INSTS_SALU) from profiles/{tag}_pmc_counters.csv; extend_per_call = all extension launches of a call (sift + 48-VGPR + full kernel)"
json.dump(issue, open(os.path.join(P, "pmc_issue.json"), "w"), indent=1)
print(issue)

# where the wave-cycles of each kernel go under the bench's own sharing of the device (SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_*
# count quad-cycles, MI355X_MICROARCH.md; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES): fractions of the wave-cycles
act = {}
for k in ("extend", "extend_sift", "extend_full", "swalign2"):
    wc = mean(k, "SQ_WAVE_CYCLES")
    if not wc:
        continue
    act[k] = {name: round(mean(k, c) / wc, 4) for c, name in (("SQ_ACTIVE_INST_ANY", "active_any"), ("SQ_ACTIVE_INST_VALU", "active_valu"),
                                                              ("SQ_ACTIVE_INST_SCA", "active_sca"), ("SQ_ACTIVE_INST_LDS", "active_lds"),
                                                              ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_WAIT_INST_LDS", "wait_inst_lds"),
                                                              ("SQ_WAIT_ANY", "wait_any")) if mean(k, c) is not None}
    act[k]["wave_quad_cycles_per_launch"] = int(wc)
    act[k]["launches"] = len(per[k]["SQ_WAVE_CYCLES"])
if act:
    act["note"] = ("fractions of SQ_WAVE_CYCLES per kernel under the bench command (rocprofv3 --pmc, one pass; the launches share the device "
                   "with each other as in the timed region): active = an instruction of the wave is issuing, wait_inst_any = ready but not "
                   "issued (the pipe or the arbiter is busy), wait_any = parked (dependencies, waitcnt, instruction fetch)")
    json.dump(act, open(os.path.join(P, f"{tag}_sq_activity.json"), "w"), indent=1)
    with open(os.path.join(P, f"{tag}_sq_activity.csv"), "w") as f:
        f.write("kernel,quantity,value\n")
        for k, d in act.items():
            if isinstance(d, dict):
                for q, v in d.items():
                    f.write(f"{k},{q},{v}\n")
    print(act)
