#!/bin/bash
# SQ activity / stall counters of LONE launches of each hot kernel (the device to itself), next to the passes under the bench command
# (tools/collect_profiles.sh):   tools/pmc_lone_kernels.sh OUTDIR      (run on a GPU box)
#   ext rows   : tools/ext_row_cost.py 100 (every shortcut off, every row swept) at 8 persistent workgroups per CU = 8 waves per SIMD
#   ext batch  : tools/ext_batch_instr.py 63 3 (a bench-shaped configs[2] batch: sift kernel + extension kernel, default grids)
#   swalign2   : tools/sw_kernel_time.py 57664 (a saturating batch of rescue jobs)
set -e
out=$1; mkdir -p $out; out=$(cd $out && pwd)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_IFETCH SQ_WAVES"
run() {  # tag, pass, counters, command...
  local tag=$1 pass=$2 ctr=$3; shift 3
  rocprofv3 --pmc $ctr --output-format csv -d $out/$tag/$pass -o r -- "$@" > $out/$tag.$pass.log 2>&1
}
export BPSW_EXT_SHORT_BLOCKS_PER_CU=8
run ext_rows a "$A" python3 $root/tools/ext_row_cost.py 100 0.05 0.01 0 65536 3
run ext_rows b "$B" python3 $root/tools/ext_row_cost.py 100 0.05 0.01 0 65536 3
unset BPSW_EXT_SHORT_BLOCKS_PER_CU
run ext_batch a "$A" python3 $root/tools/ext_batch_instr.py 63 3
run ext_batch b "$B" python3 $root/tools/ext_batch_instr.py 63 3
run swalign2 a "$A" python3 $root/tools/sw_kernel_time.py 57664
run swalign2 b "$B" python3 $root/tools/sw_kernel_time.py 57664
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
print("# lone launches (the device to itself): fractions of SQ_WAVE_CYCLES (quad-cycles) and wave-instructions per launch")
for tag in ("ext_rows", "ext_batch", "swalign2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
            if m and ("ext_" in m.group(1) or "swp_" in m.group(1)):
                acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        mean = {c: sum(x) / len(x) for c, x in v.items()}
        wc = mean.get("SQ_WAVE_CYCLES")
        if not wc or mean.get("SQ_INSTS_VALU", 0) + mean.get("SQ_INSTS_SALU", 0) < 1000:
            continue
        fr = {c.replace("SQ_", "").lower(): round(mean[c] / wc, 3) for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS",
                                                                       "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY") if c in mean}
        ins = {c.replace("SQ_INSTS_", "").lower(): round(mean[c] / 1e6, 2) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS") if c in mean}
        print(f"{tag:10s} {k:28s} launches {len(v['SQ_WAVE_CYCLES'])}  {fr}  M-instr/launch {ins}")
PY
