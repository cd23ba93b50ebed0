#!/bin/bash
# Where a DP row of the extension kernel spends its cycles: SQ activity / stall / instruction-fetch counters of a LONE launch of
# tools/ext_row_cost.py (every shortcut off, every row swept) at a chosen number of persistent workgroups per CU.
#   tools/pmc_row_stalls.sh OUTDIR L BLOCKS_PER_CU [NTASK]      (run on a GPU box; 8 blocks per CU = eight waves per SIMD)
set -e
out=$1; L=$2; bpc=$3; nt=${4:-32768}
SUB=${SUB:-0.05}; INDEL=${INDEL:-0.01}; MASK=${MASK:-0}
mkdir -p $out
out=$(cd $out && pwd)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export BPSW_EXT_SHORT_BLOCKS_PER_CU=$bpc BPSW_EXT_BLOCKS_PER_CU=$bpc
tag=L${L}_b${bpc}
pass() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $out/$tag/$name -o r -- python3 $root/tools/ext_row_cost.py $L $SUB $INDEL $MASK $nt 3 > $out/$tag.$name.log 2>&1
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC
pass c SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAVES SQ_CYCLES SQ_INSTS_LDS
python3 - "$out" "$tag" <<'PY'
import ast, csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
d = None
for fn in sorted(glob.glob(f"{out}/{tag}.*.log")):
    ls = [l for l in open(fn) if l.startswith("{")]
    if ls:
        d = ast.literal_eval(ls[-1])
tot, cnt = {}, {}
for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ext_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
            cnt[r["Counter_Name"]] = cnt.get(r["Counter_Name"], 0) + 1
print(tag, d)
rows = d["rows"] * d["launches"] if d else 1
print("  per DP row:", {k: round(v / rows, 2) for k, v in sorted(tot.items())})
wc = tot.get("SQ_WAVE_CYCLES")
if wc:
    print("  share of wave-cycles:", {k: round(tot[k] / wc, 3) for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS") if k in tot})
PY
