#!/bin/bash
# What the fabric-side requests of the two hot kernels are made of (L2 -> fabric read/write requests by size and target, L2 hits
# and misses, instruction- and scalar-cache requests), under the bench command: tools/pmc_tcc_breakdown.sh OUT [ENV=VAL ...]
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
k=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_IO_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_GMI_32B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_WRREQ_WRITE_IO_32B_sum" \
           "TCC_EA0_WRREQ_ATOMIC_DRAM_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
           "SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_ICACHE_MISSES SQC_DCACHE_MISSES"; do
  k=$((k + 1))
  ( export "$@" DUMMY_=1; rocprofv3 --pmc $set --output-format csv -d $out/p$k -o b -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $out/p$k.json 2> $out/p$k.err ) || echo "pass $k failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = ("extend_full" if ", 0>" in r["Kernel_Name"] else "extend") if "ext_kernel" in r["Kernel_Name"] else ("swalign2" if "swp_kernel" in r["Kernel_Name"] else None)
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"{k},{c},{sum(v) / len(v):.1f},{len(v)}")
PY
