#!/bin/bash
# SQ activity counters of the two hot kernels under the bench command: tools/pmc_sq_quick.sh OUT
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $out/a -o b -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $out/a.json 2> $out/a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $out/b -o b -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $out/b.json 2> $out/b.err || true
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = ("extend_full" if ", 0>" in r["Kernel_Name"] else "extend") if "ext_kernel" in r["Kernel_Name"] else ("swalign2" if "swp_kernel" in r["Kernel_Name"] else None)
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, {c: f"{x:.3e}" for c, x in m.items()})
    if "SQ_WAVE_CYCLES" in m:
        print("   per wave-cycle: active_inst_any", round(m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 3), "wait_inst_any", round(m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 3),
              "wait_any", round(m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 3), "valu", round(m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_WAVE_CYCLES"], 3),
              "sca", round(m.get("SQ_ACTIVE_INST_SCA", 0) / m["SQ_WAVE_CYCLES"], 3), "lds", round(m.get("SQ_ACTIVE_INST_LDS", 0) / m["SQ_WAVE_CYCLES"], 3))
PY
