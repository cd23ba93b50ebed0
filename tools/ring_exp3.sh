run() { # label, steps, warmup, env...
  label=$1; st=$2; wu=$3; shift; shift; shift
  env "$@" timeout -k 10 300 python bench.py --steps $st --warmup $wu --no-extras --no-cpu-baseline > gpurun_out/b_$label.json 2> gpurun_out/b_$label.err
  python - "$label" <<PY
import json,sys
f="gpurun_out/b_%s.json"%sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1]); h=d["host"]["phase_ms_per_call"]
    print(sys.argv[1], "value %.3e"%d["value"], "ms/step %.1f"%d["ms_per_step"], "cpus %.1f"%d["host"]["cpus_busy"], "ext dev %.2f"%h.get("extend",{}).get("dev",0), "grp dev %.2f"%h.get("matesw_group",{}).get("dev",0))
except Exception as e: print(f, "ERR", e)
PY
}
run r0_10_3 10 3 BPSW_RING=0
run r0_4_1 4 1 BPSW_RING=0
run r1_10_3 10 3 BPSW_RING=1
run r1w2_10_3 10 3 BPSW_RING=1 BPSW_RING_WG_PER_CU=2
