# usage: tools/ring_sweep.sh "label env..." ...   (each argument one run of bench.py, 10 steps / 3 warm-up, no extras)
for spec in "$@"; do
  set -- $spec
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/b_$label.json 2> gpurun_out/b_$label.err
  python - "$label" <<PY
import json,sys
f="gpurun_out/b_%s.json"%sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1]); h=d["host"]["phase_ms_per_call"]
    print("%-14s"%sys.argv[1], "value %.3e"%d["value"], "ms/step %.1f"%d["ms_per_step"], "cpus %.1f"%d["host"]["cpus_busy"], "ext dev %.2f wait %.2f"%(h.get("extend",{}).get("dev",0), h.get("extend",{}).get("wait",0)), "grp dev %.2f"%h.get("matesw_group",{}).get("dev",0), flush=True)
except Exception as e: print(f, "ERR", e)
PY
done
