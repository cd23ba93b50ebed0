# usage: tools/ring_sweep_cfg.sh CONFIG "label env..." ...
cfg=$1; shift
for spec in "$@"; do
  set -- $spec
  label=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config $cfg --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/c${cfg}_$label.json 2> gpurun_out/c${cfg}_$label.err
  python - "c${cfg}_$label" <<PY
import json,sys
f="gpurun_out/%s.json"%sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1]); h=d["host"]["phase_ms_per_call"]
    print("%-14s"%sys.argv[1], "value %.3e"%d["value"], "ms/step %.1f"%d["ms_per_step"], "cpus %.1f"%d["host"]["cpus_busy"], "ext dev %.2f wait %.2f"%(h.get("extend",{}).get("dev",0), h.get("extend",{}).get("wait",0)), "grp dev %.2f"%h.get("matesw_group",{}).get("dev",0), "cpu/call", d["host"].get("call_cpu_ms"), flush=True)
except Exception as e: print(f, "ERR", e)
PY
done
