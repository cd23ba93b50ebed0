for v in base noprio noacq; do
  if [ $v = base ]; then unset BPSW_LIB; else export BPSW_LIB=$PWD/cloud-scale-bwamem_amd/lib_exp/libbPSW_hip_$v.so; fi
  timeout -k 10 200 python tools/ring_probe.py > gpurun_out/probe_$v.txt 2>&1
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err
done
grep -h "n=   428\|n=  2000" gpurun_out/probe_base.txt gpurun_out/probe_noprio.txt gpurun_out/probe_noacq.txt
python - <<PY
import json
for v in ("base","noprio","noacq"):
    f="gpurun_out/bench_%s.json"%v
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(v, d["value"], d["ms_per_step"], d.get("host",{}).get("cpus_busy")); print(d["host"]["phase_ms_per_call"])
    except Exception as e: print(f, "ERR", e)
PY
