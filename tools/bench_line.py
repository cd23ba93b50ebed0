#!/usr/bin/env python3
"""one-line digest of bench.py JSON logs: python tools/bench_line.py LOG..."""
import json
import sys

for p in sys.argv[1:]:
    try:
        line = [l for l in open(p) if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:  # noqa: BLE001
        print(p, "unreadable:", e)
        continue
    k, h = d["kernels"], d["host"]["call_ms"]
    print(f"{p}: T={d['config']['host_threads_per_gpu']} value={d['value'] / 1e6:.1f}M ms/step={d['ms_per_step']:.2f} cpus={d['host'].get('cpus_busy')} "
          f"ext call={h['extend']['mean'] if h['extend'] else 0} (k {k['extend']['avg_ms']} h2d {k['extend']['h2d_ms_avg']} d2h {k['extend']['d2h_ms_avg']}) "
          f"grp call={h['matesw_group']['mean'] if h['matesw_group'] else 0} (k {k['swalign2']['avg_ms']} h2d {k['swalign2']['h2d_ms_avg']} d2h {k['swalign2']['d2h_ms_avg']}) "
          f"phases={d['host'].get('phase_ms_per_call')} pcie={d['host']['pcie_GBps']}GB/s verified={d.get('verified')}")
