#!/bin/bash
# like ab_bench.sh, but prints the per-call host phases of each run:  tools/ab_phases.sh OUTFILE "ENV1" "ENV2" ... -- [bench args]
out=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
for e in "${envs[@]}"; do
  if [ "$e" = "-" ]; then e=""; fi
  line=$(env $e python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
  echo "$e | $(python3 -c "
import json,sys
d=json.loads(sys.argv[1]); h=d['host']
print('value %.4g ms/step %.2f cpus %.1f' % (d['value'], d['ms_per_step'], h['cpus_busy']), 'call_ms', {k: v['mean'] for k, v in h['call_ms'].items()}, 'phases', h['phase_ms_per_call'])
" "$line" 2>/dev/null || echo "FAILED: $line")" | tee -a $out
done
