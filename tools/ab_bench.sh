#!/bin/bash
# A/B runs of bench.py on a GPU box: tools/ab_bench.sh OUTFILE "ENV1" "ENV2" ... -- [bench args]
# each ENV is a space-separated list of VAR=value (or "-" for none); prints one short line per run
out=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
for e in "${envs[@]}"; do
  if [ "$e" = "-" ]; then e=""; fi
  line=$(env $e python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | tail -1)
  echo "$e | $@ | $(python3 -c "
import json,sys
d=json.loads(sys.argv[1])
print('value %.4g %s ms/step %.2f' % (d['value'], d.get('unit',''), d['ms_per_step']), 'verified', d.get('verified'))
" "$line" 2>/dev/null || echo "FAILED: $line")" | tee -a $out
done
