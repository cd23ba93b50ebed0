#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 summaries of `python bench.py` for profiles/.
#   tools/collect_profiles.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write,sq}/...   (CSV)
# Counter passes are separate runs with --pmc only (no trace domains), as the pool requires.
set -e
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/bench_under_stats.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $out/write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/sq -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $out/sq.err
find $out -name "*.csv" | head -20
