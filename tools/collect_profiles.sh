#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 summaries of `python bench.py` for profiles/.
#   tools/collect_profiles.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write,sq}/...   (CSV)
# Counter passes are separate runs with --pmc only (no trace domains), as the pool requires.
set -e
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 > $out/bench_under_stats.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2> $out/write.err
# request sizes: FETCH_SIZE tallies every L2->fabric read request at 64 B, so its bytes are only right when the request mix is known
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/rdreq -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2> $out/rdreq.err
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_IO_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum --output-format csv -d $out/wrreq -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2> $out/wrreq.err
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_IO_32B_sum SQC_TC_INST_REQ SQC_ICACHE_MISSES --output-format csv -d $out/rdsrc -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2> $out/rdsrc.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d $out/sq -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2> $out/sq.err
# where the wave-cycles go: active / issue-stalled / parked, per kernel, under the bench's own sharing of the device
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/sqact -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2> $out/sqact.err
# the raw per-dispatch CSVs of a 12-pass step run to hundreds of MB: summarise here, keep the summaries and the kernel stats only
PROFILES_OUT=$out/summary python3 tools/summarize_profiles.py $out $tag > $out/summarize.log 2>&1 || true
cp $out/stats/*kernel_stats.csv $out/summary/ 2>/dev/null || true
rm -rf $out/stats $out/fetch $out/write $out/rdreq $out/wrreq $out/rdsrc $out/sq $out/sqact
ls -la $out/summary
