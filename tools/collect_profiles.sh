#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 summaries of `python bench.py` for profiles/.
#   tools/collect_profiles.sh <tag>     -> gpurun_out/prof_<tag>/summary/...   (CSV / JSON)
# The kernel trace is taken under the bench command itself.  Counter passes are separate runs with --pmc only (no trace domains), as
# the pool requires; rocprofv3 runs one dispatch at a time while it collects counters, and the rescue path's kernel is RESIDENT (one
# launch per epoch of the submission ring, csrc/bpsw_ring.h) -- under the mixed workload every extension launch would wait for an epoch
# to idle out -- so the counter passes take the two boundaries one at a time (BENCH_ONLY=ext / grp: the same batches, the same
# kernels, per-launch and per-batch counts are properties of the kernels and their inputs).  BPSW_RING_LONE_LAUNCH=0 in the rescue pass:
# with the dispatches serialised a batch often finds itself alone, and the library would give it a launch of its own -- the bench's mixed
# workload, which these counts are for, never does.
set -e
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 > $out/bench_under_stats.json 2> $out/stats.err
pmc() {  # name, counters...
  name=$1; shift
  BENCH_ONLY=ext rocprofv3 --pmc "$@" --output-format csv -d $out/$name/ext -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2> $out/$name.ext.err
  BENCH_ONLY=grp BPSW_RING_CAPACITY=4096 BPSW_RING_LONE_LAUNCH=0 rocprofv3 --pmc "$@" --output-format csv -d $out/$name/grp -o bench -- python bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2> $out/$name.grp.err
  echo "pmc pass $name done"
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
# request sizes: FETCH_SIZE tallies every L2->fabric read request at 64 B, so its bytes are only right when the request mix is known
pmc rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pmc wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_IO_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum
pmc rdsrc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_IO_32B_sum SQC_TC_INST_REQ SQC_ICACHE_MISSES
pmc sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_LDS
# where the wave-cycles go: active / issue-stalled / parked, per kernel
pmc sqact SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
# the raw per-dispatch CSVs run to hundreds of MB: summarise here, keep the summaries and the kernel stats only
# (PROF_SW_BATCHES: the rescue batches one counter pass submits -- 2 steps x 16 passes x 256 groups)
PROFILES_OUT=$out/summary PROF_SW_BATCHES=8192 python3 tools/summarize_profiles.py $out $tag > $out/summarize.log 2>&1 || true
cp $out/stats/*kernel_stats.csv $out/summary/ 2>/dev/null || true
rm -rf $out/stats $out/fetch $out/write $out/rdreq $out/wrreq $out/rdsrc $out/sq $out/sqact
ls -la $out/summary
