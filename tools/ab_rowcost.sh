#!/bin/bash
# A/B of two library builds on the row-cost workload (every shortcut off) at several grid sizes:
#   tools/ab_rowcost.sh OUT "L..." "blocks_per_cu..." lib_a lib_b ...     (lib = path, or "default")
out=$1; Ls=$2; bpcs=$3; shift 3
mkdir -p $out
for L in $Ls; do for bpc in $bpcs; do for lib in "$@"; do
  if [ "$lib" = default ]; then unset BPSW_LIB; else export BPSW_LIB=$GRAFT_REPO_ROOT/$lib; fi
  BPSW_EXT_SHORT_BLOCKS_PER_CU=$bpc BPSW_EXT_BLOCKS_PER_CU=$bpc python3 tools/ext_row_cost.py $L ${SUB:-0.05} ${INDEL:-0.01} ${MASK:-0} ${NTASK:-65536} 5 2>/dev/null | grep "^{" | sed "s|^|L=$L bpc=$bpc lib=$(basename $lib) |"
done; done; done | tee $out/ab.txt
