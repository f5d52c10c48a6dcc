#!/bin/bash
# Runs on the GPU box (through gpurun): instruction counters of the pileup kernels -- SQ_INSTS_VALU / SALU / LDS /
# VMEM_RD and SQ_WAVES per launch (one PMC pass, kernel trace only) for the north star's step, config 3's coverage,
# config 4's call and bamCount on config 3's tiling.  scripts/summarize_insts.py turns the CSVs into a table.
# Usage: scripts/pmc_insts.sh r04
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
C="--kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv"
timeout 400 rocprofv3 $C -d "$OUT/${TAG}_insts_ns" -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-also > "$OUT/${TAG}_insts_ns.log" 2>&1 || exit 1
for c in C3 C4 count bins; do
    timeout 400 rocprofv3 $C -d "$OUT/${TAG}_insts_$c" -- python3 $R/scripts/profile_case.py $c > "$OUT/${TAG}_insts_$c.log" 2>&1 || exit 1
done
echo "[insts] done"
