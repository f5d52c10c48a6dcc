#!/bin/bash
# Runs on the GPU box (through gpurun): the default bench, the rocprofv3 kernel trace of the same
# command, and the two PMC passes (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950).
# Usage: scripts/profile_round.sh r01        -> gpurun_out/r01_*
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
cd $R
timeout 900 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 64 --warmup 16 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $R/bench.py $ARGS > $OUT/${TAG}_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $R/bench.py $ARGS > $OUT/${TAG}_pmc_write.log 2>&1
# the device-side decode kernels (whole file + index-driven) on the same 5e7-read BAM
rm -rf $OUT/${TAG}_decode_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_decode_trace -- python3 $R/scripts/decode_device_time.py > $OUT/${TAG}_decode_trace.log 2>&1
ls $OUT | grep $TAG
