#!/bin/bash
# Runs on the GPU box (through gpurun): for every BASELINE kernel the un-profiled timing, the
# rocprofv3 kernel trace of the same command and the two PMC passes (FETCH_SIZE and WRITE_SIZE need
# separate passes on gfx950; counters are never combined with other trace domains).  Nothing is
# deleted: scripts/summarize_profile.py takes the newest output of every case.
# Usage: scripts/profile_round.sh r03 [cases...]    -> gpurun_out/r03_*   (then scripts/summarize_profile.py r03)
set -u
TAG=${1:-r03}
shift || true
CASES=${*:-ns c2 c3 c4 c5 count bins decode decodereal}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_case() {   # name, program + args (python3 script ...)
    local name=$1; shift
    timeout 900 "$@" > "$OUT/${TAG}_${name}_plain.json" 2> "$OUT/${TAG}_${name}_plain.err" || return 1
    echo "[$name] plain done"
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_${name}_trace" -- "$@" > "$OUT/${TAG}_${name}_trace.log" 2>&1 || return 1
    echo "[$name] trace done"
    timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_${name}_pmc_fetch" -- "$@" > "$OUT/${TAG}_${name}_pmc_fetch.log" 2>&1 || return 1
    timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_${name}_pmc_write" -- "$@" > "$OUT/${TAG}_${name}_pmc_write.log" 2>&1 || return 1
    echo "[$name] pmc done"
}
for c in $CASES; do
    case $c in
    ns)    run_case ns python3 $R/bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-e2e --no-also || exit 1 ;;
    c2)    run_case c2 python3 $R/bench.py --config C2 --steps 64 --warmup 16 --no-cpu-baseline --no-e2e || exit 1 ;;
    c5)    run_case c5 python3 $R/bench.py --config C5 --steps 32 --warmup 8 --no-cpu-baseline --no-e2e --no-also || exit 1 ;;
    c3)    run_case c3 python3 $R/scripts/profile_case.py C3 || exit 1 ;;
    c4)    run_case c4 python3 $R/scripts/profile_case.py C4 || exit 1 ;;
    count) run_case count python3 $R/scripts/profile_case.py count || exit 1 ;;
    bins)  run_case bins python3 $R/scripts/profile_case.py bins || exit 1 ;;
    t500)  run_case t500 python3 $R/scripts/profile_case.py t500 || exit 1 ;;
    t1000) run_case t1000 python3 $R/scripts/profile_case.py t1000 || exit 1 ;;
    decode)
        # the device-side decode kernels (whole file + index-driven) on the 5e7-read BAM
        timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_decode_trace" -- python3 $R/scripts/decode_device_time.py > "$OUT/${TAG}_decode_trace.log" 2>&1 || exit 1
        echo "[decode] trace done" ;;
    decodereal)
        # k_inflate on literal-heavy blocks: a BAM with read names, bases and qualities (2e7 reads, 2.1 GB)
        timeout 900 python3 $R/scripts/decode_realshaped_device_time.py > "$OUT/${TAG}_decodereal_plain.txt" 2>&1 || exit 1
        timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_decodereal_trace" -- python3 $R/scripts/decode_realshaped_device_time.py > "$OUT/${TAG}_decodereal_trace.log" 2>&1 || exit 1
        echo "[decodereal] trace done" ;;
    decodens)
        # the north star's file (5e8 bare reads, 3 GB, 327,000 blocks): three decodes in a row
        timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_decodens_trace" -- python3 $R/scripts/decode_ns_time.py 500000000 3 > "$OUT/${TAG}_decodens_trace.log" 2>&1 || exit 1
        echo "[decodens] trace done" ;;
    esac
done
ls "$OUT" | grep "${TAG}_" | head -50
