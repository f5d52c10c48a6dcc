#!/bin/bash
# Diagnostic: the same launches under two builds of the library on ONE box, alternating (A = BSIG_LIB_PATH=$1, B = the
# tree's own build): north star's step, config 2, config 5's share, coverage and count on config 3's tiling.
# Usage (through gpurun): bash scripts/ab_builds.sh bamsignals_amd/libbamsignals_hip_prev.so [rounds]
A=$(realpath "$1"); ROUNDS=${2:-2}
one() {  # label, env...
    local label=$1; shift
    for c in ns c2 c5 C3 C4 count; do
        case $c in
        ns) out=$(env "$@" python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-e2e --no-also 2>/dev/null | tail -1) ;;
        c2) out=$(env "$@" python bench.py --config C2 --steps 64 --warmup 16 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1) ;;
        c5) out=$(env "$@" python bench.py --config C5 --steps 32 --warmup 8 --no-cpu-baseline --no-e2e --no-also 2>/dev/null | tail -1) ;;
        *)  out=$(env "$@" python scripts/profile_case.py $c 2>/dev/null | tail -1) ;;
        esac
        echo "$out" | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
ms = d.get('kernel_ms') or d['roofline']['kernel_ms']
print('$label', '$c', 'kernel_ms %.4f' % ms, 'step_ms', d.get('ms_per_step'))"
    done
}
for r in $(seq $ROUNDS); do
    one A BSIG_LIB_PATH=$A
    one B BSIG_AB=1
done
