#!/bin/bash
# Diagnostic: L2 counters of k_inflate alone (scripts/inflate_bench.py's child under rocprofv3 --pmc, one pass per
# counter set) at several numbers of resident blocks: does the match loads' latency come from L2 misses?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LIB=${LIB:-$R/bamsignals_amd/libbamsignals_hip.so}
export BSIG_BENCH_FILES="${BSIG_BENCH_FILES:-bare real}" BSIG_BENCH_KEEP=1 BSIG_VARIANTS=$R/bamsignals_amd/libbamsignals_hip.so BSIG_BENCH_REPS=1
FILES=$(BSIG_BENCH_BLOCKS=64 python3 $R/scripts/inflate_bench.py | grep '^FILES ' | cut -c7-)
echo "files: $FILES"
IFS=';' read -ra SETS <<< "${PMC_SETS:-TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum;TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum}"
for nb in ${BLOCKS:-28672 57344}; do
  for set in "${SETS[@]}"; do
    tag=$(echo ${nb}_$set | tr ' ' '_' | cut -c1-48)
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_ib_${LTAG}$tag -- python3 $R/scripts/inflate_bench.py --child $LIB "$FILES" $nb 2 > $R/gpurun_out/pmc_ib_${LTAG}$tag.log 2>&1 || echo "set failed: $set"
    f=$(ls $R/gpurun_out/pmc_ib_${LTAG}$tag/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" $nb <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_inflate" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("blocks", sys.argv[2], k, "launches=%d" % len(v), "per launch:", " ".join("%.4g" % x for x in v[:12]))
PY
  done
done
rm -f /tmp/ib_*.bam /tmp/ib_*.bam.bai
