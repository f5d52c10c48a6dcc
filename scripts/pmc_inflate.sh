# Diagnostic: hardware counters of k_inflate (one rocprofv3 --pmc pass per counter set) over the real-shaped
# file's decode (default) or the north star's (`bash scripts/pmc_inflate.sh ns`).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
KIND=${1:-real}
export BSIG_KEEP_BAM=1
if [ "$KIND" = ns ]; then TARGET="$R/scripts/decode_ns_time.py 500000000 2"; else TARGET="$R/scripts/decode_realshaped_device_time.py"; fi
# (PMC_SETS="set one;set two": other counter sets, e.g. the cache ones:
#  PMC_SETS="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum;TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum")
DEFAULT_SETS="VALUBusy SALUBusy;MemUnitBusy MemUnitStalled;LDSBankConflict L2CacheHit;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
IFS=';' read -ra SETS <<< "${PMC_SETS:-$DEFAULT_SETS}"
for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/r03_pmc_inflate_${KIND}_$tag -- python3 $TARGET > $R/gpurun_out/r03_pmc_inflate_${KIND}_$tag.log 2>&1 || echo "set failed: $set"
  f=$(ls $R/gpurun_out/r03_pmc_inflate_${KIND}_$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_inflate" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "n=%d" % len(v), "mean=%.4g" % (sum(v) / len(v)), "per launch:", " ".join("%.3g" % x for x in v[:12]))
PY
done
rm -f /tmp/ns_synth.bam /tmp/ns_synth.bam.bai
