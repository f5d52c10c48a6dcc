# Diagnostic: k_inflate's time per pass of the north star's decode (BSIG_DIAG_DECODE lines) at several
# occupancies (BAMSIGNALS_INFLATE_LDS_PAD: 0 = 224 lanes per CU, 136 = 192, 340 = 160, 676 = 96).
mkdir -p gpurun_out
export BSIG_KEEP_BAM=1
for pad in ${PADS:-0 136 340 676}; do
  echo "== LDS pad $pad"
  BAMSIGNALS_INFLATE_LDS_PAD=$pad timeout -k 10 400 python scripts/decode_ns_time.py 500000000 3 2>&1 | grep -E "^decode|k_inflate of" | tail -5
done
rm -f /tmp/ns_synth.bam /tmp/ns_synth.bam.bai
