# Diagnostic: blocks per wave of k_inflate (BAMSIGNALS_INFLATE_LANES) on the real-shaped file, whose passes
# hold 31,000 blocks each -- fewer 32-lane waves (970) than the chip has SIMDs (1,024) -- and on the north star's.
mkdir -p gpurun_out
export BSIG_KEEP_BAM=1
for lanes in 32 16 8 64; do
  echo "== lanes $lanes"
  BAMSIGNALS_INFLATE_LANES=$lanes timeout -k 10 200 python scripts/decode_realshaped_device_time.py 2>&1 | tail -2 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print('real-shaped rep', d['rep'], 'decode', d['decode_s'], 'inflate', d['inflate'], 'GB/s', d['inflate_output_GBps'])"
  BAMSIGNALS_INFLATE_LANES=$lanes timeout -k 10 400 python scripts/decode_ns_time.py 500000000 3 2>&1 | grep -E "^decode" | tail -2
done
rm -f /tmp/ns_synth.bam /tmp/ns_synth.bam.bai
