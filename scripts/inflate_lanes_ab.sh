# Diagnostic: blocks per wave of k_inflate (BAMSIGNALS_INFLATE_LANES) on the real-shaped file, whose passes
# hold 31,000 blocks each -- fewer 32-lane waves (970) than the chip has SIMDs (1,024).
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in libbamsignals_hip.so libbamsignals_hip_hot64.so; do
for lanes in 32 16 8 4; do
  echo "== $lib lanes $lanes"
  BSIG_LIB_PATH=$R/bamsignals_amd/$lib BAMSIGNALS_INFLATE_LANES=$lanes timeout -k 10 200 python scripts/decode_realshaped_device_time.py 2>&1 | tail -2 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print('real-shaped rep', d['rep'], 'decode', d['decode_s'], 'inflate', d['inflate'], 'GB/s', d['inflate_output_GBps'])"
done
done
