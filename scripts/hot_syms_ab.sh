# Diagnostic: k_inflate with 128 (default) / 64 of the long-code symbols in LDS (inflate_lane.h kHotSyms; 224 / 256
# resident lanes per CU), on the real-shaped file and on the north star's.  Needs
#   (cd bamsignals_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC \
#     -mllvm -amdgpu-kernarg-preload-count=8 -DBSIG_HOT_SYMS=64 -shared -o ../libbamsignals_hip_hot64.so \
#     kernels.hip runtime.hip devdecode.hip collect.hip bamio.cpp fileapi.cpp -lz -lpthread -ldl)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
export BSIG_KEEP_BAM=1
for v in "X=1" "BSIG_LIB_PATH=$R/bamsignals_amd/libbamsignals_hip_hot64.so" "X=2"; do
  echo "== $v"
  env $v timeout -k 10 200 python scripts/decode_realshaped_device_time.py 2>&1 | tail -3 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print('real-shaped rep', d['rep'], 'decode', d['decode_s'], 'inflate', d['inflate'], 'GB/s', d['inflate_output_GBps'])"
  env $v timeout -k 10 400 python scripts/decode_ns_time.py 500000000 4 2>&1 | grep -E "^decode" | tail -3
done
rm -f /tmp/ns_synth.bam /tmp/ns_synth.bam.bai
