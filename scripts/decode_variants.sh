# Diagnostic: the north star's and the real-shaped file's decode under k_inflate's register budgets.
# Needs builds without the three-waves-per-SIMD request:
#   for w in 2 1; do (cd bamsignals_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=8 \
#      -DBSIG_INFLATE_WAVES=$w -shared -o ../libbamsignals_hip_w$w.so kernels.hip runtime.hip devdecode.hip collect.hip bamio.cpp fileapi.cpp -lz -lpthread -ldl); done
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
export BSIG_KEEP_BAM=1
for v in "X=1" "BSIG_LIB_PATH=$R/bamsignals_amd/libbamsignals_hip_w2.so" "X=2"; do
  echo "== $v"
  env $v timeout -k 10 200 python scripts/decode_realshaped_device_time.py 2>&1 | tail -2 | cut -c1-230
  env $v timeout -k 10 300 python scripts/decode_ns_time.py 500000000 3 2>&1 | grep -E "^decode" | tail -2
done
rm -f /tmp/ns_synth.bam /tmp/ns_synth.bam.bai
