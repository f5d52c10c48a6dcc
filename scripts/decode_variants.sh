# Diagnostic: the north star's cold decode under k_inflate's occupancy settings.
# Needs a second build without the three-waves-per-SIMD request:
#   (cd bamsignals_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=8 \
#      -DBSIG_INFLATE_WAVES=1 -shared -o ../libbamsignals_hip_w1.so kernels.hip runtime.hip devdecode.hip collect.hip bamio.cpp fileapi.cpp -lz -lpthread -ldl)
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "X=1" "BAMSIGNALS_INFLATE_LDS_PAD=416" "BAMSIGNALS_TWO_VIEWS=1" "BSIG_LIB_PATH=$R/bamsignals_amd/libbamsignals_hip_w1.so" "BSIG_LIB_PATH=$R/bamsignals_amd/libbamsignals_hip_w1.so BAMSIGNALS_INFLATE_LDS_PAD=416"; do
  echo "== $v"
  env $v timeout -k 10 200 python scripts/decode_ns_time.py 500000000 3 2>&1 | grep -E "^decode|waited .* for k_inflate" | tail -5
done
