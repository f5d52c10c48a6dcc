mkdir -p gpurun_out
for v in "BAMSIGNALS_ONE_VIEW=1" "BAMSIGNALS_ONE_VIEW=1 BAMSIGNALS_INFLATE_LDS_PAD=416" "BAMSIGNALS_INFLATE_LDS_PAD=416" "BAMSIGNALS_INFLATE_LDS_PAD=200" "X=1"; do
  echo "== $v"
  env $v timeout -k 10 200 python scripts/decode_ns_time.py 500000000 3 2>&1 | grep -E "^decode|waited .* for k_inflate" | tail -10
done
