for f in "4,4" "3,4" "2,4" "3,2" "2,2" "1,2" "1,1" "3,4,2" "2,4,2" "4,4,2" "3,2,2" "2,2,2" "4,4,4" "3,4,4"; do
  echo "==== force $f"
  CYCLEVAE_GEMM_FORCE=$f python tools/gemm_log.py 64 80 2>&1 | grep -A30 "dec pass B=64 T=80: forward" | grep GEMM
  CYCLEVAE_GEMM_FORCE=$f python tools/gemm_log.py 64 80 2>&1 | grep -B40 "dec pass B=64 T=80: forward" | grep -A20 "enc pass B=64 T=80: forward" | grep GEMM
done
