for f in 40400 30400 20400 30200 20200 10200 10100 30402 20402 40402 30202 20202 40404 30404; do
  echo "==== force $f (TM*10000 + TN*100 + ks)"
  python tools/gemm_log.py 64 80 $f 2>&1 | grep -A30 "dec pass B=64 T=80: forward" | grep GEMM
  python tools/gemm_log.py 64 80 $f 2>&1 | grep -B40 "dec pass B=64 T=80: forward" | grep -A20 "enc pass B=64 T=80: forward" | grep GEMM
done
