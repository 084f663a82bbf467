for i in 1 2; do
echo "== A (HEAD)"; PMSTEP_LIB=$PWD/pysco_amd/_ab/libA.so timeout -k 10 200 python tools/bench_fft.py 1024 2>&1 | grep -i "fused"
echo "== B (work)"; timeout -k 10 200 python tools/bench_fft.py 1024 2>&1 | grep -i "fused"
done
