for i in 1 2; do
echo "== A (HEAD)"; PMSTEP_LIB=$PWD/pysco_amd/_ab/libA.so timeout -k 10 200 python tools/bench_kernels.py --ncells 512 --only kdwdep 2>&1 | grep -v amdgpu
echo "== B (work)"; timeout -k 10 200 python tools/bench_kernels.py --ncells 512 --only kdwdep 2>&1 | grep -v amdgpu
done
