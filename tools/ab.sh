# same-box A/B of tools/bench_kernels.py: A = pysco_amd/_ab/libA.so (tools/build_ab.sh: git HEAD), B = working tree
#   ONLY=fusedgather,kdwdep NCELLS=512 bash tools/ab.sh
ONLY=${ONLY:-fusedgather,kdwdep}; NCELLS=${NCELLS:-512}
for i in 1 2; do
echo "== A (HEAD)"; PMSTEP_LIB=$PWD/pysco_amd/_ab/libA.so timeout -k 10 200 python tools/bench_kernels.py --ncells $NCELLS --only $ONLY 2>&1 | grep -v amdgpu
echo "== B (work)"; timeout -k 10 200 python tools/bench_kernels.py --ncells $NCELLS --only $ONLY 2>&1 | grep -v amdgpu
done
