set -u
OUT=gpurun_out/r4c34; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 120 python tests/probes/bwd_kernel_ab.py 2>&1 | tail -3 | tee $OUT/ab.txt
for v in A B; do
PLM_HIP_LIB=$PWD/evcouplings_amd/libplm_$v.so timeout 120 python tests/probes/bwd_kernel_ab.py 2>&1 | grep "KERNEL=1" | sed "s/^/$v /" | tee -a $OUT/variants.txt
done
