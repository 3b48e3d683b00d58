set -u
OUT=gpurun_out/r4c50; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 120 python tests/probes/fwd_kernel_ab.py 2>&1 | tail -3 | tee $OUT/ab.txt
PLM_HIP_LIB=$PWD/evcouplings_amd/libplm_fswar.so timeout 120 python tests/probes/fwd_kernel_ab.py 2>&1 | tail -3 | sed "s/^/swar /" | tee -a $OUT/ab.txt
PLM_HIP_LIB=$PWD/evcouplings_amd/libplm_fswar.so timeout 120 python tests/probes/fwd_kernel_ab.py 3000 100 2>&1 | tail -3 | sed "s/^/swar-small /" | tee -a $OUT/ab.txt
