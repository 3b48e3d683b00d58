set -u
OUT=gpurun_out/r4c29; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tests/probes/bwd_kernel_ab.py 20000 200 2>&1 | tail -3 | tee $OUT/ab_small.txt
timeout 300 python tests/probes/bwd_kernel_ab.py 2>&1 | tail -3 | tee $OUT/ab.txt
