set -u
OUT=gpurun_out/r4c35; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 120 python tests/probes/fwd_kernel_ab.py 3000 100 2>&1 | tail -3 | tee $OUT/ab_small.txt
timeout 120 python tests/probes/fwd_kernel_ab.py 2>&1 | tail -3 | tee $OUT/ab.txt
