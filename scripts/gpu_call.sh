set -u
OUT=gpurun_out/r4c30; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tests/probes/bwd_kernel_ab.py 2>&1 | tail -3 | tee $OUT/ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "backward_kernel or both_backward or sharded_state or edge_shapes or eval_matches_oracle or stop_rule" 2>&1 | tail -8 | tee $OUT/tests.txt
