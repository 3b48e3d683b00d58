set -u
OUT=gpurun_out/r4c38; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "forward_kernel or both_forward or sharded_state or edge_shapes or eval_matches_oracle or gap_mode or fit_reaches or accurate_and_plain or fit_switches" 2>&1 | tail -8 | tee $OUT/tests.txt
