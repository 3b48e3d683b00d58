set -u
OUT=gpurun_out/r4c22; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_host_layer.py -m gpu -q -k "group_regulariser or error_conventions or end_to_end or sharded_state_eval or eval_matches" 2>&1 | tail -30 ) > $OUT/pytest1.log 2>&1; tail -30 $OUT/pytest1.log | cut -c1-220
