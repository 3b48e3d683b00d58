set -u
OUT=gpurun_out/r3c24; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 100 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -m gpu -x -q -k "pin_kit or real_alignment or solver_selection or shards_over or launch_failure or plain_c" 2>&1 | tail -6 ) > $OUT/pytest_last.log 2>&1; tail -4 $OUT/pytest_last.log
