set -u
OUT=gpurun_out/r3c21; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 200 python scripts/config_table.py c3 headline c2 > $OUT/ct.log 2>&1; cut -c1-230 $OUT/ct.log
timeout 120 python scripts/config3_flow.py > $OUT/c3_flow.log 2>&1; tail -3 $OUT/c3_flow.log
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fit or pin_kit or joint or resumed" 2>&1 | tail -5 ) > $OUT/pytest_fit.log 2>&1; tail -3 $OUT/pytest_fit.log
