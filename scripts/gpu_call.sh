set -u
OUT=gpurun_out/r3c25; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( timeout 60 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "marginals or eval_matches or fit_reaches or alignment_accel or gap_mode" 2>&1 | tail -3 ) > $OUT/pytest_last.log 2>&1; tail -2 $OUT/pytest_last.log
