set -u
OUT=gpurun_out/r3c13; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -30 ) > $OUT/pytest_parity.log 2>&1
grep -E "passed|failed|^FAILED|^E  " $OUT/pytest_parity.log | head
( timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -s 2>&1 ) > $OUT/pytest_scale.log 2>&1
grep -E "passed|failed|^FAILED|config[0-9]:|headline:|^E  " $OUT/pytest_scale.log | cut -c1-200 | head -30
