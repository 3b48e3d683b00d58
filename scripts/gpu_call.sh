set -u
OUT=gpurun_out/r3c16; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pin_kit or real_alignment" 2>&1 | tail -40 ) > $OUT/pytest_pin.log 2>&1
grep -E "passed|failed|^FAILED|^E  |max \|dCN|vp |joint " $OUT/pytest_pin.log | head -30
timeout 60 python scripts/time_kernels.py > $OUT/time_kernels.log 2>&1; tail -1 $OUT/time_kernels.log
