set -u
OUT=gpurun_out/r5c6; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=25 2>&1 | tail -45 | tee $OUT/tests_all.txt
