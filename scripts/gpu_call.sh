set -u
OUT=gpurun_out/r4c48; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1100 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=8 > $OUT/full.txt 2>&1; echo "exit $?" >> $OUT/full.txt; tail -25 $OUT/full.txt
