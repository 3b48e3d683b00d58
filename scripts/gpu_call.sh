set -u
OUT=gpurun_out/r4c41; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $OUT/full.txt 2>&1; echo "exit $?" >> $OUT/full.txt; tail -40 $OUT/full.txt
