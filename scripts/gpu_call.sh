set -u
OUT=gpurun_out/r4c54; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for f in 4 2; do
PLM_ACC_FACTOR=$f timeout 120 python scripts/c3diag.py 2>&1 | head -1 | sed "s/^/acc factor $f: /" | tee -a $OUT/c3.txt
done
