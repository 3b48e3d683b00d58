set -u
OUT=gpurun_out/r4c46; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/c3diag.py > $OUT/c3.txt 2>&1; head -1 $OUT/c3.txt
timeout 600 python scripts/config_table.py c2 headline c4 c5 headline_g c3_g 2>&1 | cut -c1-230 | tee $OUT/table.txt
