set -u
OUT=gpurun_out/r4c20; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python scripts/time_kernels.py > $OUT/tk1.log 2>&1; tail -1 $OUT/tk1.log
PLM_FWD_NSG=3 python scripts/time_kernels.py > $OUT/tk3.log 2>&1; tail -1 $OUT/tk3.log
PLM_FWD_NSG=3 timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eval_matches or fit_reaches or gap_mode or edge_shapes" 2>&1 | tail -2
