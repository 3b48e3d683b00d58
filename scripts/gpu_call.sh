set -u
OUT=gpurun_out/r4c53; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "fit_reaches or forward_kernels_agree or both_backward or fit_default or resumed or cancelled" 2>&1 | tail -4 | tee $OUT/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
