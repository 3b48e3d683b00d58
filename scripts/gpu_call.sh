set -u
OUT=gpurun_out/r4c55; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "fit_reaches or fit_switches or fit_default or forward_kernels_agree" 2>&1 | tail -3 | tee $OUT/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
