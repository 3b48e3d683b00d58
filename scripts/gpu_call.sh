set -u
OUT=gpurun_out/r4c18; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/small_fit_probe.py > $OUT/small.log 2>&1; grep -v "trial" $OUT/small.log | grep mode
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_abi.py tests/test_gpu_variants.py tests/test_reference_pipeline.py tests/test_host_layer.py -m gpu -q -p no:cacheprovider > $OUT/pytest_parity.log 2>&1
tail -12 $OUT/pytest_parity.log | cut -c1-200
