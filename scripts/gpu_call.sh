set -u
mkdir -p gpurun_out/r3c1
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 ) > gpurun_out/r3c1/pytest.log 2>&1
echo "pytest done rc=$?" >> gpurun_out/r3c1/pytest.log
for lib in "" evcouplings_amd/libplm_rows4.so; do
  PLM_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 120 python scripts/time_kernels.py >> gpurun_out/r3c1/time_kernels.log 2>&1
  PLM_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 120 python scripts/time_kernels.py >> gpurun_out/r3c1/time_kernels.log 2>&1
done
timeout 120 python scripts/headline_fit.py > gpurun_out/r3c1/headline_fit.log 2>&1
PLM_HIP_LIB=$GRAFT_REPO_ROOT/evcouplings_amd/libplm_rows4.so timeout 120 python scripts/headline_fit.py > gpurun_out/r3c1/headline_fit_rows4.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3c1/bench.json 2> gpurun_out/r3c1/bench.err
PLM_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-fit --no-cpu > gpurun_out/r3c1/bench2.json 2> gpurun_out/r3c1/bench2.err
echo "bench2 rc=$?" >> gpurun_out/r3c1/bench2.err
tail -5 gpurun_out/r3c1/pytest.log; cat gpurun_out/r3c1/time_kernels.log; cat gpurun_out/r3c1/headline_fit.log | head -3; head -3 gpurun_out/r3c1/headline_fit_rows4.log; cut -c1-600 gpurun_out/r3c1/bench.json; cut -c1-300 gpurun_out/r3c1/bench2.json
