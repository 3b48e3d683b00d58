set -u
OUT=gpurun_out/r4c21; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "arbitrary_alphabet or edge_shapes or other_alphabets or invalid_inputs or meanfield_arbitrary" 2>&1 | tail -30 ) > $OUT/pytest1.log 2>&1; tail -30 $OUT/pytest1.log | cut -c1-200
