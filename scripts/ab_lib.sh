#!/bin/bash
# ab_lib.sh TAG LIB...: the bench window of several builds of the library (evcouplings_amd/libplm_LIB.so, PLM_HIP_LIB) on
# ONE box, twice each, interleaved -- box-to-box variance is larger than most kernel changes (NOTES_r06.md)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
for lib in "$@"; do
  PLM_HIP_LIB=$GRAFT_REPO_ROOT/evcouplings_amd/libplm_$lib.so python bench.py --no-cpu --no-fit 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('$lib', round(d['value'],2), round(d['ms_per_step'],3), 'fwd', round(k['forward'],4), 'fwd_iso', round(k['forward_isolated'],4), 'bwd', round(k['backward'],4), 'bwd_iso', round(k['backward_isolated'],4), 'fields', round(k['fields'],4))"
done; done | tee gpurun_out/$TAG/ab.txt
