#!/bin/bash
# ab_quick.sh TAG [pytest -k expression]: kernel statistics of the bench command under rocprofv3, two plain bench windows,
# and a slice of the GPU parity tests -- the quick look after a kernel change (run through gpurun).
TAG=${1:-ab}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu > $OUT/bench_under_rocprof.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/stats
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  python bench.py --no-cpu --no-fit 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print(d['value'], d['ms_per_step'], 'fields', k['fields'], 'vectors', k['lbfgs_vector'], 'fwd', k['forward'], 'bwd', k['backward'])"
done | tee $OUT/windows.txt
if [ -n "${2:-}" ]; then python -m pytest tests/test_gpu_parity.py -x -q -k "$2" 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $OUT/tests.txt; fi
python - <<PY
import json,re
t=open("$OUT/bench_under_rocprof.log").read()
for line in t.splitlines():
    if line.startswith("{"):
        d=json.loads(line); f=d["fit"]["to_epsilon_1e-3"]; print("fit", f["seconds_total"], f["iterations"], f["evaluations"], "value", d["value"])
PY
