set -u
OUT=gpurun_out/r4c40; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --warmup 5 --no-cpu > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print({k: d[k] for k in ('value','ms_per_step')}); print(d['roofline']); print(d['fit'])"
python scripts/time_kernels.py | tee $OUT/kernels.txt
