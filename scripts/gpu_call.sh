set -u
OUT=gpurun_out/r4c51; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print({k: d[k] for k in ('value','ms_per_step','joint_lbfgs_iterations_per_s')}); r=d['roofline']; print(r['kernel'], r['frac'], r['traffic']); print(r['kernel_ms']); print(d['fit']['to_epsilon_1e-3']['seconds_total'], d['fit']['ignore_gaps']['seconds_optimize']); print(d['cpu_baseline']['value'])"
