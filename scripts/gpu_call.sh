set -u
OUT=gpurun_out/r4c25; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print({k: d[k] for k in ('value','ms_per_step','joint_lbfgs_iterations_per_s')}); print(d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('traffic_eval_total'), d['roofline'].get('traffic_eval_over_alg')); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores']); print(d['fit']['to_epsilon_1e-3']['seconds_total'], d['fit']['ignore_gaps']['seconds_optimize'])"
python bench.py --steps 30 --warmup 5 --no-cpu --no-fit > $OUT/bench30.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench30.json')); print('steps30', {k: d[k] for k in ('value','ms_per_step')})"
