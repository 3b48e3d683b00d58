set -u
OUT=gpurun_out/r4c47; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/config_table.py > $OUT/config_table.log 2>&1; cp gpurun_out/config_table.json $OUT/; cut -c1-200 $OUT/config_table.log | tail -8
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print({k: d[k] for k in ('value','ms_per_step','joint_lbfgs_iterations_per_s')}); r=d['roofline']; print(r['kernel'], r['frac'], r['traffic'], r.get('traffic_eval_total'), r.get('traffic_eval_over_alg'), r.get('measured_hbm_frac')); print(r['kernel_ms']); print(d['fit']['to_epsilon_1e-3']['seconds_total'], d['fit']['to_epsilon_1e-3']['first_time_cond_below'], d['fit']['ignore_gaps']['seconds_optimize'], d['fit']['reference_default_100_iterations']['seconds_total'], d['fit']['reference_default_100_iterations']['final_cond'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
