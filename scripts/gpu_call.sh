set -u
OUT=gpurun_out/r4c43; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "(optimality_certificate and (config3 or config2)) or resumed or fit_switches or stop_rule or small_tight or fit_reaches_oracle_optimum" 2>&1 | tail -6 | tee $OUT/tests.txt
