#!/bin/bash
# profile_round.sh TAG: rocprofv3 kernel-trace statistics of the default bench command (CPU leg skipped: it launches
# no kernels) + the PMC passes (each counter set is its own run, never combined with trace domains other than
# --kernel-trace); summaries left under gpurun_out/prof_TAG for copying into profiles/ (run through gpurun).
set -u
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu > $OUT/bench_under_rocprof.log 2>&1
echo "stats rc=$?"
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_I8" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc/p$i -o p$i --output-format csv -- python $GRAFT_REPO_ROOT/scripts/eval_loop.py > $OUT/pmc_p$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").split("(")[0].replace(", ", "_")   # k_fwd<21, 3> -> k_fwd<21_3>
        acc[(row["Counter_Name"], k)].append(float(row["Counter_Value"]))
with open("$OUT/pmc_counters.csv", "w") as o:
    o.write("# rocprofv3 --pmc passes over scripts/eval_loop.py (headline MSA); mean per launch\n")
    o.write("kernel,Counter_Name,mean,count\n")
    for (c, k), v in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        if k.startswith("k_"):
            o.write("%s,%s,%r,%d\n" % (k, c, sum(v) / len(v), len(v)))
print(open("$OUT/pmc_counters.csv").read()[:1500])
PY
rm -rf $OUT/pmc $OUT/stats/*/ 2>/dev/null
head -14 $OUT/kernel_stats.csv | cut -c1-150
