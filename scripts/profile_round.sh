#!/bin/bash
# profile_round.sh TAG: rocprofv3 kernel-trace statistics of the default bench command + the PMC passes,
# summaries left under gpurun_out/ for copying into profiles/ (run through gpurun).
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-fit --no-cpu > $OUT/bench_under_rocprof.log 2>&1
echo "stats rc=$?"
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
bash $GRAFT_REPO_ROOT/scripts/pmc_passes.sh > $OUT/pmc_passes.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
        acc[(row["Counter_Name"], k)].append(float(row["Counter_Value"]))
with open("$OUT/pmc_counters.csv", "w") as o:
    o.write("kernel,Counter_Name,mean,count\n")
    for (c, k), v in sorted(acc.items()):
        if k.startswith("k_"):
            o.write("%s,%s,%r,%d\n" % (k, c, sum(v) / len(v), len(v)))
print(open("$OUT/pmc_counters.csv").read()[:300])
PY
head -12 $OUT/kernel_stats.csv
