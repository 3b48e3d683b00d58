#!/bin/bash
# pmc_fetch.sh TAG: HBM bytes per launch of the kernels of scripts/eval_loop.py (FETCH_SIZE / WRITE_SIZE passes), for A/B
# runs with PLM_HIP_LIB.  Output: gpurun_out/TAG.fetch.txt
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG.pmc
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/$set -o p --output-format csv -- python $R/scripts/eval_loop.py > $O/$set.log 2>&1
done
python - "$O" > $R/gpurun_out/$TAG.fetch.txt <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").split("(")[0]
        acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
ks = sorted({k for k, _ in acc})
for k in ks:
    f = acc.get((k, "FETCH_SIZE"), [0]); w = acc.get((k, "WRITE_SIZE"), [0])
    gb = (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024 / 1e9
    if gb > 0.05: print("%-60s %.3f GB per launch (fetch %.3f, write %.3f)" % (k[:60], gb, 2 * sum(f) / len(f) * 1024 / 1e9, sum(w) / len(w) * 1024 / 1e9))
PY
rm -rf $O
cat $R/gpurun_out/$TAG.fetch.txt
