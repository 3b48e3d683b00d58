#!/bin/bash
# two quick rocprofv3 counter passes (SQ issue/wait counters, GRBM cycles) for kernel A/B work
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python $GRAFT_REPO_ROOT/scripts/eval_loop.py > $OUT/p$i.log 2>&1
  echo "pass $i ($set): rc=$?"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
        if k in ("k_fwd", "k_bwd", "k_fwd_w", "k_bwd_w"):
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%s,%s,%.6g,%d" % (k, c, sum(v) / len(v), len(v)))
PY
