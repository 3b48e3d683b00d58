#!/bin/bash
# kstats.sh TAG CMD...: rocprofv3 --kernel-trace --stats of CMD, per-kernel table (calls, avg us, total ms) to gpurun_out/TAG.kstats.txt
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG.kstats
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- "$@" > $O.log 2>&1
F=$(find $O -name "*kernel_stats.csv" | head -1)
python - "$F" > $R/gpurun_out/$TAG.kstats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print("%-72s calls=%6s avg=%9.1fus total=%9.2fms %5.1f%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
rm -rf $O
head -30 $R/gpurun_out/$TAG.kstats.txt
