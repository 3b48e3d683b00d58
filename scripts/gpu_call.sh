set -u
OUT=gpurun_out/r3c17; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tests/probes/consensus_reference_probe.py > $OUT/consensus_probe.log 2>&1; cat $OUT/consensus_probe.log
