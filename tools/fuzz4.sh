#!/bin/bash
# Runs on the GPU box: four fuzz processes side by side (different seeds), output under gpurun_out/fuzz/.
# usage: fuzz4.sh <cases per process> <first seed> [max width] [max height]
O=$GRAFT_REPO_ROOT/gpurun_out/fuzz
mkdir -p $O
cd $GRAFT_REPO_ROOT
N=$1; S=$2; MW=${3:-1300}; MH=${4:-1000}
pids=()
for k in 0 1 2 3; do
  timeout -k 10 1000 python3 -u tools/fuzz_parity.py $N $((S + k)) $MW $MH > $O/fuzz_$((S + k)).log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
tail -n 1 $O/fuzz_*.log
exit $rc
