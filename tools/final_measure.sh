#!/bin/bash
# The round's measurement set on the GPU box; everything lands under gpurun_out/<tag>/.
# usage: tools/final_measure.sh <tag>
set -u
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err || echo "bench failed"
timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --streaming --no-cpu-baseline --no-photo > $O/streaming_line.json 2>> $O/bench.err || echo "streaming bench failed"
timeout -k 10 300 python3 bench.py --gpus 1 --mode config4 --steps 2 --warmup 1 --no-cpu-baseline --no-ceiling > $O/config4_1gpu_line.json 2>> $O/bench.err || echo "config4 bench failed"
BENCH_ONE_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 2 --mode config4 --steps 2 --warmup 1 --no-cpu-baseline --no-ceiling > $O/config4_2ranks_1device_line.json 2>> $O/bench.err || echo "config4 2-rank bench failed"
timeout -k 10 100 python3 tools/single_image_latency.py > $O/single_image_latency.txt 2>&1
timeout -k 10 100 python3 tools/single_image_latency.py 800 640 >> $O/single_image_latency.txt 2>&1
timeout -k 10 300 python3 tools/cli_throughput.py 1024 1 > $O/cli_throughput.txt 2>&1
timeout -k 10 900 bash tools/prof_pmc.sh $TAG/pmc --steps 20 --warmup 3 > $O/pmc_stdout.txt 2>&1
find $O/pmc/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
ls -la $O
