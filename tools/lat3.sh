#!/bin/bash
set -e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/lat3
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_planes.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
timeout -k 10 100 python3 tools/single_image_latency.py > $O/latency.txt 2>&1
timeout -k 10 100 python3 tools/single_image_latency.py 800 640 >> $O/latency.txt 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 tools/latency_probe.py --iters 30 > $O/prof.log 2>&1
python3 tools/timeline.py $O/tr 10 > $O/timeline.txt 2>&1
rm -rf $O/tr
timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
