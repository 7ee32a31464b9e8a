#!/bin/bash
# Runs on the GPU box: one rocprofv3 counter pass per invocation of bench.py
# (counters never share a run with tracing), results under gpurun_out/<tag>/.
# usage: tools/prof_pmc.sh <tag> [bench args...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
pass() { # name counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --no-cpu-baseline --no-ceiling --no-photo "${BENCH_ARGS[@]}" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
BENCH_ARGS=("$@")
pass fetch FETCH_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU &&
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE &&
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline --no-ceiling --no-photo "${BENCH_ARGS[@]}" > "$OUT/trace.log" 2>&1 &&
pass waits SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM
python3 tools/pmc_summary.py "$OUT" "$OUT/pmc_traffic.json" 256 > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
