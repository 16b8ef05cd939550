#!/bin/bash
# smoke + bench + rocprofv3 kernel stats; outputs under gpurun_out/
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --steps ${STEPS:-10} --warmup 3 > gpurun_out/bench.log 2>&1 || { tail -30 gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1 || { tail -30 $R/gpurun_out/prof.log; exit 1; }
tail -1 $R/gpurun_out/prof.log
find $R/gpurun_out/prof -name "*stats*" | head
