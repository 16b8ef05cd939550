#!/bin/bash
# per-dispatch trace of one training iteration -> gpurun_out/trace_iter.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -- python3 $R/scratch/trace_iter.py > $R/gpurun_out/trace.log 2>&1 || { tail -30 $R/gpurun_out/trace.log; exit 1; }
python3 $R/scratch/trace_post.py $R/gpurun_out/trace > $R/gpurun_out/trace_iter.txt
rm -rf $R/gpurun_out/trace
tail -3 $R/gpurun_out/trace_iter.txt
