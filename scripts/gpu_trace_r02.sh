#!/bin/bash
# per-dispatch trace of one training iteration in a given configuration -> gpurun_out/trace_<name>.txt
# usage: scripts/gpu_trace_r02.sh NAME [wl_iteration.py args...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_$name
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$name -- python3 $R/scripts/wl_iteration.py --iters 3 "$@" > $R/gpurun_out/trace_$name.log 2>&1 || { tail -30 $R/gpurun_out/trace_$name.log; exit 1; }
python3 $R/scratch/trace_post.py $R/gpurun_out/trace_$name > $R/gpurun_out/trace_$name.txt
rm -rf $R/gpurun_out/trace_$name
tail -2 $R/gpurun_out/trace_$name.txt
