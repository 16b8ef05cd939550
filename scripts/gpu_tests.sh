#!/bin/bash
# The whole -m gpu suite in ONE process, as the driver runs it, with the slowest tests listed (budget: <= 400 s of the
# driver's 900 s step limit).  Output: gpurun_out/gpu_tests.log (+ gate_observed.json).
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 ${GPU_TEST_LIMIT:-1000} python -m pytest tests -m gpu -q -rA --durations=30 -p no:cacheprovider "$@" > gpurun_out/gpu_tests.log 2>&1
rc=$?
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -n 3
grep -E "^(FAILED|ERROR)" gpurun_out/gpu_tests.log | head -n 20
sed -n '/slowest/,/short test summary/p' gpurun_out/gpu_tests.log | head -n 40
grep -h "gate guard headroom" gpurun_out/gpu_tests.log | tail -n 1
echo "gpu tests rc=$rc"
exit $rc
