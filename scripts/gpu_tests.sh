#!/bin/bash
# run the GPU parity tests in two stages; a crash/timeout (rc > 1) in stage 1 stops the run
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider > gpurun_out/ops.log 2>&1
rc=$?
tail -n 25 gpurun_out/ops.log
echo "ops rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_hip_step.py -m gpu -q -rA -p no:cacheprovider > gpurun_out/step.log 2>&1
rc2=$?
tail -n 40 gpurun_out/step.log
echo "step rc=$rc2"
exit $(( rc > rc2 ? rc : rc2 ))
