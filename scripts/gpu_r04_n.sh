#!/bin/bash
# round 4: four-voxel critic-input kernel -- step tests (both storage modes, 1-3 condition channels), then configs[2] and the metric
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04n
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_step.py tests/test_hip_condchannels.py tests/test_hip_api.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1
rc=$?
tail -n 6 $O/tests.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline > $O/cfg3.json 2> $O/cfg3.err || { tail -5 $O/cfg3.err; exit 1; }
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/fp32.json 2> $O/fp32.err || { tail -5 $O/fp32.err; exit 1; }
python - $O/cfg3.json $O/fp32.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], {k: v["ms_per_iteration"] for k, v in r["kernel_classes"].items()})
PY
