#!/bin/bash
# round 4: tiled layer-2 weight gradient (k_d2_wgrad_slab_t16) -- op test vs the definition + one-hot probe, engine equality, nd64 off / on
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_ops.py tests/test_hip_bf16.py -m gpu -x -q -p no:cacheprovider -k "d2_wgrad" > $O/tests.log 2>&1
rc=$?
tail -n 14 $O/tests.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
for v in 0 1; do
timeout -k 10 300 python bench.py --config 5 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --opt d2_wgrad_slab=$v > $O/cfg5_w$v.json 2> $O/cfg5.err || { tail -5 $O/cfg5.err; exit 1; }
python - "$O/cfg5_w$v.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], {k: v["ms_per_iteration"] for k, v in r["kernel_classes"].items()})
for x in r["launches"]:
    if "layer2" in x["what"]: print("   ", x["what"], x["kernel"], x["samples"], x["launches_per_iteration"], x["ms_per_launch"], x["ms_per_iteration"], x["frac"])
PY
done
