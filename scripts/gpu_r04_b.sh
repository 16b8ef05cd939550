#!/bin/bash
# round 4: tiled block-3 slab kernel -- op test, engine equality tests, nd64 oracle tests, nd64 bench with and without it
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_ops.py tests/test_hip_bf16.py -m gpu -x -q -p no:cacheprovider -k "tiled or (storage_forward_and_step and 64)" > $O/tests.log 2>&1
rc=$?
tail -n 15 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for v in 0 1; do
  timeout -k 10 300 python bench.py --config 5 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --opt upconv_slab_t=$v > $O/cfg5_t$v.json 2> $O/cfg5_t$v.err || { tail -5 $O/cfg5_t$v.err; exit 1; }
  python - "$O/cfg5_t$v.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dominant", r["kernel"][:40], r["frac"], r["avg_launch_ms"])
for x in r["launches"]:
    if "block3" in x["what"] or "64->1" in x["what"]:
        print("   ", x["what"], x["kernel"], x["launches_per_iteration"], x["ms_per_launch"], x["frac"])
PY
done
