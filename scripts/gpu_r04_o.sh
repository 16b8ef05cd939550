#!/bin/bash
# round 4: wide weight-gradient tiles (k_wgrad_gemm_ws16<256,128>, three stages) -- op tests vs the oracle, engine equality, nd64 / configs[2] off / on
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04o
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_ops.py tests/test_hip_bf16.py -m gpu -x -q -p no:cacheprovider -k "wgrad_bf16 or wide_wgrad" > $O/tests.log 2>&1
rc=$?
tail -n 12 $O/tests.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
show() {
python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], {k: v["ms_per_iteration"] for k, v in r["kernel_classes"].items()})
for x in r["launches"]:
    if "wgrad_gemm_ws16" in x["kernel"]: print("   ", x["what"][:40], x["kernel"], x["samples"], x["launches_per_iteration"], x["ms_per_launch"], x["ms_per_iteration"], x["frac"])
PY
}
for v in 0 1; do
timeout -k 10 300 python bench.py --config 5 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --opt wgrad_wide=$v > $O/cfg5_w$v.json 2> $O/cfg5.err || { tail -5 $O/cfg5.err; exit 1; }
show $O/cfg5_w$v.json
done
for v in 0 1; do
timeout -k 10 300 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --opt wgrad_wide=$v > $O/cfg3_w$v.json 2> $O/cfg3.err || { tail -5 $O/cfg3.err; exit 1; }
show $O/cfg3_w$v.json
done
