#!/bin/bash
# round 4: scheduling fences in the fp32 GEMM loops, direct dense wgrad, colsum with 1024 threads -- tests + bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests/test_hip_step.py tests/test_hip_ops.py tests/test_golden.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1
rc=$?
tail -n 6 $O/tests.log
[ $rc -ne 0 ] && exit $rc
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "dominant", r["frac"], r["avg_launch_ms"], {k: v["ms_per_iteration"] for k, v in r["kernel_classes"].items()})
for x in r["launches"]:
    if x["kind"] == "wgrad" or "dense" in x["what"]:
        print("   ", x["what"], x["kernel"], x["launches_per_iteration"], x["ms_per_launch"], x["frac"])
PY
}
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/fp32_bs256.json 2> $O/fp32.err || { tail -5 $O/fp32.err; exit 1; }
show $O/fp32_bs256.json
timeout -k 10 300 python bench.py --config 5 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/cfg5.json 2> $O/cfg5.err || { tail -5 $O/cfg5.err; exit 1; }
show $O/cfg5.json
timeout -k 10 300 python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline > $O/cfg3.json 2> $O/cfg3.err || { tail -5 $O/cfg3.err; exit 1; }
show $O/cfg3.json
