#!/bin/bash
# fp32 metric configuration under rocprofv3 --kernel-trace --stats (per-kernel averages), plus the nd64 line with the one-wave dense kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/fp32_under_rocprof.json 2> $O/prof.err || { tail -5 $O/prof.err; exit 1; }
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/fp32_bs256_kernel_stats.csv \;
rm -rf $O/prof
head -45 $O/fp32_bs256_kernel_stats.csv | cut -c1-130
cd $R
timeout -k 10 300 python bench.py --config 5 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/cfg5.json 2> $O/cfg5.err || { tail -5 $O/cfg5.err; exit 1; }
python - "$O/cfg5.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], {k: v["ms_per_iteration"] for k, v in r["kernel_classes"].items()})
for x in r["launches"]:
    if "dense" in x["what"]: print("   ", x["what"], x["kernel"], x["launches_per_iteration"], x["ms_per_launch"], x["ms_per_iteration"], x["frac"])
PY
