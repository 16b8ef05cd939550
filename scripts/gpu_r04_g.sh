#!/bin/bash
# full GPU suite (time budget, gate margins) + fp32 line + its kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04g
rm -rf $O; mkdir -p $O
cd $R
bash scripts/gpu_tests.sh || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/fp32_bs256.json 2> $O/fp32.err || { tail -5 $O/fp32.err; exit 1; }
python - "$O/fp32_bs256.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("fp32", d["value"], d["ms_per_step"], r["frac"], {k: v["ms_per_iteration"] for k, v in r["kernel_classes"].items()})
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/fp32_under_rocprof.json 2> $O/prof.err || { tail -5 $O/prof.err; exit 1; }
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/fp32_bs256_kernel_stats.csv \;
rm -rf $O/prof
grep -E "k_reduce_partials|k_colsum|k_weight_transform|k_wgrad_reduce" $O/fp32_bs256_kernel_stats.csv | cut -c1-140
