#!/bin/bash
# round 4, first measurement: the fp32 metric line (colsum change), the nd64 shard line and its per-kernel statistics
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04a
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/fp32_bs256.json 2> $O/fp32.err || { tail -5 $O/fp32.err; exit 1; }
python - "$O/fp32_bs256.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("fp32", d["value"], d["ms_per_step"], r["frac"], r["kernel_classes"])
PY
timeout -k 10 300 python bench.py --config 5 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/cfg5_nd64_bs64.json 2> $O/cfg5.err || { tail -5 $O/cfg5.err; exit 1; }
python - "$O/cfg5_nd64_bs64.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("cfg5", d["value"], d["ms_per_step"], r["frac"], r["kernel_classes"])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof5 -- python3 $R/bench.py --config 5 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/prof5.json 2> $O/prof5.err || { tail -5 $O/prof5.err; exit 1; }
find $O/prof5 -name "*kernel_stats.csv" -exec cp {} $O/cfg5_kernel_stats.csv \;
rm -rf $O/prof5
head -40 $O/cfg5_kernel_stats.csv | cut -c1-150
