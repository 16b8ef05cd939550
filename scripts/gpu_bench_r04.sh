#!/bin/bash
# round-4 bench lines for BASELINE.md section 4 (one MI355X): the metric with its CPU baseline, then the other configurations
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/bench_r04
mkdir -p $O
cd $R
python bench.py > $O/config2_fp32_bs256.json 2> $O/config2.err || exit 1
python bench.py --n-critic 5 --steps 20 --warmup 5 --no-cpu-baseline > $O/fp32_nc5.json 2>/dev/null
python bench.py --config 3 --steps 10 --warmup 3 > $O/config3_bf16_bs2048_nc5.json 2>/dev/null
python bench.py --config 3 --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline > $O/cfg4_shard_bs1024.json 2>/dev/null
python bench.py --config 5 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/cfg5_shard_nd64_bs64.json 2>/dev/null
python bench.py --opt bf16=1 --steps 30 --warmup 5 --no-cpu-baseline > $O/bf16_bs256.json 2>/dev/null
python bench.py --ndomain 64 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/nd64_fp32_bs64.json 2>/dev/null
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "median", d["iteration_ms"]["median"], "dominant", r["frac"], "iteration", r["iteration"]["frac"], r["iteration"]["tflops"], r["iteration_direct_equiv_tflops"], d.get("cpu_baseline", {}).get("value"))
PY
done
