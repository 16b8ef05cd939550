#!/bin/bash
# per-kernel statistics of the nd64 shard configuration (bench.py --config 5 --batch 64)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof5 -- python3 $R/bench.py --config 5 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/prof5.json 2> $O/prof5.err || { tail -5 $O/prof5.err; exit 1; }
find $O/prof5 -name "*kernel_stats.csv" -exec cp {} $O/cfg5_kernel_stats.csv \;
rm -rf $O/prof5
python3 - $O/cfg5_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = [int(r['Calls']) for r in rows if r['Name'].startswith('k_gen_loss')][0]
tot = 0
for r in rows[:42]:
    ms = float(r['TotalDurationNs']) / 1e6 / n; tot += ms
    print(f"{r['Name'][:70]:70s} {int(r['Calls'])/n:6.1f}/it avg {float(r['AverageNs'])/1e3:8.1f} us {ms:7.3f} ms/it")
print("iterations", n, "sum of the rows", round(tot, 3))
PY
