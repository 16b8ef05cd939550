#!/bin/bash
# round-2 profiles: per-kernel statistics of bench.py (fp32 metric config, bf16 at bs 256, config 3) and the HBM traffic
# counters (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) of every kernel of one iteration.  Outputs: gpurun_out/prof_r02/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; return 1; }
  tail -c 400 $O/$name.json; echo
  find $O/$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  rm -rf $O/$name
}
run_stats fp32_bs256 --steps 10 --warmup 3 && \
run_stats bf16_bs256 --steps 10 --warmup 3 --opt bf16=1 && \
run_stats bf16_cfg3 --config 3 --steps 3 --warmup 1 || exit 1
# third pass: L2 requests (128 B each) / hits / misses -- what a kernel pulls through L2 into LDS, against its HBM bytes
for mode in 0 1; do
  for set in FETCH_SIZE WRITE_SIZE "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    d=$O/pmc_bf16${mode}_${set%% *}
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $d -- python3 $R/scripts/wl_iteration.py --bf16 $mode > $d.log 2>&1 || { echo "pmc $mode $set failed"; tail -5 $d.log; exit 1; }
    find $d -name "*counter_collection.csv" -exec cp {} $O/pmc_bf16${mode}_${set%% *}.csv \;
    rm -rf $d
  done
done
ls -la $O
