#!/bin/bash
# round-4 profiles: per-kernel statistics of bench.py (fp32 metric config, bf16 at bs 256, config 3), the HBM traffic counters
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) of every kernel of one iteration, and SQ counters of the bf16 GEMMs
# (matrix-pipe busy cycles, wave cycles, wait cycles).  Outputs: gpurun_out/prof_r04/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; return 1; }
  tail -c 300 $O/$name.json; echo
  find $O/$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  rm -rf $O/$name
}
run_stats fp32_bs256 --steps 10 --warmup 3 && \
run_stats bf16_bs256 --steps 10 --warmup 3 --opt bf16=1 && \
run_stats bf16_cfg3 --config 3 --steps 3 --warmup 1 && \
run_stats bf16_cfg5_nd64_bs64 --config 5 --batch 64 --steps 5 --warmup 2 || exit 1
for mode in 0 1; do
  for set in FETCH_SIZE WRITE_SIZE; do
    d=$O/pmc_bf16${mode}_${set%% *}
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $d -- python3 $R/scripts/wl_iteration.py --bf16 $mode > $d.log 2>&1 || { echo "pmc $mode $set failed"; tail -5 $d.log; exit 1; }
    find $d -name "*counter_collection.csv" -exec cp {} $O/pmc_bf16${mode}_${set%% *}.csv \;
    rm -rf $d
  done
done
# SQ counters of the bf16 iteration: what the GEMM waves spend their cycles on (one pass, <= 8 SQ counters)
d=$O/pmc_sq_bf16
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $d -- python3 $R/scripts/wl_iteration.py --bf16 1 > $d.log 2>&1 || { echo "pmc sq failed"; tail -5 $d.log; }
find $d -name "*counter_collection.csv" -exec cp {} $O/pmc_sq_bf16.csv \;
rm -rf $d
d=$O/pmc_grbm_bf16
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $d -- python3 $R/scripts/wl_iteration.py --bf16 1 > $d.log 2>&1 || { echo "pmc grbm failed"; tail -5 $d.log; }
find $d -name "*counter_collection.csv" -exec cp {} $O/pmc_grbm_bf16.csv \;
rm -rf $d
# the same two passes over the fp32 iteration (the weight-gradient kernels of the metric configuration)
d=$O/pmc_sq_fp32
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $d -- python3 $R/scripts/wl_iteration.py --bf16 0 > $d.log 2>&1 || { echo "pmc sq fp32 failed"; tail -5 $d.log; }
find $d -name "*counter_collection.csv" -exec cp {} $O/pmc_sq_fp32.csv \;
rm -rf $d
d=$O/pmc_grbm_fp32
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $d -- python3 $R/scripts/wl_iteration.py --bf16 0 > $d.log 2>&1 || { echo "pmc grbm fp32 failed"; tail -5 $d.log; }
find $d -name "*counter_collection.csv" -exec cp {} $O/pmc_grbm_fp32.csv \;
rm -rf $d
ls -la $O
