#!/bin/bash
# round 4: SQ counters of the gather GEMMs at the shape of BASELINE configs[2] (bs 2048, one critic + one generator step), with the
# fragment GEMM (k_conv_gemm_f16) on and off: matrix-pipe busy cycles, wave / wait cycles per launch.  Outputs: gpurun_out/pmc_f16/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_f16
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  d=$O/sq_f$v
  timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $d -- python3 $R/scripts/wl_iteration.py --bf16 1 --batch 2048 --opt conv_f16=$v > $d.log 2>&1 || { echo "sq $v failed"; tail -5 $d.log; exit 1; }
  mkdir -p $O/f$v; find $d -name "*counter_collection.csv" -exec cp {} $O/f$v/pmc_sq_bf16.csv \;
  rm -rf $d
  d=$O/grbm_f$v
  timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $d -- python3 $R/scripts/wl_iteration.py --bf16 1 --batch 2048 --opt conv_f16=$v > $d.log 2>&1 || { echo "grbm $v failed"; tail -5 $d.log; exit 1; }
  find $d -name "*counter_collection.csv" -exec cp {} $O/f$v/pmc_grbm_bf16.csv \;
  rm -rf $d
done
ls -la $O/f0 $O/f1
