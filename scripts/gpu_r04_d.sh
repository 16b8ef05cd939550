#!/bin/bash
# A/B in one process: fp32 weight-gradient kernels with the fragment prefetch pinned (1: bulk, 2: one read per MFMA) vs hipcc's schedule
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d
rm -rf $O; mkdir -p $O
cd $R
for v in 1 2; do
  timeout -k 10 300 python scratch/ab_libs.py pr_disagg_radar_gan_amd/librdgan_hip.so scratch/lib_wsched$v.so --rounds 4 > $O/ab_sched$v.txt 2>&1 || { tail -5 $O/ab_sched$v.txt; exit 1; }
  head -1 $O/ab_sched$v.txt | cut -c1-200
  grep -i "wgrad" $O/ab_sched$v.txt | head -14
done
