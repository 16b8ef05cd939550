#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for l in pr_disagg_radar_gan_amd/librdgan_hip.so scratch/librdgan_abl1.so; do
  timeout -k 10 120 python scratch/abl.py $PWD/$l 2>&1 | tail -1
done
