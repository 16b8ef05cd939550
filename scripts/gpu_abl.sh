#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 120 python scratch/abl_w.py $PWD/pr_disagg_radar_gan_amd/librdgan_hip.so 2>&1 | tail -1
timeout -k 10 120 python scratch/abl_w.py $PWD/scratch/librdgan_abl1.so 2>&1 | tail -1
