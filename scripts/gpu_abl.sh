#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 120 python scratch/stamp.py $PWD/scratch/librdgan_stamp.so 2>&1 | tail -6
timeout -k 10 120 python scratch/abl.py $PWD/pr_disagg_radar_gan_amd/librdgan_hip.so 2>&1 | tail -1
