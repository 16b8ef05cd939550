#!/bin/bash
# HBM traffic of the generator-forward kernels: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limits)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
cat > /tmp/wl.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import synthetic_batch_device
eng = Engine(16, 256)
rng = np.random.default_rng(0)
gs = eng.to_slab(W.init_generator(rng, 16))
x, c, z = synthetic_batch_device(256, 16, 1, eng.device)
for i in range(3):
    eng.gen_forward(gs, z, c)
torch.cuda.synchronize()
PY
for set in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc/t_$set -- python3 /tmp/wl.py > $R/gpurun_out/pmc/t_$set.log 2>&1 || { echo "pass $set failed"; tail -5 $R/gpurun_out/pmc/t_$set.log; }
done
