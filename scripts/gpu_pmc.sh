#!/bin/bash
# PMC passes on a generator-forward-only workload (separate passes, no trace domains combined with --pmc)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1 || true
cat > /tmp/wl.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pr_disagg_radar_gan_amd import Engine, weights as W
from pr_disagg_radar_gan_amd.trainer import synthetic_batch_device
eng = Engine(16, 256)
rng = np.random.default_rng(0)
gs = eng.to_slab(W.init_generator(rng, 16)); ds = eng.to_slab(W.init_critic(rng, 16))
x, c, z = synthetic_batch_device(256, 16, 1, eng.device)
for i in range(2):
    eng.gen_grad(ds, gs, z, c, 7)
torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc/p$i -- python3 /tmp/wl.py > $R/gpurun_out/pmc/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/pmc/p$i.log; }
done
ls $R/gpurun_out/pmc/*
