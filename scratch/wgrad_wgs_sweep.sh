#!/bin/bash
# fp32 metric configuration with the weight-gradient launches aimed at N workgroups (RDGAN_WGRAD_WGS; default 1024 = two rounds)
O=gpurun_out/wgs; mkdir -p $O
for n in 1024 512 768 1536 2048 1024; do
  RDGAN_WGRAD_WGS=$n python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/wgs_$n.json 2>/dev/null || exit 1
  python - $O/wgs_$n.json $n <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
w = [l for l in d["roofline"]["launches"] if l["kind"] == "wgrad" and "k_wgrad_gemm_ws" in l["kernel"]]
print(sys.argv[2], d["value"], d["iteration_ms"]["median"], "wgrad class", d["roofline"]["kernel_classes"]["gen_conv_wgrad"]["ms_per_iteration"],
      " ".join(f'{l["kernel"][15:]}:{l["ms_per_launch"]:.4f}' for l in w[:8]))
PY
done
