#!/bin/bash
# fp32 metric configuration: workgroup target of the weight-gradient launches (RDGAN_WGRAD_WGS), box plans of the critic included
O=gpurun_out/wgsb; mkdir -p $O
for n in 1024 2048 4096 512 1024; do
  RDGAN_WGRAD_WGS=$n python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/wgs_$n.json 2>/dev/null || exit 1
  python - $O/wgs_$n.json $n <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
w = [l for l in d["roofline"]["launches"] if l["kind"] == "wgrad" and l["what"].startswith("critic layer") and "ws" in l["kernel"]]
print(sys.argv[2], d["value"], d["iteration_ms"]["median"], " ".join(f'{l["what"][7:13]}:{l["ms_per_launch"]:.4f}' for l in w))
PY
done
