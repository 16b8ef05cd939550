"""Condense the outputs of scripts/gpu_profile_r02.sh into the small tables kept under profiles/:
   python3 scripts/summarize_profile.py gpurun_out/prof_r02 profiles r02
* <tag>_<run>_kernel_stats.csv : the rocprofv3 --stats table (name shortened), unchanged numbers
* <tag>_hbm_traffic_<mode>.csv  : per kernel name: launches per iteration, mean FETCH_SIZE x 2 (gfx950: the counter tallies
  half the bytes of wide streaming reads, MI355X_MICROARCH.md) and WRITE_SIZE per launch, the launch's mean duration from the
  --stats run of the same configuration, and the resulting HBM GB/s.
* <tag>_l2_requests_<mode>.csv  : per kernel name: mean TCC_REQ / TCC_HIT / TCC_MISS per launch, the requested bytes (128 B per
  request: the dominant bf16 launch's count x 128 B equals its tiles x stage bytes) and the rate they were served at."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


stats = {}
for f in sorted(glob.glob(os.path.join(src, "*_kernel_stats.csv"))):
    run = os.path.basename(f)[:-len("_kernel_stats.csv")]
    rows = list(csv.DictReader(open(f)))
    stats[run] = {short(r["Name"]): r for r in rows}
    with open(os.path.join(dst, f"{tag}_{run}_kernel_stats.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

for mode, run in (("0", "fp32_bs256"), ("1", "bf16_bs256")):
    acc = defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for cset in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(src, f"pmc_bf16{mode}_{cset}.csv")
        if not os.path.exists(f):
            continue
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cset:
                continue
            k = r["Dispatch_Id"]
            per_dispatch[k] += float(r["Counter_Value"])
            names[k] = short(r["Kernel_Name"])
        for k, v in per_dispatch.items():
            acc[names[k]][cset].append(v)
    if not acc:
        continue
    out = os.path.join(dst, f"{tag}_hbm_traffic_{'bf16' if mode == '1' else 'fp32'}_bs256.csv")
    with open(out, "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["Name", "launches_in_2_iterations", "fetch_MB_per_launch_x2_corrected", "write_MB_per_launch", "avg_us_from_stats",
                    "hbm_GBps"])
        rows = []
        for n, d in acc.items():
            # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
            fe = 2.0 * sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1) * 1024 / 1e6
            wr = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1) * 1024 / 1e6
            st = stats.get(run, {}).get(n)
            us = float(st["AverageNs"]) / 1e3 if st else float("nan")
            rows.append((n, len(d["FETCH_SIZE"]) or len(d["WRITE_SIZE"]), fe, wr, us, (fe + wr) * 1e6 / (us * 1e-6) / 1e9 if st else float("nan")))
        rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.2f}", f"{r[3]:.2f}", f"{r[4]:.1f}", f"{r[5]:.0f}"])
    print("wrote", out)

# profiles/hbm_traffic_dominant.json: what bench.py reports as roofline.traffic (bytes per launch of the dominant kernel: its own
# kernel symbol, NAMETAG = 1), rewritten from the counter tables of THIS profile run so that the number cannot go stale
import json
dom = {"entries": []}
for mode, bf in (("fp32", False), ("bf16", True)):
    f = os.path.join(dst, f"{tag}_hbm_traffic_{mode}_bs256.csv")
    if not os.path.exists(f):
        continue
    # the dominant launch's kernel symbol(s): with the fused last conv the slab kernel has one symbol for the launches that store
    # their output (generator step) and one for those that do not (critic steps) -> launch-weighted mean
    hits = [r for r in csv.DictReader(open(f))
            if re.match(r"k_conv_gemm_ws<256, 64, 4, 1, \d, (true|false), 1", r["Name"]) or r["Name"].startswith("k_upconv_slab16<1")]
    if hits:
        n = sum(float(r["launches_in_2_iterations"]) for r in hits)
        by = sum((float(r["fetch_MB_per_launch_x2_corrected"]) + float(r["write_MB_per_launch"])) * 1e6 * float(r["launches_in_2_iterations"])
                 for r in hits) / n
        dom["entries"].append({"bf16": bf, "ndomain": 16, "batch": 256, "bytes_per_launch": by, "kernel": " + ".join(r["Name"] for r in hits),
                               "source": f"profiles/{tag}_hbm_traffic_{mode}_bs256.csv (separate rocprofv3 --pmc passes: FETCH_SIZE x 2 "
                                         f"+ WRITE_SIZE per launch; launch-weighted mean over the listed symbols)"})
if dom["entries"]:
    with open(os.path.join(dst, "hbm_traffic_dominant.json"), "w") as o:
        json.dump(dom, o, indent=1)
    print("wrote", os.path.join(dst, "hbm_traffic_dominant.json"))

for mode, run in (("0", "fp32_bs256"), ("1", "bf16_bs256")):
    f = os.path.join(src, f"pmc_bf16{mode}_TCC_REQ_sum.csv")
    if not os.path.exists(f):
        continue
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    out = os.path.join(dst, f"{tag}_l2_requests_{'bf16' if mode == '1' else 'fp32'}_bs256.csv")
    rows = []
    for n, d in acc.items():
        m = {k: sum(v.values()) / max(len(v), 1) for k, v in d.items()}
        st = stats.get(run, {}).get(n)
        us = float(st["AverageNs"]) / 1e3 if st else float("nan")
        req = m.get("TCC_REQ_sum", 0.0)
        rows.append((n, len(d.get("TCC_REQ_sum", {})), req, m.get("TCC_HIT_sum", 0.0), m.get("TCC_MISS_sum", 0.0), req * 128 / 1e6, us,
                     req * 128 / (us * 1e-6) / 1e12 if st else float("nan")))
    rows.sort(key=lambda r: -r[2] * r[1])
    with open(out, "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["Name", "launches_in_2_iterations", "TCC_REQ_per_launch", "TCC_HIT_per_launch", "TCC_MISS_per_launch",
                    "requested_MB_per_launch_at_128B", "avg_us_from_stats", "L2_request_TBps"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.0f}", f"{r[3]:.0f}", f"{r[4]:.0f}", f"{r[5]:.1f}", f"{r[6]:.1f}", f"{r[7]:.2f}"])
    print("wrote", out)

# SQ counters of the bf16 GEMM launches (one --pmc pass; SQ_* count quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES, which counts cycles:
# MI355X_MICROARCH.md constants table): per kernel name, means per launch and the shares of wave time
for sqmode, mops in (("bf16", "SQ_INSTS_VALU_MFMA_MOPS_BF16"), ("fp32", "SQ_INSTS_VALU_MFMA_MOPS_F32")):
    f = os.path.join(src, f"pmc_sq_{sqmode}.csv")
    if os.path.exists(f):
        acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        grbm = defaultdict(lambda: defaultdict(float))
        g = os.path.join(src, f"pmc_grbm_{sqmode}.csv")
        if os.path.exists(g):
            for r in csv.DictReader(open(g)):
                grbm[short(r["Kernel_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        out = os.path.join(dst, f"{tag}_sq_counters_{sqmode}_bs256.csv")
        with open(out, "w", newline="") as o:
            w = csv.writer(o)
            cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
                    mops]
            w.writerow(["Name", "launches"] + [c + "_per_launch" for c in cols] + ["wait_any_share_of_wave_cycles", "wait_inst_share", "active_share",
                       "mfma_busy_cycles_per_SIMD_cycle (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)", "GRBM_GUI_ACTIVE_per_launch"])
            rows = []
            for n, d in acc.items():
                m = {k: sum(v.values()) / max(len(v), 1) for k, v in d.items()}
                if m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
                    continue
                wc = max(m.get("SQ_WAVE_CYCLES", 0.0), 1.0)
                ga = sum(grbm[n].values()) / max(len(grbm[n]), 1) if n in grbm else float("nan")
                busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (ga / 8 * 1024) if ga == ga and ga > 0 else float("nan")
                rows.append([n, len(d.get("SQ_WAVE_CYCLES", {}))] + [f"{m.get(c, 0.0):.0f}" for c in cols] +
                            [f"{m.get('SQ_WAIT_ANY', 0) / wc:.3f}", f"{m.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}", f"{m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}",
                             f"{busy:.3f}", f"{ga:.0f}"])
            rows.sort(key=lambda r: -float(r[8]))
            for r in rows:
                w.writerow(r)
        print("wrote", out)
