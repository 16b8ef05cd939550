"""Post-process rocprofv3 kernel_trace.csv: print the dispatches of the last iteration in launch order."""
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_adam")]
lo = adam[-3] + 1 if len(adam) >= 3 else 0
t0 = int(rows[lo]["Start_Timestamp"])
tot = 0
for r in rows[lo:adam[-1] + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {d:8.1f}us  wg={grid:6d} x{r['Workgroup_Size_X']:>4s} lds={r.get('LDS_Block_Size','?'):>6s} {name}")
print(f"sum of kernel time {tot/1e3:.3f} ms; wall {(int(rows[adam[-1]]['End_Timestamp'])-t0)/1e6:.3f} ms")
