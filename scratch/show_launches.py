import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
L = d["roofline"]["launches"]; tot = 0
for l in sorted(L, key=lambda l: -l["ms_per_iteration"]):
    print(f"{l['what'][:28]:28s} {l['kind']:6s} {l['kernel'][:36]:36s} B={l['samples']:5d} n={l['launches_per_iteration']:4.1f} ms={l['ms_per_launch']:.4f} it={l['ms_per_iteration']:.3f} frac={l['frac']:.3f}")
    tot += l["ms_per_iteration"]
print(tot)
