#!/usr/bin/env python3
"""Copy what tools/profile_onebox.sh measured into profiles/: usage tools/fold_onebox.py gpurun_out/TAG r03
kernel_stats: sdrx kernels (and the runtime's fills / copies) only, argument lists stripped."""
import csv, glob, os, shutil, sys
src, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(src, "bench_*.json")):
    shutil.copy(f, os.path.join("profiles", f"{tag}_{os.path.basename(f)}"))
for w in ("headline", "decim64", "chan32", "cfg4"):
    fs = glob.glob(os.path.join(src, f"trace_{w}", "*", "*_kernel_stats.csv"))
    if not fs:
        continue
    rows = list(csv.reader(open(fs[0])))
    with open(os.path.join("profiles", f"{tag}_{w}_kernel_stats.csv"), "w", newline="") as o:
        wr = csv.writer(o)
        wr.writerow(rows[0])
        for r in rows[1:]:
            n = r[0]
            if "sdrx" in n or "rocclr" in n.lower():
                r[0] = n.split("(")[0]
                wr.writerow(r)
for w in ("decim64", "chan32"):
    f = os.path.join(src, f"sq_counters_{w}.txt")
    if os.path.exists(f):
        shutil.copy(f, os.path.join("profiles", f"{tag}_sq_counters_{w}.txt"))
