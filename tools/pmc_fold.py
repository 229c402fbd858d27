#!/usr/bin/env python3
"""Fold the passes of tools/pmc_run.sh: mean counter values per sdrx kernel and per position of the launch inside a step
(the k-th launch of that kernel since the last tree_hist_kernel / hist_update_kernel = the k-th pass of a bank feed;
rocprofv3's Grid_Size column repeats the first dispatch's value under --pmc, so it cannot tell the passes apart)."""
import csv, glob, collections, sys
O = sys.argv[1]
acc = collections.defaultdict(list)
dur = collections.defaultdict(list)
for name in ("p1", "p2", "p3", "p4"):
    for p in glob.glob("%s/%s/*/*_counter_collection.csv" % (O, name)):
        rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Dispatch_Id"]))
        pos = collections.Counter(); seen = {}
        for r in rows:
            k = r["Kernel_Name"]
            if "sdrx" not in k or "stream_sum" in k:
                continue
            short = k.split("(")[0].split("sdrx::")[-1][:40]
            if "hist" in short:
                pos.clear(); continue
            d = r["Dispatch_Id"]
            if d not in seen:
                seen[d] = pos[short]; pos[short] += 1
                dur[(short, seen[d])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            acc[(short, seen[d], r["Counter_Name"])].append(float(r["Counter_Value"]))
import os
ONLY = os.environ.get("PMC_ONLY", "tree_kernel,decim_fast,decim_chain").split(",")
last = None
for k, v in sorted(acc.items()):
    if not any(o in k[0] for o in ONLY) or k[1] > 2: continue
    if (k[0], k[1]) != last:
        d = dur[(k[0], k[1])]
        print("--- %s  launch #%d of a step: %d samples, %.1f us under --pmc" % (k[0], k[1], len(d), sum(d) / len(d))); last = (k[0], k[1])
    print("   %-28s %.4g" % (k[2], sum(v) / len(v)))
