#!/bin/bash
# shader clock under load: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, per kernel of the headline step.
# Durations come from the same pass (--kernel-trace is allowed next to --pmc; no other trace domain).
O=gpurun_out/${1:-clock}; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/c -- python3 bench.py --no-cpu --no-also --steps 3 --warmup 1 > $O/c.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
cnt=collections.defaultdict(list); dur={}
for p in glob.glob("$O/c/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(p)): dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for p in glob.glob("$O/c/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        n=r["Kernel_Name"]
        if "sdrx" in n and r["Counter_Name"]=="GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            cnt[n.split("(")[0][-34:]].append((float(r["Counter_Value"]), dur[r["Dispatch_Id"]]))
for k,v in cnt.items():
    v=[x for x in v if x[1]>100000]
    if not v: continue
    c=sum(x[0] for x in v)/len(v); d=sum(x[1] for x in v)/len(v)
    print("%-36s launches %3d  GRBM_GUI_ACTIVE %12.0f  avg %.3f ms  shader clock %.2f GHz" % (k, len(v), c, d/1e6, c/8/d))
PY
