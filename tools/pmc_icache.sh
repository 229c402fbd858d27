#!/bin/bash
# instruction-cache and issue-stall counters of the tree kernel (separate pass, no trace domains)
O=gpurun_out/${1:-icache}; mkdir -p $O; export TMPDIR=/tmp
C="python3 bench.py --workload chan32 --no-cpu --steps 2 --warmup 1"
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH --output-format csv -d $O/i1 -- $C > $O/i1.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
for p in glob.glob("$O/i1/*/*_counter_collection.csv"):
    by=collections.defaultdict(dict)
    for r in csv.DictReader(open(p)):
        if "tree_kernel" in r["Kernel_Name"]: by[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"])
    for i in sorted(by)[-3:]: print(i, by[i])
PY
