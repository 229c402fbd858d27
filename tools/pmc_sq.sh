#!/bin/bash
# SQ counters for the two hot kernels (separate passes; no trace domains combined with --pmc)
O=gpurun_out/${1:-sq}; mkdir -p $O; export TMPDIR=/tmp
run() { # name counters... -- cmd
  name=$1; shift; ctrs=$1; shift
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1
}
D="python3 bench.py --workload decim64 --no-cpu --steps 2 --warmup 1"
C="python3 bench.py --workload chan32 --no-cpu --steps 2 --warmup 1"
run d1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" $D
run d2 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" $D
run c1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" $C
run c2 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" $C
python3 - <<PY
import csv,glob,collections
for name in ("d1","d2","c1","c2"):
    acc=collections.defaultdict(list)
    for p in glob.glob("$O/%s/*/*_counter_collection.csv"%name):
        for r in csv.DictReader(open(p)):
            if "sdrx::" in r["Kernel_Name"] and ("fast" in r["Kernel_Name"] or "tree_kernel" in r["Kernel_Name"]):
                acc[(r["Kernel_Name"].split("sdrx::")[1][:24], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()):
        print(name, k[0], k[1], "mean %.4g" % (sum(v)/len(v)), "n", len(v))
PY
