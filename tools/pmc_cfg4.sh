#!/bin/bash
# SQ / memory counters for the cfg 4 back-end kernels (separate --pmc passes; no trace domains combined with --pmc)
O=gpurun_out/${1:-cfg4pmc}; mkdir -p $O; export TMPDIR=/tmp
run() { name=$1; shift; ctrs=$1; shift
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1; }
C="python3 bench.py --workload cfg4 --batch 67108864 --no-cpu --steps 2 --warmup 1"
run p1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY" $C &&
run p2 "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $C &&
run p3 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" $C
python3 - <<PY
import csv,glob,collections
for name in ("p1","p2","p3"):
    acc=collections.defaultdict(list)
    for p in glob.glob("$O/%s/*/*_counter_collection.csv"%name):
        for r in csv.DictReader(open(p)):
            if "sdrx::be_" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("sdrx::")[1].split("(")[0][:24], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()):
        print(name, k[0].ljust(24), k[1].ljust(22), "mean %.5g" % (sum(v)/len(v)), "n", len(v))
PY
