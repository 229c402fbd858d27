#!/bin/bash
# SQ counters for the MFMA experiment kernel (separate passes; no trace domains combined with --pmc)
O=gpurun_out/${1:-mfx_pmc}; mkdir -p $O; export TMPDIR=/tmp
N=${2:-268435456}
run() { name=$1; shift; ctrs=$1; shift
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1
}
D="python3 tools/mfma_experiment_rate.py $N 64"
run m1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" $D
run m2 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" $D
run m3 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS" $D
run m4 "GRBM_GUI_ACTIVE GRBM_COUNT" $D
python3 - <<PY
import csv,glob,collections
for name in ("m1","m2","m3","m4"):
    acc=collections.defaultdict(list)
    for p in glob.glob("$O/%s/*/*_counter_collection.csv"%name):
        for r in csv.DictReader(open(p)):
            k=r["Kernel_Name"]
            if "decim64_mfma" in k or "decim_fast" in k:
                acc[(("mfx" if "mfma" in k else "valu"), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()):
        print(name, k[0], k[1], "mean %.4g" % (sum(v)/len(v)), "n", len(v))
PY
tail -3 $O/m3.log
