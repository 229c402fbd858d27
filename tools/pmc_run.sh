#!/bin/bash
# SQ / GRBM counters of the sdrx kernels a command launches, per (kernel, grid size): separate rocprofv3 --pmc passes
# (never combined with trace domains).  usage: tools/pmc_run.sh TAG -- python3 bench.py --workload chan32 --no-cpu --steps 2 --warmup 1
TAG=$1; shift; [ "$1" = "--" ] && shift
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
run() { name=$1; shift; ctrs=$1; shift
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1
}
run p1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "$@"
run p2 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "$@"
run p3 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_I8" "$@"
run p4 "GRBM_GUI_ACTIVE" "$@"
python3 tools/pmc_fold.py $O
