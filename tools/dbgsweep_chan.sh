#!/bin/bash
# timing by elimination for the bank's tree kernel (SDRX_CHAN_DBG bits: tree_kernel.hpp; results are wrong when set)
# the hooks are compiled out of the product build: this rebuilds the bank's object with them (on the GPU box, ~1 min) and restores it afterwards
mkdir -p gpurun_out/r3a
touch sdrangel_amd/csrc/tree_kernel.hpp && make -s -C sdrangel_amd/csrc EXTRA=-DSDRX_TK_DBG=1 > /dev/null || exit 1
trap 'touch sdrangel_amd/csrc/tree_kernel.hpp; make -s -C sdrangel_amd/csrc > /dev/null' EXIT
for d in ${@:-0 1 16}; do
  SDRX_CHAN_DBG=$d timeout -k 10 120 python bench.py --workload chan32 --steps 5 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dbg $d', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
