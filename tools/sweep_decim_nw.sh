#!/bin/bash
# FAST decimator: single-wave workgroups (NW = 1) against four-wave workgroups (NW = 4) over launch sizes
O=gpurun_out/${1:-nw}; mkdir -p $O
for B in ${SIZES:-1048576 4194304 10485760 16777216 33554432 67108864 134217728 268435456 1073741824}; do
  for NW in ${NWS:-1 2 4}; do
    SDRX_DECIM_NW=$NW timeout -k 10 120 python bench.py --workload decim64 --no-cpu --steps 20 --warmup 3 --batch $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('B=%d NW=%s value=%.0f MS/s kernel_ms=%.4f step_ms=%.4f grid=%d' % ($B, '$NW', d['value'], r['kernel_ms'], d['ms_per_step'], r['grid']))
" >> $O/sweep.txt || exit 1
  done
done
cat $O/sweep.txt
