#!/bin/bash
# usage: tools/sweep_decim.sh [outdir] -- bench.py --workload decim64 over batch sizes x sub-chunks per wave (SDRX_DECIM_SPW), prints value + kernel_ms
O=gpurun_out/${1:-spw}; mkdir -p $O
for B in 10485760 33554432 67108864 134217728 268435456 536870912; do
  for SPW in 0 8 12 16 20 24 28 32 40 48 64; do
    if [ "$SPW" = "0" ]; then unset SDRX_DECIM_SPW; else export SDRX_DECIM_SPW=$SPW; fi
    timeout -k 10 120 python bench.py --workload decim64 --no-cpu --steps 20 --warmup 3 --batch $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('B=%d SPW=%s value=%.0f MS/s kernel_ms=%.4f frac=%.4f grid=%d' % ($B, '$SPW', d['value'], r['kernel_ms'], r['frac'], r['grid']))
" >> $O/sweep.txt || exit 1
  done
done
cat $O/sweep.txt
