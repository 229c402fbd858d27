#!/bin/bash
# usage: tools/sweep_decim.sh  -- runs bench.py over batch sizes / SPW values, prints value + kernel_ms
for B in 67108864 268435456; do
  for SPW in 0 16 32 64 128 256; do
    if [ "$SPW" = "0" ]; then unset SDRX_DECIM_SPW; else export SDRX_DECIM_SPW=$SPW; fi
    python bench.py --no-cpu --steps 10 --warmup 2 --batch $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('B=%d SPW=%s value=%.0f MS/s kernel_ms=%.4f frac=%.4f grid=%d' % ($B, '$SPW', d['value'], r['kernel_ms'], r['frac'], r['grid']))
"
  done
done
