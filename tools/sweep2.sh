#!/bin/bash
for ST in 0 1 2 4 8; do
  export SDRX_DECIM_STAGGER=$ST
  python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('STAGGER=$ST value=%.0f MS/s kernel_ms=%.4f frac=%.4f grid=%d' % (d['value'], r['kernel_ms'], r['frac'], r['grid']))"
done
unset SDRX_DECIM_STAGGER
for SPW in 36 48 72; do
  export SDRX_DECIM_SPW=$SPW
  python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('SPW=$SPW value=%.0f MS/s kernel_ms=%.4f frac=%.4f grid=%d' % (d['value'], r['kernel_ms'], r['frac'], r['grid']))"
done
timeout 300 python -m pytest tests/test_decim_gpu.py -m gpu -x -q 2>&1 | tail -2
