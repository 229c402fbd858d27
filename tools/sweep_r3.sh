#!/bin/bash
# round-3 launch-shape sweeps of the matrix-core kernels (environment switches only)
O=gpurun_out/${1:-r3s}; mkdir -p $O
run() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], r['kernel_ms'], r.get('grid'))"; }
echo "== bank, 1 Gi samples: chunks per segment (default = one round of 4 workgroups per CU)"
for c in 64 128 192 256; do SDRX_CHAN_CPS=$c timeout -k 10 100 python bench.py --workload chan32 --no-cpu --steps 5 2>/dev/null | run "chan32 1Gi cps=$c"; done
echo "== bank, 61.44 M samples"
for c in 4 6 8 12 15 16; do SDRX_CHAN_CPS=$c timeout -k 10 100 python bench.py --workload chan32 --no-cpu --steps 10 --batch 61440000 2>/dev/null | run "chan32 61M cps=$c"; done
timeout -k 10 100 python bench.py --workload chan32 --no-cpu --steps 10 --batch 61440000 2>/dev/null | run "chan32 61M default"
echo "== decimator, 10 M samples, four-wave flavour: sub-chunks per segment"
for s in 2 3 4 5 6 8; do SDRX_DECIM_NW=4 SDRX_DECIM_SPW=$s timeout -k 10 100 python bench.py --workload decim64 --no-cpu --steps 30 --batch 10000000 2>/dev/null | run "decim 10M nw4 spw=$s"; done
timeout -k 10 100 python bench.py --workload decim64 --no-cpu --steps 30 --batch 10000000 2>/dev/null | run "decim 10M default"
