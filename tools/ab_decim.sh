#!/bin/bash
# A/B of the decimator engines on one box: matrix cores (default) vs SDRX_DECIM_ENGINE=valu
for e in mfma valu; do
  SDRX_DECIM_ENGINE=$e timeout -k 10 200 python bench.py --workload decim64 --no-cpu ${@} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
