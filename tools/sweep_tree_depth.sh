#!/bin/bash
# tree_kernel: levels per pass against LDS budget.  Deeper first passes write fewer node-stream bytes and need more LDS.
O=gpurun_out/${1:-treedepth}; mkdir -p $O
for cfg in "6 40" "6 64" "7 64" "8 80" "9 100" "10 150"; do
  set -- $cfg
  echo "== max_levels $1 lds_kb $2" >> $O/sweep.txt
  SDRX_CHAN_MAX_LEVELS=$1 SDRX_CHAN_LDS_KB=$2 SDRX_CHAN_DEBUG=1 timeout -k 10 200 python bench.py --workload chan32 --no-cpu --steps 3 --warmup 1 > $O/b_$1_$2.json 2> $O/b_$1_$2.err || exit 1
  grep -c "sdrx plan: pass 0" $O/b_$1_$2.err >> $O/sweep.txt
  grep "sdrx plan" $O/b_$1_$2.err | sort | uniq -c | sort -rn | sed -n 1,4p >> $O/sweep.txt
  python - $O/b_$1_$2.json >> $O/sweep.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; print("value %.1f GS/s  kernel_ms %.3f launches %s lds %s" % (d["value"]/1e3 if d["unit"].startswith("MS") else d["value"], r["kernel_ms"], r.get("launches"), r.get("lds_bytes")))
PY
done
cat $O/sweep.txt
