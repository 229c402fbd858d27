#!/bin/bash
# tree_kernel: levels per pass against LDS budget.  usage: tools/sweep_tree_depth.sh [outdir] [workload] ["L KB" ...]
O=gpurun_out/${1:-treedepth}; mkdir -p $O
WL=${2:-chan32}
shift; shift
CFGS=("$@")
if [ ${#CFGS[@]} -eq 0 ]; then CFGS=("6 40" "6 64" "7 64" "8 80" "9 100" "10 150"); fi
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  echo "== max_levels $1 lds_kb $2" >> $O/sweep.txt
  SDRX_CHAN_MAX_LEVELS=$1 SDRX_CHAN_LDS_KB=$2 SDRX_CHAN_DEBUG=1 timeout -k 10 200 python bench.py --workload $WL --no-cpu --steps 3 --warmup 1 > $O/b_$1_$2.json 2> $O/b_$1_$2.err || exit 1
  grep "sdrx plan" $O/b_$1_$2.err | sed 's/stream [0-9]* (trie node [0-9]*, /(/' | sort | uniq -c | sort -k3,3n | head -8 >> $O/sweep.txt
  python - $O/b_$1_$2.json >> $O/sweep.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; print("value %.1f GS/s  kernel_ms %.3f launches %s lds %s" % (d["value"]/1e3 if d["unit"].startswith("MS") else d["value"], r["kernel_ms"], r.get("launches"), r.get("lds_bytes")))
PY
done
cat $O/sweep.txt
