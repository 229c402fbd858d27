#!/bin/bash
# The numbers the judge compares, from ONE box in one call (boxes differ by +-5 % in clock): bench lines, rocprofv3 kernel traces of the
# same commands, HBM traffic passes.  usage: tools/profile_onebox.sh TAG ; outputs under gpurun_out/TAG, fold with tools/fold_onebox.py
TAG=${1:-r03x}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
stop() { if [ "$1" = 124 ] || [ "$1" = 137 ]; then echo "a step was killed at its time limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 300 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err || stop $?
timeout -k 10 300 python bench.py --workload decim64 --no-cpu > $O/bench_decim64.json 2>/dev/null || stop $?
timeout -k 10 300 python bench.py --workload chan32 --steps 5 --no-cpu > $O/bench_chan32.json 2>/dev/null || stop $?
timeout -k 10 300 python bench.py --workload cfg4 --steps 5 --no-cpu > $O/bench_cfg4.json 2>/dev/null || stop $?
timeout -k 10 300 python bench.py --workload chan128 --steps 5 --no-cpu > $O/bench_chan128.json 2>/dev/null || stop $?
for w in headline decim64 chan32 cfg4; do
  a="--workload $w"; [ $w = headline ] && a="--no-also"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 bench.py $a --no-cpu --steps 10 > $O/trace_$w.log 2>&1 || stop $?
done
for w in decim64 chan32; do
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- python3 bench.py --workload $w --no-cpu --steps 3 --warmup 1 > $O/pmc_fetch_$w.log 2>&1 || stop $?
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -- python3 bench.py --workload $w --no-cpu --steps 3 --warmup 1 > $O/pmc_write_$w.log 2>&1 || stop $?
done
bash tools/pmc_run.sh $TAG/pmc_decim64 -- python3 bench.py --workload decim64 --no-cpu --steps 2 --warmup 1 > $O/sq_counters_decim64.txt 2>&1
bash tools/pmc_run.sh $TAG/pmc_chan32 -- python3 bench.py --workload chan32 --no-cpu --steps 2 --warmup 1 > $O/sq_counters_chan32.txt 2>&1
for f in $O/bench_*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r.get('kernel_ms'), r.get('frac'))"; done
