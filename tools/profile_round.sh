#!/bin/bash
# One GPU session: tests, bench (headline + per-workload), rocprofv3 kernel trace + PMC passes, host-path rate.
# Outputs under gpurun_out/$1 (default r02); copy the summaries you want judged into profiles/.
TAG=${1:-r03}
PART=${2:-all}      # a: tests + bench lines, b: rocprofv3 traces + PMC passes, all: both (may exceed one gpurun call)
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
set -x
# a step that hits its time limit ends the session: no further GPU step after a killed one
stop() { if [ "$1" = 124 ] || [ "$1" = 137 ]; then echo "a step was killed at its time limit (rc $1): stopping"; exit 1; fi; }
if [ "$PART" != b ]; then
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 || stop $?
tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err || stop $?
cat $O/bench_headline.json
timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-also > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2.err || stop $?
cat $O/bench_gpus2_rehearsal.json
timeout -k 10 300 python bench.py --workload decim64 --no-cpu > $O/bench_decim64.json 2> $O/bench_decim64.err || stop $?
cat $O/bench_decim64.json
timeout -k 10 300 python bench.py --workload decim64 --streams-per-gpu 64 --batch 67108864 --no-cpu > $O/bench_decim64_64streams_64Mi.json 2>/dev/null || stop $?
cat $O/bench_decim64_64streams_64Mi.json
timeout -k 10 300 python bench.py --workload decim64 --batch 67108864 --no-cpu > $O/bench_decim64_1stream_64Mi.json 2>/dev/null || stop $?
cat $O/bench_decim64_1stream_64Mi.json
timeout -k 10 300 python bench.py --workload chan32 --steps 5 --no-cpu > $O/bench_chan32.json 2> $O/bench_chan32.err || stop $?
cat $O/bench_chan32.json
# cfg 4 (256 channels + demod front), cfg 5 share (128 channels) and the float decimators
timeout -k 10 300 python bench.py --workload cfg4 --steps 5 --no-cpu > $O/bench_cfg4.json 2> $O/bench_cfg4.err || stop $?
cat $O/bench_cfg4.json
timeout -k 10 300 python bench.py --workload chan128 --steps 5 --no-cpu > $O/bench_chan128.json 2> $O/bench_chan128.err || stop $?
cat $O/bench_chan128.json
timeout -k 10 300 python bench.py --workload fi64 --no-cpu > $O/bench_fi64.json 2> $O/bench_fi64.err || stop $?
cat $O/bench_fi64.json
# A/B of the half-band engines on this box: matrix cores (default) against the dot2 kernels of rounds 1-2
for e in mfma valu; do
  SDRX_DECIM_ENGINE=$e timeout -k 10 300 python bench.py --workload decim64 --no-cpu > $O/bench_decim64_$e.json 2>/dev/null || stop $?
  SDRX_CHAN_ENGINE=$e timeout -k 10 300 python bench.py --workload chan32 --steps 5 --no-cpu > $O/bench_chan32_$e.json 2>/dev/null || stop $?
done
python3 - <<PY > $O/engine_ab.txt
import json
for w in ("decim64", "chan32"):
    for e in ("mfma", "valu"):
        d = json.load(open("$O/bench_%s_%s.json" % (w, e)))
        print(w, e, "value %.0f MS/s" % d["value"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], d["roofline"]["kernel"])
PY
cat $O/engine_ab.txt
fi
if [ "$PART" != a ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_headline -- python3 bench.py --no-cpu --no-also --steps 10 > $O/trace_headline.log 2>&1 || stop $?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_decim64 -- python3 bench.py --workload decim64 --no-cpu --steps 10 > $O/trace_decim64.log 2>&1 || stop $?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_chan32 -- python3 bench.py --workload chan32 --no-cpu --steps 5 > $O/trace_chan32.log 2>&1 || stop $?
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_decim64 -- python3 bench.py --workload decim64 --no-cpu --steps 3 --warmup 1 > $O/pmc_fetch_decim64.log 2>&1 || stop $?
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_decim64 -- python3 bench.py --workload decim64 --no-cpu --steps 3 --warmup 1 > $O/pmc_write_decim64.log 2>&1 || stop $?
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_chan32 -- python3 bench.py --workload chan32 --no-cpu --steps 3 --warmup 1 > $O/pmc_fetch_chan32.log 2>&1 || stop $?
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_chan32 -- python3 bench.py --workload chan32 --no-cpu --steps 3 --warmup 1 > $O/pmc_write_chan32.log 2>&1 || stop $?
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg4 -- python3 bench.py --workload cfg4 --no-cpu --steps 5 > $O/trace_cfg4.log 2>&1 || stop $?
bash tools/pmc_run.sh ${TAG}/pmc_decim64 -- python3 bench.py --workload decim64 --no-cpu --steps 2 --warmup 1 > $O/sq_counters_decim64.txt 2>&1; cat $O/sq_counters_decim64.txt
bash tools/pmc_run.sh ${TAG}/pmc_chan32 -- python3 bench.py --workload chan32 --no-cpu --steps 2 --warmup 1 > $O/sq_counters_chan32.txt 2>&1; cat $O/sq_counters_chan32.txt
fi
find $O -name "*stats*.csv" | head -30
