#!/bin/bash
# One GPU session: tests, bench (both workloads), rocprofv3 kernel trace + PMC passes, host-path rate.
# Outputs under gpurun_out/$1 (default r01); copy the summaries you want judged into profiles/.
TAG=${1:-r01}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
set -x
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench_decim64.json 2> $O/bench_decim64.err; cat $O/bench_decim64.json
timeout -k 10 300 python bench.py --workload chan32 --steps 5 --no-cpu > $O/bench_chan32.json 2> $O/bench_chan32.err; cat $O/bench_chan32.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_decim64 -- python3 bench.py --no-cpu --steps 10 > $O/trace_decim64.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_chan32 -- python3 bench.py --workload chan32 --no-cpu --steps 5 > $O/trace_chan32.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_decim64 -- python3 bench.py --no-cpu --steps 3 --warmup 1 > $O/pmc_fetch_decim64.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_decim64 -- python3 bench.py --no-cpu --steps 3 --warmup 1 > $O/pmc_write_decim64.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_chan32 -- python3 bench.py --workload chan32 --no-cpu --steps 3 --warmup 1 > $O/pmc_fetch_chan32.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_chan32 -- python3 bench.py --workload chan32 --no-cpu --steps 3 --warmup 1 > $O/pmc_write_chan32.log 2>&1
# cfg 4 (256 channels + demod front) and the float decimators
timeout -k 10 300 python bench.py --workload cfg4 --steps 5 --no-cpu > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cat $O/bench_cfg4.json
timeout -k 10 300 python bench.py --workload fi64 > $O/bench_fi64.json 2> $O/bench_fi64.err; cat $O/bench_fi64.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg4 -- python3 bench.py --workload cfg4 --no-cpu --steps 5 > $O/trace_cfg4.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fi64 -- python3 bench.py --workload fi64 --no-cpu --steps 10 > $O/trace_fi64.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_fi64 -- python3 bench.py --workload fi64 --no-cpu --steps 3 --warmup 1 > $O/pmc_fetch_fi64.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_fi64 -- python3 bench.py --workload fi64 --no-cpu --steps 3 --warmup 1 > $O/pmc_write_fi64.log 2>&1
timeout -k 10 120 python tools/host_path_rate.py > $O/host_path_rate.txt 2>&1; cat $O/host_path_rate.txt
find $O -name "*.csv" | head -30
