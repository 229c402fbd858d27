#!/bin/bash
# A/B of two builds of libsdrx.so on ONE box, interleaved (box-to-box spread is +-3 %): experiments/ab/libsdrx_{A,B}.so
# usage: tools/ab_libs.sh ROUNDS -- bench.py arguments
R=$1; shift; [ "$1" = "--" ] && shift
cp sdrangel_amd/libsdrx.so /tmp/libsdrx_keep.so
trap 'cp /tmp/libsdrx_keep.so sdrangel_amd/libsdrx.so' EXIT
for r in $(seq 1 $R); do for v in A B; do
  cp experiments/ab/libsdrx_$v.so sdrangel_amd/libsdrx.so
  timeout -k 10 200 python bench.py --no-cpu "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline'].get('frac'))"
done; done
