#!/bin/bash
# Ablation of the skinning kernel options on the GPU box: prints vertices/s and roofline fraction.
OUT=${1:-gpurun_out/ablate}
mkdir -p "$OUT"
run() { python bench.py --steps 1500 --warmup 100 --no-cpu-baseline "$@" 2>>"$OUT/err.log" | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-60s %6.2f us  %6.1f GB/s  %.3f' % (' '.join(sys.argv[1:]), r['avg_launch_us'], r['achieved'], r['frac']))" "$@"; }
for rs in 0 1; do for st in 1 2; do
  run --opt lbs.streams=$st
done; done
run --opt lbs.streams=2 --random-bones
run --opt lbs.streams=2 --random-bones
run --opt lbs.streams=2 --opt lbs.exact=0
run --opt lbs.streams=1 --opt lbs.block=512 --opt lbs.blocks_per_cu=4
run --opt lbs.streams=2 --opt lbs.block=512 --opt lbs.blocks_per_cu=4
run --opt lbs.streams=2 --opt lbs.blocks_per_cu=6
run --opt lbs.streams=2 --opt lbs.blocks_per_cu=12
run --opt lbs.streams=2 --opt lbs.blocks_per_cu=16
run --opt lbs.streams=3
run --opt lbs.streams=2 --opt lbs.prefetch=1
