#!/bin/bash
# Run on the GPU box (through gpurun): per-kernel durations of the crowd frame (tools/bench_pose.py) and of the
# 256-character scene frame (tools/bench_scene.py), and -- with "pmc" as the first argument -- the SQ counters of the
# pose kernels DESIGN.md 4.2 quotes (waves, VALU / SALU / vector-memory instructions, wave cycles, cycles parked).
# Outputs under gpurun_out/{tp,ts,pmc_pose}.
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/tp -o pose -- python $ROOT/tools/bench_pose.py --frames 100 --palette-output > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ts -o scene -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 50 --batched-only > /dev/null 2>&1 )
if [ "${1:-}" = "pmc" ]; then
  ( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_pose -o pmc -- python $ROOT/tools/bench_pose.py --frames 20 --warmup 5 --palette-output > /dev/null 2> $ROOT/gpurun_out/pmc_pose.err )
fi
grep -E "pose_|lbs_" gpurun_out/tp/pose_kernel_stats.csv gpurun_out/ts/scene_kernel_stats.csv | cut -d, -f1-4
