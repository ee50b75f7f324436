mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$(pwd)
timeout 400 python -m pytest tests/test_anim_gpu.py -x -q 2>&1 | tail -1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/tp -o pose -- python $ROOT/tools/bench_pose.py --frames 100 --palette-output > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ts -o scene -- python $ROOT/tools/bench_scene.py --characters 256 --instances 1 --verts 5000 --frames 50 --batched-only > /dev/null 2>&1 )
echo done
