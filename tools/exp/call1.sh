set -u
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/c1/gputests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c1/bench_driver.json 2> gpurun_out/c1/bench_driver.err
BENCH="python $ROOT/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/c1/trace_s1" -o bench -- $BENCH --opt lbs.streams=1 > "$ROOT/gpurun_out/c1/bench_under_trace_s1.json" 2> "$ROOT/gpurun_out/c1/trace_s1.err" )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/c1/trace_pose" -o pose -- python $ROOT/tools/bench_pose.py --frames 200 > "$ROOT/gpurun_out/c1/pose_under_trace.json" 2> "$ROOT/gpurun_out/c1/trace_pose.err" )
find gpurun_out/c1 -name "*kernel_trace.csv" -delete
cat gpurun_out/c1/gputests.log
cut -c1-1500 gpurun_out/c1/bench_driver.json
