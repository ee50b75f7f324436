# Third session's closing run: the suite in order, smoke, the driver's bench command plain and under the kernel trace.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06h; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; grep -E "passed|failed" $OUT/gpu_tests.txt | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 --full-record $OUT/bench_driver_args_full.json > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; wc -c $OUT/bench_driver_args.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --full-record $OUT/bench_under_trace_full.json > $OUT/bench_under_trace.json 2> $OUT/trace.err )
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/driver_kernel_stats.csv; rm -rf $OUT/trace
head -5 $OUT/driver_kernel_stats.csv | cut -c1-200
