O=gpurun_out/r06b; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 --full-record $O/bench_driver_args_full.json ) > $O/bench_driver_args.json 2> $O/bench_driver_args.err; tail -4 $O/bench_driver_args.err
( time python bench.py --full-record $O/bench_plain_full.json ) > $O/bench_plain.json 2> $O/bench_plain.err; tail -4 $O/bench_plain.err
wc -c $O/bench_driver_args.json $O/bench_plain.json
