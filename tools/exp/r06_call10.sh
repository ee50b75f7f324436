#!/bin/bash
# Round 6: the whole GPU suite on the current library, then the bench line with the driver's arguments and the default ones.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_final; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gpu_tests.txt; cat $O/gpu_tests.txt | cut -c1-300
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "rc $?"; cp bench_full.json $O/bench_driver_args_full.json
wc -c $O/bench_driver_args.json; cat $O/bench_driver_args.json
timeout 900 python bench.py > $O/bench_plain.json 2> $O/bench_plain.err; echo "rc $?"; cp bench_full.json $O/bench_plain_full.json
cat $O/bench_plain.json | cut -c1-1500
