#!/bin/bash
# Round 6, first GPU call: the compact bench line with the driver's arguments, the counter list, the XCD-range A/B.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_call1; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.line 2> $O/bench_driver_args.err; echo "bench rc $?" > $O/rc.txt
cp bench_full.json $O/bench_driver_args_full.json 2>/dev/null
(cd /tmp && (rocprofv3 -L > $O/counters.txt 2>&1 || rocprofv3-avail list > $O/counters.txt 2>&1))
for rep in 1 2; do timeout 600 python tools/exp/r06_sets_sweep.py lbs.dyn_map 0,1 >> $O/sets_sweep_dyn_map.jsonl 2>> $O/sweep.err; done
wc -c $O/bench_driver_args.line; tail -c 600 $O/bench_driver_args.line; echo; cat $O/sets_sweep_dyn_map.jsonl
