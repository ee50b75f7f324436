#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06_dma; mkdir -p $O
timeout 600 python tools/exp/r06_dma_check.py > $O/check.json 2> $O/check.err; echo "check rc $?"; cat $O/check.json; tail -5 $O/check.err
for rep in 1 2; do timeout 600 python tools/exp/r06_sets_sweep.py lbs.dyn 1,2 >> $O/sets_sweep_dma.jsonl 2>> $O/sweep.err; done
cat $O/sets_sweep_dma.jsonl; tail -3 $O/sweep.err
