#!/bin/bash
# second pass: the five-wave form with fewer workgroups per CU / three launch streams, then the drawn-kernel tests on it
mkdir -p gpurun_out
out=gpurun_out/dyn_waves2.jsonl
: > $out
W5=$PWD/tools/exp/libs/libfyrox_hip_w5.so
timeout 60 python tools/exp/dyn_waves.py product >> $out 2>gpurun_out/dyn_waves2.err
FYX_LIB_PATH=$W5 timeout 60 python tools/exp/dyn_waves.py w5 >> $out 2>>gpurun_out/dyn_waves2.err
FYX_LIB_PATH=$W5 timeout 60 python tools/exp/dyn_waves.py w5_bpc4 lbs.dyn_bpc=4 >> $out 2>>gpurun_out/dyn_waves2.err
FYX_LIB_PATH=$W5 timeout 60 python tools/exp/dyn_waves.py w5_s3 lbs.streams=3 >> $out 2>>gpurun_out/dyn_waves2.err
timeout 60 python tools/exp/dyn_waves.py product_s3 lbs.streams=3 >> $out 2>>gpurun_out/dyn_waves2.err
cat $out
FYX_TEST_NO_TORCH=1 FYX_LIB_PATH=$W5 timeout 100 python -m pytest tests/test_lbs_gpu.py -m gpu -x -q -k "drawn_kernel" 2>&1 | grep -v "^$" | tail -4
