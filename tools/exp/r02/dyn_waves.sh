#!/bin/bash
# lbs_skin_dyn at 4 (SEQ math), 5 and 6 waves per SIMD against the product build, one box, one run.
# Build first (CPU): tools/exp/build_variants.sh w4 "-DFYX_EXP_DYN_WAVES=4" w5 "-DFYX_EXP_DYN_WAVES=5" w6 "-DFYX_EXP_DYN_WAVES=6"
mkdir -p gpurun_out
out=gpurun_out/dyn_waves.jsonl
: > $out
timeout 60 python tools/exp/dyn_waves.py product >> $out 2>gpurun_out/dyn_waves.err
for t in w5 w4 w6; do
  FYX_LIB_PATH=$PWD/tools/exp/libs/libfyrox_hip_$t.so timeout 60 python tools/exp/dyn_waves.py $t >> $out 2>>gpurun_out/dyn_waves.err
done
timeout 60 python tools/exp/dyn_waves.py product2 >> $out 2>>gpurun_out/dyn_waves.err
cat $out
