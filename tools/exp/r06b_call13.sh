mkdir -p gpurun_out/r06b
python -m pytest tests/test_lbs_gpu.py -q -x -k "interleaved or vertex_buffer or ex_ or skin_ex or blend_shapes" 2>&1 | tail -3
timeout 900 python tools/fuzz_lbs_gpu.py --ex --count 400 --seed 7 --out gpurun_out/r06b/fuzz_lbs_ex2.json > /dev/null 2>&1; cut -c1-1500 gpurun_out/r06b/fuzz_lbs_ex2.json
bash tools/exp/r06b_call12.sh 2>&1 | grep "^ex_\|^vb_"
