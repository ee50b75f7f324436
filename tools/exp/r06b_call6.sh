O=gpurun_out/r06b; mkdir -p $O
timeout 1200 python tools/fuzz_lbs_gpu.py --ex --count 500 --out $O/fuzz_lbs_ex.json > /dev/null 2> $O/fuzz_lbs_ex.err; cut -c1-4000 $O/fuzz_lbs_ex.json; tail -3 $O/fuzz_lbs_ex.err | cut -c1-400
