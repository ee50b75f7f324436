mkdir -p gpurun_out/r06b
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r06b/gputests.txt 2>&1; tail -3 gpurun_out/r06b/gputests.txt
timeout 900 python tools/fuzz_gpu.py --first 100 --count 250 --out gpurun_out/r06b/fuzz_plain.json > /dev/null 2> gpurun_out/r06b/fuzz_plain.err; cut -c1-1500 gpurun_out/r06b/fuzz_plain.json
timeout 900 python tools/fuzz_gpu.py --first 100 --count 250 --listy --out gpurun_out/r06b/fuzz_listy.json > /dev/null 2> gpurun_out/r06b/fuzz_listy.err; cut -c1-1500 gpurun_out/r06b/fuzz_listy.json
( time python bench.py --full-record gpurun_out/r06b/bench_plain_full.json ) > gpurun_out/r06b/bench_plain.json 2> gpurun_out/r06b/bench_plain.err; tail -c 600 gpurun_out/r06b/bench_plain.err; wc -c gpurun_out/r06b/bench_plain.json
