O=gpurun_out/r06b; mkdir -p $O
run() { name=$1; shift; timeout 900 python tools/fuzz_gpu.py "$@" --out $O/$name.json > /dev/null 2> $O/$name.err; cut -c1-2500 $O/$name.json; tail -2 $O/$name.err | cut -c1-300; }
run fuzz_curves --first 0 --count 600 --curves --bones 6
run fuzz_curves_edits --first 1000 --count 600 --curves --edits --bones 9
