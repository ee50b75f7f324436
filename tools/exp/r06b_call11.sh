O=gpurun_out/r06b; mkdir -p $O
run() { name=$1; shift; timeout 1200 python tools/fuzz_gpu.py "$@" --out $O/$name.json > /dev/null 2> $O/$name.err; cut -c1-2500 $O/$name.json; tail -2 $O/$name.err | cut -c1-300; }
run fuzz_diverge --first 100 --count 500 --diverge
run fuzz_diverge_listy --first 700 --count 300 --diverge --listy
