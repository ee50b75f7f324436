O=gpurun_out/r06b; mkdir -p $O
run() { name=$1; shift; timeout 900 python tools/fuzz_gpu.py "$@" --out $O/$name.json > /dev/null 2> $O/$name.err; cut -c1-2500 $O/$name.json; tail -2 $O/$name.err | cut -c1-300; }
run fuzz_skin --first 100 --count 400 --skin
run fuzz_skin_listy --first 600 --count 300 --skin --listy --bones 12
