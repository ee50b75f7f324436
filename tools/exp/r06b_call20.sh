O=gpurun_out/r06b/long2; mkdir -p $O
run() { name=$1; shift; timeout 1500 python tools/fuzz_gpu.py "$@" --out $O/$name.json > /dev/null 2> $O/$name.err; cut -c1-600 $O/$name.json; }
run a --first 20000 --count 1500 --listy --edits --bones 9
run b --first 22000 --count 1000 --edits --bones 17
run c --first 24000 --count 1000 --curves --edits --bones 5
run d --first 26000 --count 900 --scene 6 --listy --bones 8 --opt anim.overlap=1
run e --first 28000 --count 600 --skin --listy --bones 10
run f --first 30000 --count 600 --diverge --bones 9
run g --first 32000 --count 900 --scene 9 --bones 16 --opt debug.frame_skin=0
