O=gpurun_out/r06b; mkdir -p $O
run() { name=$1; shift; timeout 900 python tools/fuzz_gpu.py "$@" --out $O/$name.json > /dev/null 2> $O/$name.err; cut -c1-1800 $O/$name.json; tail -2 $O/$name.err | cut -c1-300; }
run fuzz_edits --first 100 --count 600 --edits
run fuzz_edits_listy --first 100 --count 600 --edits --listy
run fuzz_scene --first 100 --count 600 --scene 6
run fuzz_scene_listy --first 1000 --count 300 --scene 5 --listy
run fuzz_scene_one_launch_off --first 2000 --count 300 --scene 6 --opt anim.one_launch=0
run fuzz_scene_overlap --first 3000 --count 300 --scene 6 --opt anim.overlap=1
