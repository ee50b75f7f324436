O=gpurun_out/r06b/long; mkdir -p $O
run() { name=$1; shift; timeout 1500 python tools/fuzz_gpu.py "$@" --out $O/$name.json > /dev/null 2> $O/$name.err; cut -c1-1200 $O/$name.json; tail -1 $O/$name.err | cut -c1-200; }
run listy_edits_b3 --first 5000 --count 800 --listy --edits --bones 3
run listy_edits_b33 --first 6000 --count 600 --listy --edits --bones 33
run plain_edits_b64 --first 7000 --count 300 --edits --bones 64
run listy_b64 --first 7500 --count 300 --listy --bones 64
run curves_b20 --first 8000 --count 800 --curves --edits --bones 20
run scene_listy_b12 --first 9000 --count 600 --scene 6 --listy --bones 12
run skin_b33 --first 10000 --count 400 --skin --bones 33
run diverge_b12 --first 11000 --count 400 --diverge --listy --bones 12
timeout 1200 python tools/fuzz_lbs_gpu.py --count 600 --seed 11 --out $O/lbs.json > /dev/null 2>&1; cut -c1-800 $O/lbs.json
timeout 1200 python tools/fuzz_lbs_gpu.py --ex --count 600 --seed 12 --out $O/lbs_ex.json > /dev/null 2>&1; cut -c1-800 $O/lbs_ex.json
