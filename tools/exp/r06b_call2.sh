O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_anim_gpu.py -q -x -k "lists_of_values" 2>&1 | tail -2
timeout 900 python tools/fuzz_gpu.py --first 100 --count 1500 --listy --out $O/fuzz_listy.json > /dev/null 2> $O/fuzz_listy.err; cut -c1-1200 $O/fuzz_listy.json
timeout 600 python tools/fuzz_gpu.py --first 2000 --count 600 --listy --bones 19 --out $O/fuzz_listy_19.json > /dev/null 2> $O/fuzz_listy_19.err; cut -c1-1200 $O/fuzz_listy_19.json
timeout 600 python tools/fuzz_gpu.py --first 350 --count 1000 --out $O/fuzz_plain.json > /dev/null 2> $O/fuzz_plain.err; cut -c1-1200 $O/fuzz_plain.json
timeout 900 python tools/fuzz_lbs_gpu.py --count 400 --out $O/fuzz_lbs.json > /dev/null 2> $O/fuzz_lbs.err; cut -c1-2500 $O/fuzz_lbs.json; tail -3 $O/fuzz_lbs.err
