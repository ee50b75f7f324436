# Third session: fuzz under the launch-form options and with subnormal values (lattice mode on: comparisons at equality)
set -u
OUT=gpurun_out/r06i; mkdir -p $OUT
python tools/fuzz_gpu.py --first 50000 --count 600 --subnormal --bones 9 --out $OUT/fuzz_subnormal.json 2>&1 | tail -1 | cut -c1-330
python tools/fuzz_gpu.py --first 51000 --count 500 --subnormal --listy --lattice --edits --bones 8 --out $OUT/fuzz_subnormal_listy_lattice_edits.json 2>&1 | tail -1 | cut -c1-330
python tools/fuzz_gpu.py --first 52000 --count 500 --lattice --opt anim.one_launch=0 --bones 9 --out $OUT/fuzz_lattice_one_launch_off.json 2>&1 | tail -1 | cut -c1-330
python tools/fuzz_gpu.py --first 53000 --count 500 --lattice --listy --opt anim.one_launch=0 --opt anim.inline_ctrl=0 --opt anim.update_lean=0 --bones 9 --out $OUT/fuzz_lattice_separate_launches_uploaded_block.json 2>&1 | tail -1 | cut -c1-330
python tools/fuzz_gpu.py --first 54000 --count 480 --lattice --scene 6 --opt anim.overlap=1 --bones 8 --out $OUT/fuzz_lattice_scene_overlap.json 2>&1 | tail -1 | cut -c1-330
python tools/fuzz_gpu.py --first 55000 --count 300 --subnormal --skin --bones 10 --out $OUT/fuzz_subnormal_skin.json 2>&1 | tail -1 | cut -c1-330
