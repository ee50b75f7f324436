# Third session's closing sweep: every mode of tools/fuzz_gpu.py on new seeds (60 000 +)
OUT=gpurun_out/long3; mkdir -p $OUT
python tools/fuzz_gpu.py --first 60000 --count 1500 --bones 9 --out $OUT/a_plain.json 2>&1 | tail -1 | cut -c1-60,330-420
python tools/fuzz_gpu.py --first 62000 --count 1200 --listy --edits --bones 13 --out $OUT/b_listy_edits.json 2>&1 | tail -1 | cut -c1-60,330-420
python tools/fuzz_gpu.py --first 64000 --count 1000 --curves --edits --bones 6 --out $OUT/c_curves_edits.json 2>&1 | tail -1 | cut -c1-60,330-420
python tools/fuzz_gpu.py --first 66000 --count 900 --listy --scene 7 --bones 9 --out $OUT/d_scene_listy.json 2>&1 | tail -1 | cut -c1-60,330-420
python tools/fuzz_gpu.py --first 68000 --count 600 --diverge --listy --lattice --bones 8 --out $OUT/e_diverge_listy_lattice.json 2>&1 | tail -1 | cut -c1-60,330-420
python tools/fuzz_gpu.py --first 70000 --count 500 --skin --listy --lattice --bones 12 --out $OUT/f_skin_listy_lattice.json 2>&1 | tail -1 | cut -c1-60,330-420
python - <<'P'
import json,glob
t=0;f=0
for p in sorted(glob.glob("gpurun_out/long3/*.json")):
    d=json.load(open(p)); t+=d["seeds"]; f+=d["failures"]
print({"machines":t,"failures":f})
P
