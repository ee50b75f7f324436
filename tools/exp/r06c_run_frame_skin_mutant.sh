python -m pytest tests/test_frame_skin_gpu.py -m gpu -x -q -k "homogeneous" 2>&1 | tail -3
python tools/mutants.py run --only frame_skin_projective_test_looks_at_m33_only --tests tests/test_frame_skin_gpu.py --k "homogeneous" --out gpurun_out/mutants_frame_skin.json 2>&1 | tail -2
python tools/mutants.py run --only frame_skin_projective_test_looks_at_m33_only --tests tests/test_frame_skin_gpu.py --k "not homogeneous" --out gpurun_out/mutants_frame_skin_without_the_new_test.json 2>&1 | tail -2
