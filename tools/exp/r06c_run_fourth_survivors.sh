# the five survivors of the fourth device batch against the tests written for them (tools/mutants.py run --tests / --k)
python -m pytest tests/test_lbs_gpu.py -m gpu -x -q -k "last_row or one_matrix_short" 2>&1 | tail -3
python tools/mutants.py run --only palette_commit_ignores_m32,dyn_projective_test_looks_at_m33_only,stage_palette_ignores_m31,projective_divides_by_zero_too,palette_one_bone_short_is_accepted --tests tests/test_lbs_gpu.py --k "last_row or one_matrix_short" --out gpurun_out/mutants_fourth_survivors.json 2>&1 | tail -7
