cp fyrox_amd/libfyrox_hip.so /tmp/good.so
cp fyrox_amd/libfyrox_hip_nofix.so fyrox_amd/libfyrox_hip.so
echo "--- without the fix (expected: failures on root motion)"; python -m pytest tests/test_anim_gpu.py -q -k "duplicate_bindings" 2>&1 | tail -6 | cut -c1-300
cp /tmp/good.so fyrox_amd/libfyrox_hip.so
echo "--- with the fix"; python -m pytest tests/test_anim_gpu.py tests/test_frame_skin_gpu.py -q -x 2>&1 | tail -3
