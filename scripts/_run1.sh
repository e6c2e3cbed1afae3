out=gpurun_out/sdf1; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py tests/test_gpu_losses.py tests/test_gpu_dp.py -x -q -m gpu -k "sdf or hybrid or normals or deepsdf or bad_batch" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
python scripts/_sdfnum.py 2>/dev/null | grep -E "mpoints|frac|ms_per|\"train|fwd_"
