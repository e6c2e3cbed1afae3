mkdir -p gpurun_out/r03
L=gpurun_out/r03/c.log; rm -f $L
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py tests/test_gpu_losses.py -q --tb=short -x -k "sdf or autodecoder or sort or graph" 2>&1 | tail -6 >> $L
python scripts/sdf_train_bench.py >> $L 2>&1
cat $L | grep -v amdgpu.ids
