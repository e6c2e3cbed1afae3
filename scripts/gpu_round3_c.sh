mkdir -p gpurun_out/r03
L=gpurun_out/r03/c.log; rm -f $L
export HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3
timeout 420 python -X faulthandler -m pytest tests/test_gpu_fullsize.py -q --tb=short -x -v --timeout=100 2>&1 | grep -v "PASSED" | head -150 >> $L
cat $L | grep -v amdgpu.ids | cut -c1-200 | grep -v "site-packages\|dist-packages" | head -80
