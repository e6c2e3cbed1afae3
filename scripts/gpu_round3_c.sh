mkdir -p gpurun_out/r03
L=gpurun_out/r03/c.log; rm -f $L
timeout 700 python -X faulthandler -m pytest tests -m gpu -q --tb=short -x --timeout=150 2>&1 | tail -8 >> $L
cat $L | grep -v amdgpu.ids | cut -c1-200
