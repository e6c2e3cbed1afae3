out=gpurun_out/sdf1; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py -x -q -m gpu -k "sdf or hybrid" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
bash scripts/ab_run.sh base | grep -v "^=="
grep finish gpurun_out/ab/base.txt
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl20 -o t20 -- python $OLDPWD/scripts/sdf_step_prof.py 20000 128 > /dev/null 2>&1; f=$(find /tmp/tl20 -name "*kernel_trace.csv" | head -1); python $OLDPWD/scripts/step_timeline.py $f adam_kernel 2 | grep -E "finish|step|busy"
