out=gpurun_out/full1; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest.log
