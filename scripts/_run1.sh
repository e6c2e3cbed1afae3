timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py -x -q -m gpu -k "convT or conv_transpose or wgan or generator or gan" 2>&1 | tail -2
for i in 1 2 3; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
mkdir -p gpurun_out/r06; python scripts/edge_cold.py convT_forms > gpurun_out/r06/r06_convT_forms.json
bash scripts/convt_pmc.sh > gpurun_out/r06/r06_convT_c1_counters.txt 2>&1
