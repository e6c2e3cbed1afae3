set -x
mkdir -p gpurun_out/r03
timeout 600 python -X faulthandler -m pytest tests -m gpu -q --tb=short --timeout=200 -k "segmax or colsum or layernorm or point or Point or scatter_max or sdf_batch_sort" 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r03/pytest_e.log
tail -5 gpurun_out/r03/pytest_e.log
timeout 300 python scripts/point_gan_bench.py > gpurun_out/r03/point_gan_e.txt 2>&1
cat gpurun_out/r03/point_gan_e.txt | cut -c1-200
