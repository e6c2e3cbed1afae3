set -x
mkdir -p gpurun_out/r03
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --tb=short --timeout=200 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r03/pytest_d.log
tail -5 gpurun_out/r03/pytest_d.log
timeout 600 python bench.py > gpurun_out/r03/bench_d.json 2> gpurun_out/r03/bench_d.err
tail -c 400 gpurun_out/r03/bench_d.json
timeout 600 bash scripts/timeline_run.sh > gpurun_out/r03/timeline_d.log 2>&1
mkdir -p gpurun_out/r03/timeline_d; cp gpurun_out/timeline/*.txt gpurun_out/r03/timeline_d/
head -1 gpurun_out/r03/timeline_d/*.txt
