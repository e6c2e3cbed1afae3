set -x
export SHAPEGAN_REFERENCE_DIR=$PWD/.refscratch
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -30 > gpurun_out/r03/pytest_b.log
tail -8 gpurun_out/r03/pytest_b.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03/bench_b.json 2> gpurun_out/r03/bench_b.err
tail -3 gpurun_out/r03/bench_b.err
SG_EDGE_DEBUG=16 python scripts/edge_ab.py 2>&1 | tail -1
bash scripts/timeline_run.sh > gpurun_out/r03/timeline_b.log 2>&1
mkdir -p gpurun_out/r03/timeline_b && cp gpurun_out/timeline/*.txt gpurun_out/r03/timeline_b/
