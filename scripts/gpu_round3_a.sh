set -x
export SHAPEGAN_REFERENCE_DIR=$PWD/.refscratch
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -150 > gpurun_out/r03/pytest_a.log
tail -5 gpurun_out/r03/pytest_a.log
timeout 300 python -X faulthandler bench.py --no-extras --no-cpu-baseline > gpurun_out/r03/bench_core.json 2> gpurun_out/r03/bench_core.err
cat gpurun_out/r03/bench_core.json | cut -c1-400
timeout 600 python -X faulthandler bench.py > gpurun_out/r03/bench_a.json 2> gpurun_out/r03/bench_a.err
tail -c 1500 gpurun_out/r03/bench_a.json; tail -30 gpurun_out/r03/bench_a.err
