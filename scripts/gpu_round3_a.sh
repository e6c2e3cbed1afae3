set -x
export SHAPEGAN_REFERENCE_DIR=$PWD/.refscratch
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r03/pytest_a.log
tail -5 gpurun_out/r03/pytest_a.log
timeout 600 python bench.py > gpurun_out/r03/bench_a.json 2> gpurun_out/r03/bench_a.err
tail -c 3000 gpurun_out/r03/bench_a.json; tail -5 gpurun_out/r03/bench_a.err
timeout 300 python scripts/stream_calibration.py > gpurun_out/r03/stream_calibration.json 2> gpurun_out/r03/stream.err; tail -3 gpurun_out/r03/stream.err
