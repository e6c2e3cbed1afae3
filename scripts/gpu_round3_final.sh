set -x
export SHAPEGAN_REFERENCE_DIR=$PWD/.refscratch
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -12 > gpurun_out/r03/pytest_final.log
tail -4 gpurun_out/r03/pytest_final.log
timeout 900 python -m pytest tests/test_dropin.py -v 2>&1 | grep -E "PASSED|FAILED|SKIPPED|passed|failed" > gpurun_out/r03/dropin_gpu.log
tail -3 gpurun_out/r03/dropin_gpu.log
timeout 600 python bench.py > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err
tail -c 600 gpurun_out/r03/bench_final.json
bash scripts/profile_round.sh r03 > gpurun_out/r03/profile_round.log 2>&1
tail -5 gpurun_out/r03/profile_round.log
