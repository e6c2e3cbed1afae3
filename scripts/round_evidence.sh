#!/bin/bash
# Everything a round's evidence needs from ONE GPU box:  bash scripts/round_evidence.sh r04
#   the whole GPU test tier, the executed drop-in (only when a scratch copy of the reference's training scripts was placed in
#   .refscratch/ — git-ignored, never committed), then scripts/profile_round.sh (rocprofv3 stats / counters, timelines, bench lines)
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p $out
timeout 1800 python -m pytest tests -q -m gpu > $out/${tag}_pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?" | tee -a $out/${tag}_pytest_gpu.log
tail -3 $out/${tag}_pytest_gpu.log
if [ -f .refscratch/train_wgan.py ]; then
  SHAPEGAN_REFERENCE_DIR=$PWD/.refscratch timeout 900 python -m pytest tests/test_dropin.py -v > $out/${tag}_dropin_gpu.log 2>&1
  echo "drop-in rc=$?" | tee -a $out/${tag}_dropin_gpu.log
  tail -2 $out/${tag}_dropin_gpu.log
fi
bash scripts/profile_round.sh $tag
bash scripts/round_extras.sh $tag
