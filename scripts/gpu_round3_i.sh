set -x
mkdir -p gpurun_out/r03
timeout 600 python -X faulthandler -m pytest tests -m gpu -q --tb=short --timeout=200 -k "wgrad or conv3d_fwd_dgrad_wgrad or fullsize or discriminator or Discriminator or wgan or WGAN or trajector" 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r03/pytest_i.log
tail -4 gpurun_out/r03/pytest_i.log
python scripts/edge_ab.py > gpurun_out/r03/edge_i.json 2>gpurun_out/r03/edge_i.err
cat gpurun_out/r03/edge_i.json
