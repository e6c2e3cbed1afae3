set -x
mkdir -p gpurun_out/r03
timeout 600 python -X faulthandler -m pytest tests -m gpu -q --tb=short --timeout=200 -k "conv_transpose or convT or transpose3d or fullsize or generator or Generator or autoencoder" 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r03/pytest_g.log
tail -4 gpurun_out/r03/pytest_g.log
python scripts/edge_ab.py > gpurun_out/r03/edge_g_new.json 2>gpurun_out/r03/edge_g.err
SG_EDGE_DEBUG=32 python scripts/edge_ab.py > gpurun_out/r03/edge_g_old.json 2>>gpurun_out/r03/edge_g.err
cat gpurun_out/r03/edge_g_new.json gpurun_out/r03/edge_g_old.json
