set -x
mkdir -p gpurun_out/r03
python scripts/edge_ab.py > gpurun_out/r03/edge_h_base.json 2>gpurun_out/r03/edge_h.err
SHAPEGAN_HIP_LIB=$PWD/scripts/_abl/w2.so python scripts/edge_ab.py > gpurun_out/r03/edge_h_w2.json 2>>gpurun_out/r03/edge_h.err
cat gpurun_out/r03/edge_h_base.json gpurun_out/r03/edge_h_w2.json
