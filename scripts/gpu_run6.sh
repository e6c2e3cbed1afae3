#!/bin/bash
mkdir -p gpurun_out/r4f
for i in 1 2 3; do python scripts/edge_ab.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('conv1_fwd_128','conv1_wgrad_32','convT_fwd_64','convT_fwd_32','convT_fwd_256','conv1_wgrad_act_128')})"; done
