#!/bin/bash
# runtime knobs that might change the per-node cost of a graph replay (each variant in its own process)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
python scratch/ab_env.py NOTHING 0 0 1 | head -1
python scratch/ab_env.py HIP_FORCE_DEV_KERNARG 1 0 1
python scratch/ab_env.py DEBUG_CLR_GRAPH_PACKET_CAPTURE 1 0 1
python scratch/ab_env.py GPU_MAX_HW_QUEUES 1 8 1
python scratch/ab_env.py HSA_ENABLE_INTERRUPT 0 1 1
done
