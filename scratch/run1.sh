cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_norm_attn.py -x -q -k "groupnorm" 2>&1 | tail -4
timeout 300 python scratch/gn_slice_time.py 2>&1 | tail -10
timeout 900 python scratch/ab_step.py base:gn:1 slice:gn:3 2>&1 | tail -3
