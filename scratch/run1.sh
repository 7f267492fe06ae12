cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -k "sub_pixel" 2>&1 | tail -8
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -k "sub_pixel" 2>&1 | tail -1; done
timeout 300 python scratch/subpixel_time.py 2>&1 | tail -4
timeout 900 python scratch/ab_step.py gn3:gn:3 gn7:gn:7 gn1:gn:1 2>&1 | tail -4
