cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02z; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_topdown.py tests/test_gpu_inference.py tests/test_gpu_network_pin.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/predict_e2e_topdown.py 1024 64 resident 2>&1 | grep "frames/s" | tail -5 | cut -c1-120
timeout 300 python tools/predict_e2e_topdown.py 512 16 2>&1 | grep "frames/s" | tail -2 | cut -c1-120
