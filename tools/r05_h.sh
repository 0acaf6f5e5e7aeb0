cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05h}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_fp16.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log | cut -c1-200
timeout 900 python tests/diagnostics/fixture_sweep.py 11 8 2>&1 | grep -v amdgpu.ids > $O/fixture_sweep.txt; cat $O/fixture_sweep.txt | cut -c1-330
