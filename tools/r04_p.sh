# round 4, GPU call P: configs[4] end to end (identical instance assignments) on eight seeds the model was not fitted to
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04p}; mkdir -p $O; cd $R
SA_C4_SEEDS=304,305,306,307,308,309,310,311 timeout 1500 python -m pytest tests/test_gpu_config_parity.py -m gpu -q -s -k "configs4" -p no:cacheprovider > $O/c4_seeds.log 2>&1
grep -E "configs\[4\]|passed|failed|FAILED|Error" $O/c4_seeds.log | cut -c1-400
