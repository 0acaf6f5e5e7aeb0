# round 3: fit the configs[4] ResNet-50 task model on the GPU box (plain torch; tools/train_config_models.py --device cuda), bring
# the weights back through gpurun_out/, and run the configs[4] / configs[2] parity tests on them in the same call
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03k}; mkdir -p $O; cd $R
timeout 1500 python tools/train_config_models.py c4_resnet --device cuda --steps ${2:-5000} --batch ${3:-16} --out-dir $O > $O/train.log 2>&1; tail -n 4 $O/train.log
cp $O/config_c4_resnet.npz sleap_amd/data/config_c4_resnet.npz
timeout 1200 python -m pytest tests/test_gpu_config_parity.py -m gpu -q -s -p no:cacheprovider > $O/pytest.log 2>&1; tail -n 25 $O/pytest.log | cut -c1-220
