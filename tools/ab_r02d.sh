# round-2 session d: upsample rows-per-thread, head layer with 16 waves
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02d; O=gpurun_out/r02d
timeout 300 python -m pytest tests/test_gpu_network.py tests/test_gpu_persistent.py -q -x 2>&1 | tail -2
SA_CONV_HEAD_NW16=1 timeout 300 python -m pytest tests/test_gpu_network.py tests/test_gpu_persistent.py tests/test_gpu_fp16.py -q -x 2>&1 | tail -2
for i in 1 2; do
for v in base up1 up4 nw16; do
  unset SA_UP_ROWS SA_CONV_HEAD_NW16
  case $v in base) ;; up1) export SA_UP_ROWS=1;; up4) export SA_UP_ROWS=4;; nw16) export SA_CONV_HEAD_NW16=1;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['network_ms_per_step'])"
  grep "^up\|head" $O/layers_${v}_$i.log | cut -c1-62 | tr '\n' ';'; echo
done; done
