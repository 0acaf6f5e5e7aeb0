# round 3, GPU call 1: the -m gpu suite on the persistent stem16 / convpair kernels + new layer pins, the bench line, and an A/B
# of the persistent kernels against the round-2 ones on ONE box (alternating runs; box-to-box variance is +-3 %).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -n 15 $O/pytest.log
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.log; cut -c1-1500 $O/bench_line.json; tail -n 3 $O/bench_err.log
for i in 1 2; do
for v in new oldpair oldstem old; do
  unset SA_CONVPAIR_PERSIST SA_STEM16_PERSIST
  case $v in oldpair) export SA_CONVPAIR_PERSIST=0;; oldstem) export SA_STEM16_PERSIST=0;; old) export SA_CONVPAIR_PERSIST=0 SA_STEM16_PERSIST=0;; esac
  timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2> $O/layers_${v}_$i.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$v', j['value'], j['ms_per_step'], j['roofline']['network_ms_per_step'], j['roofline']['frac'], j['roofline']['frac_forward'])" | tee -a $O/ab.txt
  grep -E "stem\+conv|pair" $O/layers_${v}_$i.log | tee -a $O/ab.txt
done; done
unset SA_CONVPAIR_PERSIST SA_STEM16_PERSIST
for n in 1 2 3 5; do SA_STEM16_PERSIST=$n timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "stem\+conv" | sed "s/^/stem persist=$n /" | tee -a $O/ab.txt; done
for n in 1 2; do SA_CONVPAIR_PERSIST=$n timeout 200 python bench.py --layers --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "pair" | sed "s/^/pair persist=$n /" | tee -a $O/ab.txt; done
