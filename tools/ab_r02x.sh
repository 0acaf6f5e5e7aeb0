# small batches: does the mid-chunk issue rule help or hurt at 8 / 16 frames per GPU?
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for b in 8 16; do for v in 1 0; do
  echo "B=$b late_issue=$v $(SA_CONV_LATE_ISSUE=$v timeout 200 python bench.py --batch $b --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], j['roofline']['frac'])")"
done; done; done
