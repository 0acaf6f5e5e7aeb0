cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for b in 8 16 32; do for v in 1 0 1 0; do
echo "batch $b fuse64=$v: $(SA_FUSE_PAIRS64=$v python bench.py --no-cpu-baseline --no-extras --batch $b --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['value'])")"
done; done
