# round 6: nms_scan_kernel alone (bench.py's roofline_postproc block: one launch behind an evicting fill) and the whole step, by loads per thread
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-scanab}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_postproc.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 2
for i in 1 2; do for cfg in "1 2048" "2 2048" "4 2048" "8 2048" "4 512" "8 256" "1 512"; do set -- $cfg
  SA_NMS_ILP=$1 SA_NMS_GX=$2 timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); p=j['roofline_postproc']; print('ilp $1 gx $2:', 'scan alone', p['avg_launch_ms'], 'ms', p['frac'], '| whole postproc', p['whole_postproc']['ms_per_step'], '| step', j['ms_per_step'], j['value'], j['config']['result_digest'])" | tee -a $O/ab.txt
done; done
