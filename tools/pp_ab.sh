# round 6: the fused post-processing kernel at 8 and 64 frames per step, new against old build, by kernel trace (+ exactness tests)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=$R/sleap_amd/lib; O=$R/gpurun_out/${1:-ppab}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_paf_grouping_ref.py tests/test_gpu_benchmark_parity.py tests/test_gpu_config_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 3 | tee $O/tests.txt
for bsz in 8 64; do for i in 1 2; do for v in base ${PP_ALT:-libsleap_amd_fp16_pp0.so}; do
  if [ $v = base ]; then unset SLEAP_AMD_LIB_FP16; else export SLEAP_AMD_LIB_FP16=$L/$v; fi
  rm -rf $O/kt; timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python bench.py --no-cpu-baseline --no-extras --batch $bsz --steps $((bsz == 8 ? 200 : 40)) > $O/kt.log 2>&1
  db=$(find $O/kt -name "*.db" | head -1); python tools/rocpd_stats.py $db > $O/stats_${bsz}_${v}_$i.md
  echo "batch $bsz $v $i $(grep -o '"value": [0-9.]*' $O/kt.log | head -1) $(grep -o '"result_digest": "[0-9a-f]*"' $O/kt.log) postproc: $(grep bottomup_postproc $O/stats_${bsz}_${v}_$i.md | cut -d'|' -f3-7)" | tee -a $O/ab.txt
done; done; done
rm -rf $O/kt
