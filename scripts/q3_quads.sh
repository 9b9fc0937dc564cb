set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for q in 1 2 4; do
  rm -rf $O/q3q_$q
  PRESTO_AMD_FP_QUADS=$q rocprofv3 --kernel-trace --stats --output-format csv -d $O/q3q_$q -- python3 $R/scripts/bench_q3.py --steps 3 --warmup 1 > $O/q3q_$q.json 2> $O/q3q_$q.err
  echo "quads $q:"; grep -h "pa_fp_\|pa_fused_probe" $O/q3q_$q/*/*_kernel_stats.csv | cut -d, -f1,2,4 
done
