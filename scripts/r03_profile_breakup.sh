set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03/prof2
mkdir -p $OUT
B="--reps 1 --no-cpu-baseline --roofline-steps 5"
for w in berry_breakup straub_rain; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 bench.py --workload $w --steps 60 --warmup 5 $B > $OUT/stats_$w.json 2> $OUT/stats_$w.err
  python tests/prof_summary.py $OUT/stats_$w 9 | tee $OUT/stats_$w.txt
  cp $(ls $OUT/stats_$w/*/*kernel_stats.csv | tail -1) $OUT/r03_kernel_stats_$w.csv
  rm -rf $OUT/stats_$w
done
