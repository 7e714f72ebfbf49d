# rocprofv3 evidence for the round (run on the GPU box through gpurun): kernel stats of the two
# headline workloads, then one PMC pass per counter (counters never together with the stats)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03/prof
mkdir -p $OUT
B="--reps 1 --no-cpu-baseline --skip-full-experiment --roofline-steps 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_shima -- python3 bench.py --steps 100 --warmup 10 $B > $OUT/stats_shima.json 2> $OUT/stats_shima.err
python tests/prof_summary.py $OUT/stats_shima 8 | tee $OUT/stats_shima.txt
cp $(ls $OUT/stats_shima/*/*kernel_stats.csv | tail -1) $OUT/r03_kernel_stats_shima.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_kin -- python3 bench.py --workload kinematic2d --steps 40 --warmup 5 $B > $OUT/stats_kin.json 2> $OUT/stats_kin.err
python tests/prof_summary.py $OUT/stats_kin 10 | tee $OUT/stats_kin.txt
cp $(ls $OUT/stats_kin/*/*kernel_stats.csv | tail -1) $OUT/r03_kernel_stats_kinematic2d.csv
for c in TCC_HIT_sum TCC_MISS_sum FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_kin_$c -- python3 bench.py --workload kinematic2d --steps 12 --warmup 3 $B > /dev/null 2> $OUT/pmc_kin_$c.err
  python tests/pmc_summary.py $OUT/pmc_kin_$c $c > $OUT/r03_pmc_kinematic2d_$c.txt
  head -3 $OUT/r03_pmc_kinematic2d_$c.txt
done
for c in TCC_HIT_sum TCC_MISS_sum FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_shima_$c -- python3 bench.py --steps 40 --warmup 5 $B > /dev/null 2> $OUT/pmc_shima_$c.err
  python tests/pmc_summary.py $OUT/pmc_shima_$c $c > $OUT/r03_pmc_shima_$c.txt
  head -2 $OUT/r03_pmc_shima_$c.txt
done
rm -rf $OUT/stats_shima $OUT/stats_kin $OUT/pmc_kin_* $OUT/pmc_shima_*/  # (raw traces: too large to keep)
ls $OUT
