# rocprofv3 evidence of round 4 (run on the GPU box through gpurun): kernel statistics of a
# workload's bench line, then one PMC pass per counter (counters never together with the stats).
#   bash scripts/r04_profile.sh WORKLOAD "bench flags" "COUNTERS..."
# The bench's own calibration kernel (k_calib_random) runs in every pass: its counters are the
# like-for-like reference of the roofline object (profiles/tools/traffic_r04.py).
set -e
W=$1; FLAGS=$2; COUNTERS=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04/prof
mkdir -p $OUT
B="--workload $W --reps 1 --no-cpu-baseline --skip-full-experiment --roofline-steps 5 $FLAGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- python3 bench.py $B > $OUT/stats_$W.json 2> $OUT/stats_$W.err
python tests/prof_summary.py $OUT/stats_$W 10 | tee $OUT/stats_$W.txt
cp $(ls $OUT/stats_$W/*/*kernel_stats.csv | tail -1) $OUT/r04_kernel_stats_$W.csv
for c in $COUNTERS; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${W}_$c -- python3 bench.py $B > /dev/null 2> $OUT/pmc_${W}_$c.err
  python tests/pmc_summary.py $OUT/pmc_${W}_$c $c > $OUT/r04_pmc_${W}_$c.txt
  head -4 $OUT/r04_pmc_${W}_$c.txt
done
rm -rf $OUT/stats_$W $OUT/pmc_${W}_*/  # (raw traces: too large to keep)
ls $OUT
