set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/pc_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pc_stats -- python3 bench.py --workload kinematic2d --grid 75 75 --n-sd 720000 --steps 40 --warmup 5 --reps 1 --no-cpu-baseline --roofline-steps 3 > /dev/null 2>&1
cp $(ls $O/pc_stats/*/*kernel_stats.csv) $O/r02_kernel_stats_kinematic2d_75x75x128_b.csv
rm -rf $O/pc_stats
