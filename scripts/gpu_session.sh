set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pc_*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc_stats -- python3 bench.py --steps 100 --warmup 10 --reps 1 --no-cpu-baseline --roofline-steps 5 > /dev/null 2>&1
f=$(ls gpurun_out/pc_stats/*/*kernel_stats.csv); head -8 $f | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE TCC_REQ_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pc_$c -- python3 bench.py --steps 30 --warmup 5 --reps 1 --no-cpu-baseline --roofline-steps 5 > /dev/null 2>&1
python tests/pmc_summary.py gpurun_out/pc_$c $c > gpurun_out/r02_pmc_carried_$c.txt
head -4 gpurun_out/r02_pmc_carried_$c.txt
done
rm -rf gpurun_out/pc_*
