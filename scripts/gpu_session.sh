set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest_a.log 2>&1 || { tail -30 gpurun_out/r02_gputest_a.log; exit 1; }
tail -3 gpurun_out/r02_gputest_a.log
python bench.py --no-cpu-baseline > gpurun_out/r02_bench_a.json 2>gpurun_out/r02_bench_a.err
cat gpurun_out/r02_bench_a.json
for c in TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 5 > gpurun_out/pmc_$c.log 2>&1 || echo "pmc $c failed"
  python tests/pmc_summary.py gpurun_out/pmc_$c $c > gpurun_out/r02_pmc_$c.txt 2>&1 || true
  cat gpurun_out/r02_pmc_$c.txt | head -8
done
