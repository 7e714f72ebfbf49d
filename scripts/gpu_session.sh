set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_sharded.py -m gpu -x -q > gpurun_out/r02_gputest_shard.log 2>&1 || { tail -60 gpurun_out/r02_gputest_shard.log; exit 1; }
tail -3 gpurun_out/r02_gputest_shard.log
for w in kinematic2d straub_rain berry_breakup straub; do
  python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_$w.json 2>gpurun_out/r02_bench_$w.err || { tail -20 gpurun_out/r02_bench_$w.err; exit 1; }
  cat gpurun_out/r02_bench_$w.json | cut -c1-900
done
# N = 2 rehearsal on one card (gloo between the ranks): launch contract + sharded bit-identity
export SDM_BENCH_DIST_BACKEND=gloo SDM_BENCH_ALL_ON_DEVICE0=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_rehearsal_n2_kinematic2d.json 2>gpurun_out/r02_rehearsal_n2.err || { tail -30 gpurun_out/r02_rehearsal_n2.err; exit 1; }
cat gpurun_out/r02_rehearsal_n2_kinematic2d.json | cut -c1-1200
unset SDM_BENCH_DIST_BACKEND SDM_BENCH_ALL_ON_DEVICE0
for w in kinematic2d straub_rain; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$w -- python3 bench.py --workload $w --steps 20 --warmup 3 --reps 1 --no-cpu-baseline --roofline-steps 5 > gpurun_out/prof_$w.log 2>&1 || echo "rocprof $w failed"
  f=$(ls gpurun_out/prof_$w/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r02_kernel_stats_$w.csv; head -8 $f | cut -c1-160
done
