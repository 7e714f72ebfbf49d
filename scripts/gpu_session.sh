set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest_e.log 2>&1 || { tail -60 gpurun_out/r02_gputest_e.log; exit 1; }
tail -3 gpurun_out/r02_gputest_e.log
export SDM_BENCH_DIST_BACKEND=gloo SDM_BENCH_ALL_ON_DEVICE0=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_rehearsal_n2_kinematic2d.json 2>gpurun_out/r02_rehearsal_n2.err || { tail -30 gpurun_out/r02_rehearsal_n2.err; exit 1; }
cut -c1-500 gpurun_out/r02_rehearsal_n2_kinematic2d.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r02_rehearsal_n2_shima.json 2>gpurun_out/r02_rehearsal_n2s.err || { tail -30 gpurun_out/r02_rehearsal_n2s.err; exit 1; }
cut -c1-400 gpurun_out/r02_rehearsal_n2_shima.json
unset SDM_BENCH_DIST_BACKEND SDM_BENCH_ALL_ON_DEVICE0
python bench.py --workload kinematic2d --steps 40 --warmup 5 > gpurun_out/r02_bench_kinematic2d_cpu.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_kinematic2d_cpu.json')); print(d['value'], d['cpu_baseline'])"
