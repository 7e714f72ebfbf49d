set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest_x.log 2>&1 || { grep -v "^  File\|^Extension" gpurun_out/r02_gputest_x.log | tail -40; exit 1; }
tail -2 gpurun_out/r02_gputest_x.log
python bench.py --workload kinematic2d --grid 75 75 --n-sd 720000 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_kinematic2d_75x75x128.json 2>gpurun_out/err.txt || { tail -5 gpurun_out/err.txt; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_kinematic2d_75x75x128.json')); print('75x75x128', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['phase_ms_per_step'])"
python bench.py --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_kinematic2d_b.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_kinematic2d_b.json')); print('32x32x4096', d['value'], d['ms_per_step'], d['repetitions']['values'], d['roofline']['phase_ms_per_step'], d['state_digest'][:16])"
