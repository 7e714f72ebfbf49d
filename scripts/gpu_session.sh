set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest_t.log 2>&1 || { grep -v "^  File\|^Extension" gpurun_out/r02_gputest_t.log | tail -40; exit 1; }
tail -2 gpurun_out/r02_gputest_t.log
for n in 65536 262144 4194304; do
python bench.py --workload kinematic2d --n-sd $n --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_kinematic2d_n${n}_aff.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_kinematic2d_n${n}_aff.json')); print($n, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['repetitions']['values'])"
done
