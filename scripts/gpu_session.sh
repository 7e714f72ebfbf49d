set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_hip_full_size.py -m gpu -x -q > gpurun_out/r02_gputest_p.log 2>&1 || { tail -60 gpurun_out/r02_gputest_p.log; exit 1; }
tail -2 gpurun_out/r02_gputest_p.log
python bench.py --no-cpu-baseline > gpurun_out/r02_bench_quad.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_quad.json')); print('quad', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
for n in 16384 65536 262144 524288; do
python bench.py --no-cpu-baseline --n-sd $n --steps 1000 --warmup 50 > gpurun_out/r02_bench_quad_n$n.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_quad_n$n.json')); print($n, d['value'], d['ms_per_step'])"
done
