set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_hip_full_size.py -m gpu -x -q > gpurun_out/r02_gputest_r.log 2>&1 || { grep -v "^  File\|^Extension" gpurun_out/r02_gputest_r.log | tail -40; exit 1; }
tail -2 gpurun_out/r02_gputest_r.log
for w in straub straub_rain; do
python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_${w}_lds.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_${w}_lds.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
done
python bench.py --no-cpu-baseline --n-sd 4194304 > gpurun_out/r02_bench_shima_n4194304.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_shima_n4194304.json')); print('shima 2^22', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
python bench.py --no-cpu-baseline > gpurun_out/r02_bench_shima_lds.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_shima_lds.json')); print('shima', d['value'], d['ms_per_step'])"
