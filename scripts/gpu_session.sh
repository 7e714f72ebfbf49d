set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_hip_full_size.py -m gpu -x -q > gpurun_out/r02_gputest_s.log 2>&1 || { grep -v "^  File\|^Extension" gpurun_out/r02_gputest_s.log | tail -40; exit 1; }
tail -2 gpurun_out/r02_gputest_s.log
for w in berry_breakup; do
python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_${w}_probsort.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_${w}_probsort.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
SDM_NO_PRESORT=1 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_${w}_probsort_off.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_${w}_probsort_off.json')); print('$w off', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
done
python bench.py --adaptive 1 --no-cpu-baseline > gpurun_out/r02_bench_shima_adaptive_probsort.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_shima_adaptive_probsort.json')); print('shima adaptive', d['value'], d['ms_per_step'])"
python bench.py --adaptive 1 --n-sd 65536 --steps 1000 --no-cpu-baseline > gpurun_out/r02_bench_shima_adaptive_n65536_probsort.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_shima_adaptive_n65536_probsort.json')); print('shima adaptive 2^16', d['value'], d['ms_per_step'])"
