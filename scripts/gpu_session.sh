set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest_o.log 2>&1 || { tail -60 gpurun_out/r02_gputest_o.log; exit 1; }
tail -2 gpurun_out/r02_gputest_o.log
python bench.py --no-cpu-baseline > gpurun_out/r02_bench_presort2.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_presort2.json')); print('presort', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
python bench.py --no-cpu-baseline --steps 3600 --warmup 20 --reps 1 > gpurun_out/r02_bench_shima_3600steps.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_shima_3600steps.json')); print('3600', d['value'], d['ms_per_step'])"
