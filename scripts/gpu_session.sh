set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest_m.log 2>&1 || { tail -60 gpurun_out/r02_gputest_m.log; exit 1; }
tail -2 gpurun_out/r02_gputest_m.log
for w in berry_breakup straub_rain straub; do
python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_${w}_fold.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_${w}_fold.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
done
python bench.py --adaptive 1 --no-cpu-baseline > gpurun_out/r02_bench_shima_adaptive_fold.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_shima_adaptive_fold.json')); print('shima adaptive', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
