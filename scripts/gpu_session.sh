set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/exp_k2d_base2.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/exp_k2d_base2.json')); print('base', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['repetitions'])"
SDM_EXPERIMENT_LAZY_SOA=1 python bench.py --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/exp_k2d_lazy.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/exp_k2d_lazy.json')); print('lazy', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['repetitions'])"
python bench.py --workload berry_breakup --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_berry_breakup_lists.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_berry_breakup_lists.json')); print('berry', d['value'], d['ms_per_step'])"
