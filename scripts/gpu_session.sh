set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q > gpurun_out/r02_gputest_h.log 2>&1 || { tail -60 gpurun_out/r02_gputest_h.log; exit 1; }
tail -2 gpurun_out/r02_gputest_h.log
python bench.py --no-cpu-baseline > gpurun_out/r02_bench_ldstab.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_ldstab.json')); print('shima', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
python bench.py --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/exp_k2d_base.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/exp_k2d_base.json')); print('base', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['repetitions'])"
SDM_EXPERIMENT_CELL_SORTED_IDS=1 python bench.py --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/exp_k2d_sorted.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/exp_k2d_sorted.json')); print('sorted ids', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['repetitions'])"
