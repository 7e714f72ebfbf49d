set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SDM_NO_PRESORT=1 python bench.py --no-cpu-baseline > gpurun_out/exp_p21.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/exp_p21.json')); print('P21', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
SDM_NO_PRESORT=1 SDM_EXPERIMENT_FORCE_P24=1 python bench.py --no-cpu-baseline > gpurun_out/exp_p24.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/exp_p24.json')); print('P24', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
