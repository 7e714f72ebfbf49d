set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r02_gputest_final.log 2>&1 || { grep -v "^  File\|^Extension" $O/r02_gputest_final.log | tail -40; exit 1; }
tail -2 $O/r02_gputest_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/r02_bench_shima.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r02_bench_shima.json')); print('shima', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac'])"
for w in berry_breakup straub straub_rain; do
python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > $O/r02_bench_$w.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r02_bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
done
python bench.py --adaptive 1 --no-cpu-baseline > $O/r02_bench_shima_adaptive.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r02_bench_shima_adaptive.json')); print('shima adaptive', d['value'], d['ms_per_step'])"
export SDM_BENCH_DIST_BACKEND=gloo SDM_BENCH_ALL_ON_DEVICE0=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline 2>$O/r02_rehearsal_n2.err | grep '^{' > $O/r02_rehearsal_n2_kinematic2d_gloo_one_gpu.json || { tail -30 $O/r02_rehearsal_n2.err; exit 1; }
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline 2>$O/r02_rehearsal_n2s.err | grep '^{' > $O/r02_rehearsal_n2_shima_gloo_one_gpu.json || { tail -30 $O/r02_rehearsal_n2s.err; exit 1; }
unset SDM_BENCH_DIST_BACKEND SDM_BENCH_ALL_ON_DEVICE0
python bench.py --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline > $O/r02_bench_kinematic2d.json 2>/dev/null
python -c "
import json
a=json.load(open('$O/r02_bench_kinematic2d.json')); b=json.load(open('$O/r02_rehearsal_n2_kinematic2d_gloo_one_gpu.json')); c=json.load(open('$O/r02_rehearsal_n2_shima_gloo_one_gpu.json'))
print('k2d', a['value'], a['state_digest'][:16], 'N=2', b['value'], b['state_digest'][:16], a['state_digest']==b['state_digest']); print('shima N=2', c['value'], c['n_gpus'])"
