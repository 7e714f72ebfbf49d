set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_full_size.py -m gpu -x -q > gpurun_out/r02_gputest_l.log 2>&1 || { tail -60 gpurun_out/r02_gputest_l.log; exit 1; }
tail -2 gpurun_out/r02_gputest_l.log
python bench.py --no-cpu-baseline > gpurun_out/r02_bench_fastbuild3.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_fastbuild3.json')); print('shima', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
rm -rf gpurun_out/pc_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc_stats -- python3 bench.py --steps 100 --warmup 10 --reps 1 --no-cpu-baseline --roofline-steps 5 > /dev/null 2>&1
f=$(ls gpurun_out/pc_stats/*/*kernel_stats.csv); cp $f gpurun_out/r02_kernel_stats_shima_fast_build.csv; head -6 $f | cut -c1-60,200-300
rm -rf gpurun_out/pc_stats
for w in berry_breakup straub_rain; do
python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_${w}_fast_build.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_${w}_fast_build.json')); print('$w', d['value'], d['ms_per_step'])"
done
