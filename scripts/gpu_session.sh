set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/r02_bench_shima.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r02_bench_shima.json')); print('shima', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])"
rm -rf $O/pc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pc_stats -- python3 bench.py --steps 100 --warmup 10 --reps 1 --no-cpu-baseline --roofline-steps 5 > /dev/null 2>&1
cp $(ls $O/pc_stats/*/*kernel_stats.csv) $O/r02_kernel_stats_shima.csv
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pc_$c -- python3 bench.py --steps 30 --warmup 5 --reps 1 --no-cpu-baseline --roofline-steps 5 > /dev/null 2>&1
python tests/pmc_summary.py $O/pc_$c $c > $O/r02_pmc_shima_$c.txt
head -3 $O/r02_pmc_shima_$c.txt
done
rm -rf $O/pc_*
for w in berry_breakup straub straub_rain kinematic2d; do
python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > $O/r02_bench_$w.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r02_bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pc_stats -- python3 bench.py --workload $w --steps 20 --warmup 3 --reps 1 --no-cpu-baseline --roofline-steps 5 > /dev/null 2>&1
cp $(ls $O/pc_stats/*/*kernel_stats.csv) $O/r02_kernel_stats_$w.csv
rm -rf $O/pc_stats
done
python bench.py --adaptive 1 --no-cpu-baseline > $O/r02_bench_shima_adaptive.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r02_bench_shima_adaptive.json')); print('shima adaptive', d['value'], d['ms_per_step'])"
