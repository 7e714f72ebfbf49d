set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
export SDM_EXPERIMENT_HOT_RECORDS=1
timeout -k 10 600 python -m pytest tests/test_hip_full_size.py -m gpu -x -q -k "many_steps or non_adaptive_variants or shima_box_3600 or record_layouts or digests" > $O/r02_gputest_hot.log 2>&1 || { grep -v "^  File\|^Extension" $O/r02_gputest_hot.log | tail -40; exit 1; }
tail -2 $O/r02_gputest_hot.log
python bench.py --no-cpu-baseline > $O/r02_bench_hot.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r02_bench_hot.json')); print('hot', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
SDM_NO_PRESORT=1 python bench.py --no-cpu-baseline > $O/r02_bench_hot_nopresort.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r02_bench_hot_nopresort.json')); print('hot, plain k_pair_all', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
rm -rf $O/pc_*
for c in TCC_REQ_sum TCC_MISS_sum TCC_EA0_RDREQ_sum; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pc_$c -- python3 bench.py --steps 30 --warmup 5 --reps 1 --no-cpu-baseline --roofline-steps 5 > /dev/null 2>&1
python tests/pmc_summary.py $O/pc_$c $c > $O/r02_pmc_hot_records_$c.txt
head -3 $O/r02_pmc_hot_records_$c.txt
done
rm -rf $O/pc_*
unset SDM_EXPERIMENT_HOT_RECORDS
python bench.py --no-cpu-baseline > $O/r02_bench_hot_off.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r02_bench_hot_off.json')); print('off', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_step'])"
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum --output-format csv -d $O/pc_x -- python3 bench.py --steps 30 --warmup 5 --reps 1 --no-cpu-baseline --roofline-steps 5 > /dev/null 2>&1
python tests/pmc_summary.py $O/pc_x TCC_EA0_RDREQ_sum | head -3
rm -rf $O/pc_*
