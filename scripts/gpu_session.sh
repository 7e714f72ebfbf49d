set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r02_gputest_final.log 2>&1 || { grep -v "^  File\|^Extension" $O/r02_gputest_final.log | tail -40; exit 1; }
tail -2 $O/r02_gputest_final.log
python bench.py > $O/r02_bench_shima.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/r02_bench_shima.json')); print('shima', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac'], d['cpu_baseline']['value'])"
