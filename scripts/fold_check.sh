O=gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_hip_full_size.py tests/test_hip_parity.py tests/test_hip_fuzz.py -x -q -m gpu > $O/fold_tests.log 2>&1; tail -3 $O/fold_tests.log
for w in berry_breakup straub; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | grep "^{" > $O/fold_$w.json
done
python bench.py --workload shima --adaptive 1 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/fold_shima_adaptive.json
