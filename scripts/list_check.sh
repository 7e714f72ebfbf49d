O=gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_hip_full_size.py tests/test_hip_parity.py tests/test_hip_fuzz.py -x -q -m gpu > $O/list_tests.log 2>&1; tail -3 $O/list_tests.log
for w in straub straub_rain berry_breakup; do
  python bench.py --workload $w --no-cpu-baseline 2>$O/list_$w.err | grep "^{" > $O/list_$w.json
done
python bench.py --workload shima --adaptive 1 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/list_shima_adaptive.json
