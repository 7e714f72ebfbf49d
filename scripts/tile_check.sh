O=gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_hip_full_size.py tests/test_hip_parity.py -x -q -m gpu > $O/tile_tests.log 2>&1; tail -3 $O/tile_tests.log
for w in straub straub_rain shima berry_breakup; do
  python bench.py --workload $w --no-cpu-baseline 2>$O/tile2_$w.err | grep "^{" > $O/tile2_$w.json
done
python bench.py --workload shima --adaptive 1 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/tile2_shima_adaptive.json
python bench.py --workload shima --adaptive 1 --n-sd 4194304 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/tile2_shima_adaptive_22.json
SDM_REC_FORMAT=records python bench.py --workload shima --adaptive 1 --n-sd 4194304 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/tile2_shima_adaptive_22_records.json
