O=gpurun_out/r04
for v in base tile16k; do
  if [ $v != base ]; then export SDM_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libsdm_$v.so; fi
  for w in straub straub_rain; do
    python bench.py --workload $w --no-cpu-baseline 2>$O/tile_${v}_$w.err | grep "^{" > $O/tile_${v}_$w.json
  done
done
SDM_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libsdm_tile16k.so timeout -k 10 500 python -m pytest tests/test_hip_parity.py -x -q -m gpu > $O/tile16k_tests.log 2>&1; tail -3 $O/tile16k_tests.log
