O=gpurun_out/r04
B="--workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline"
python bench.py $B --cell-shape 1 2>/dev/null | grep "^{" > $O/lean_s512.json
python bench.py $B --cell-shape 4 2>/dev/null | grep "^{" > $O/lean_s384.json
SDM_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libsdm_lean256.so python bench.py $B --cell-shape 4 2>/dev/null | grep "^{" > $O/lean_s256.json
python bench.py $B --n-sd 3145728 --cell-shape 1 2>/dev/null | grep "^{" > $O/lean3m_s512.json
python bench.py $B --n-sd 3145728 --cell-shape 4 2>/dev/null | grep "^{" > $O/lean3m_s384.json
