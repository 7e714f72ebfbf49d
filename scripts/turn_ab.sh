cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
for v in g8 g1; do
  if [ $v = g1 ]; then export SDM_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libsdm_turn1.so; fi
  rocprofv3 --kernel-trace -d $O/turn_$v -o t -- python3 bench.py --workload kinematic2d --emulate-of 8 --emulate-ranks 0 --steps 40 --warmup 5 --no-cpu-baseline > $O/turn_$v.json 2> $O/turn_$v.err || exit 1
  python profiles/tools/trace_summary.py $(find $O/turn_$v -name "*.db" | head -1) > $O/turn_$v.txt
  python bench.py --workload kinematic2d --emulate-of 8 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/emul_$v.json
  python bench.py --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/kin_$v.json
done
grep -h "k_cells_turn\|k_cell_step2" $O/turn_g8.txt $O/turn_g1.txt
