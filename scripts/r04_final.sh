# the bench lines of round 4 (library as committed), one GPU: bash scripts/r04_final.sh
O=gpurun_out/r04/final
mkdir -p $O
run() { name=$1; shift; python bench.py "$@" 2>$O/$name.err | grep "^{" > $O/r04_bench_$name.json; }
run shima
run shima_driver_shape --gpus 1 --steps 20 --warmup 5
run shima_adaptive --adaptive 1 --no-cpu-baseline
run berry_breakup --workload berry_breakup --no-cpu-baseline
run straub --workload straub --no-cpu-baseline
run straub_rain --workload straub_rain --no-cpu-baseline
run kinematic2d --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline
run kinematic2d_sharded_on_one_rccl --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline --sharded-on-one
SDM_PYTHON_EXCHANGE=1 python bench.py --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline --sharded-on-one 2>/dev/null | grep "^{" > $O/r04_bench_kinematic2d_sharded_on_one_python_exchange.json
run kinematic2d_emulated_8_ranks --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline --emulate-of 8
run kinematic2d_emulated_4_ranks --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline --emulate-of 4
run kinematic2d_emulated_2_ranks --workload kinematic2d --steps 40 --warmup 5 --no-cpu-baseline --emulate-of 2
run kinematic2d_flow --workload kinematic2d_flow --steps 20 --warmup 3 --no-cpu-baseline
run kinematic2d_flow_sharded_on_one_rccl --workload kinematic2d_flow --steps 20 --warmup 3 --no-cpu-baseline --sharded-on-one
run kinematic2d_1024_per_cell --workload kinematic2d --n-sd 1048576 --steps 40 --warmup 5 --no-cpu-baseline
run kinematic2d_75x75x128 --workload kinematic2d --grid 75 75 --n-sd 720000 --steps 40 --warmup 5 --no-cpu-baseline
ls -la $O | head -40
