set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_full_size.py tests/test_hip_parity.py -m gpu -x -q -k "many_steps or 3600 or c_abi or record_layouts or trajectories or digests" > gpurun_out/r02_gputest_d.log 2>&1 || { tail -60 gpurun_out/r02_gputest_d.log; exit 1; }
tail -3 gpurun_out/r02_gputest_d.log
python bench.py --no-cpu-baseline > gpurun_out/r02_bench_d.json 2>gpurun_out/r02_bench_d.err || { tail -20 gpurun_out/r02_bench_d.err; exit 1; }
cut -c1-700 gpurun_out/r02_bench_d.json
for n in 16384 65536 262144; do python bench.py --no-cpu-baseline --n-sd $n --steps 1000 > gpurun_out/r02_bench_d_$n.json 2>/dev/null; cut -c1-330 gpurun_out/r02_bench_d_$n.json; echo; done
SDM_NO_GRAPH=1 python bench.py --no-cpu-baseline --n-sd 65536 --steps 1000 > gpurun_out/r02_bench_d_65536_nograph.json 2>/dev/null; cut -c1-330 gpurun_out/r02_bench_d_65536_nograph.json
