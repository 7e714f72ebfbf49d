set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest_b.log 2>&1 || { tail -60 gpurun_out/r02_gputest_b.log; exit 1; }
tail -3 gpurun_out/r02_gputest_b.log
python bench.py > gpurun_out/r02_bench_b.json 2>gpurun_out/r02_bench_b.err || { tail -20 gpurun_out/r02_bench_b.err; exit 1; }
cat gpurun_out/r02_bench_b.json
