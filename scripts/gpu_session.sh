set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf gpurun_out/pc_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc_stats -- python3 profiles/tools/flow_profile.py 720000 75 75 > gpurun_out/flow.txt 2>&1 || { tail -5 gpurun_out/flow.txt; exit 1; }
tail -1 gpurun_out/flow.txt
cp $(ls gpurun_out/pc_stats/*/*kernel_stats.csv) gpurun_out/r02_kernel_stats_flow_75x75x128.csv
rm -rf gpurun_out/pc_stats
