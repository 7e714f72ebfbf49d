set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python tests/soak.py 2>&1 | tee gpurun_out/r02_soak.log
