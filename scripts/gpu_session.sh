set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_full_size.py -m gpu -x -q -k "non_adaptive_variants or many_steps" 2>&1 | grep -v "^  File\|^Extension" | tail -30
