set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_resident.py -x -q -k sppf 2>&1 | tail -15
