set -e
cd $GRAFT_REPO_ROOT
P=${1:-full}
timeout -k 10 150 python tools/carry_equal.py > gpurun_out/${P}_carry_equal.log 2>&1
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/${P}_gpu_tests.log 2>&1
