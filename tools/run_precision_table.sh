set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/tile_equal.py > gpurun_out/tile_equal.log 2>&1
timeout -k 10 900 python tools/precision_table.py > gpurun_out/precision_table.log 2>&1
