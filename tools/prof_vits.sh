cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ktv
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ktv -o res -- python3 bench.py --encoder vits --steps 20 --warmup 3 --no-end-to-end --no-cpu-baseline --no-vits --encoder-streams 1 > gpurun_out/vits_under_rocprof.json 2> gpurun_out/ktv.err
python tools/rocpd_stats.py $(ls gpurun_out/ktv/*.db | tail -n 1) gpurun_out/vits_kernel_stats_now.csv
rm -rf gpurun_out/ktv
