cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt32
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt32 -o res -- python3 bench.py --precision fp32 --steps 3 --warmup 1 --no-end-to-end --no-cpu-baseline --no-vits --encoder-streams 1 > gpurun_out/fp32_under_rocprof.json 2> gpurun_out/kt32.err
python tools/rocpd_stats.py $(ls gpurun_out/kt32/*.db | tail -n 1) gpurun_out/r2_vitl518_fp32_kernel_stats.csv
rm -rf gpurun_out/kt32
