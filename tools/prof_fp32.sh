# kernel-trace statistics of the fp32 storage mode (ViT-L 518x518 window, encoder on one stream)
set -e
P=${1:-r3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o res -- python3 bench.py --precision fp32 --no-end-to-end --no-cpu-baseline --no-vits --encoder-streams 1 --steps 3 --warmup 1 > gpurun_out/${P}_vitl518_fp32_bench_under_rocprof.json 2> gpurun_out/kt.err
python tools/rocpd_stats.py $(ls gpurun_out/kt/*.db | tail -n 1) gpurun_out/${P}_vitl518_fp32_kernel_stats.csv
rm -rf gpurun_out/kt
