# Kernel-trace statistics of the flagship window with the encoder's LayerNorms folded into qkv / fc1 (1) and as separate
# passes (0), one encoder stream, same box:  gpurun -- bash tools/ab_fold.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-end-to-end --no-cpu-baseline --no-vits --no-kernel-timing --encoder-streams 1"
for v in 1 0; do
  rm -rf gpurun_out/kt
  export VDA_FOLD_LAYERNORM=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o res -- $B --steps 5 --warmup 2 > gpurun_out/fold${v}_bench_under_rocprof.json 2> gpurun_out/kt.err
  python tools/rocpd_stats.py $(ls gpurun_out/kt/*.db | tail -n 1) gpurun_out/fold${v}_kernel_stats.csv
done
rm -rf gpurun_out/kt
