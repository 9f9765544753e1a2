# Same-box A/B of two builds of the library (VDA_HIP_LIB): whole-window frames/s, ViT-L and ViT-S, interleaved A B A B.
#   gpurun -- bash tools/ab_lib.sh tools/exp/libvda_hip_prev.so
set -e
cd $GRAFT_REPO_ROOT
OTHER=$PWD/$1
B="python3 bench.py --no-end-to-end --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
for round in 1 2; do
  for lib in "" "$OTHER"; do
    VDA_HIP_LIB=$lib timeout -k 10 200 $B > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python3 -c "import json,sys; d=json.load(open('gpurun_out/ab_tmp.json')); print('round', sys.argv[1], 'lib', sys.argv[2] or 'in-tree', 'vitl', d['value'], 'vits', d['vits']['value'], flush=True)" $round "$1$( [ -z "$lib" ] && echo ' (not used)' )" | sed "s|lib $1 (not used)|lib in-tree|"
  done
done
