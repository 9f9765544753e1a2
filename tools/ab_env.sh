# Same-box A/B of an environment setting: whole-window frames/s, ViT-L and ViT-S, interleaved.
#   gpurun -- bash tools/ab_env.sh HIP_FORCE_DEV_KERNARG 1 unset
set -e
cd $GRAFT_REPO_ROOT
NAME=$1; A=$2; Bv=$3
B="python3 bench.py --no-end-to-end --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5"
for round in 1 2; do
  for v in $A $Bv; do
    if [ "$v" = unset ]; then env -u $NAME timeout -k 10 200 $B > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    else env $NAME=$v timeout -k 10 200 $B > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err; fi
    python3 -c "import json,sys; d=json.load(open('gpurun_out/ab_tmp.json')); print('round', sys.argv[1], sys.argv[2], 'vitl', d['value'], 'vits', d['vits']['value'], flush=True)" $round "$NAME=$v"
  done
done
