#!/bin/bash
# same-box A/B of library builds kept under tools/ab/*.so: bench.py once per variant, twice round-robin
L=monocular-depth-estimation_amd/lib/libvda_hip.so
cp $L /tmp/orig.so
for rep in 1 2; do
  for v in "$@"; do
    so=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs=""
    cp tools/ab/$so.so $L
    out=$(env $envs timeout -k 10 200 python bench.py --steps 5 --warmup 2 2>/dev/null | tail -n 1) || exit 1
    echo "$v rep$rep: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
cp /tmp/orig.so $L
