#!/bin/bash
# same-box A/B of library builds kept under tools/ab/*.so: bench.py once per variant, twice round-robin
L=monocular-depth-estimation_amd/lib/libvda_hip.so
cp $L /tmp/orig.so
for rep in 1 2; do
  for v in "$@"; do
    so=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs=""
    cp tools/ab/$so.so $L
    out=$(env $envs timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-vits $BENCH_ARGS 2>/dev/null | tail -n 1) || exit 1
    echo "$v rep$rep: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d.get("roofline",{}); o=r.get("other_instantiations",{}); print(d["value"], d["ms_per_step"], "dominant_us", r.get("avg_launch_us"), {k.split(" / ")[1]: v["avg_launch_us"] for k, v in o.items() if k.startswith("gemm256") and k.endswith("bf16")})')"
  done
done
cp /tmp/orig.so $L
