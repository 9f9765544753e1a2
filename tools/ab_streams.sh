#!/bin/bash
# A/B on one box: the encoder's frame groups on 1, 2 and 3 HIP streams (same binary), interleaved rounds.
cd "$(dirname "$0")/.."
for round in 1 2; do
  for n in 1 2 3; do
    echo "== VDA_ENCODER_STREAMS=$n round $round"
    VDA_ENCODER_STREAMS=$n timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-kernel-timing "$@" || exit 1
  done
done
