#!/bin/bash
# Same-box A/B of the whole training step: tools/ab_step.sh ROUNDS "ENV_A" "ENV_B" [bench args]   e.g. "IG_WGRAD8=0" "IG_WGRAD8=1"
# Interleaved rounds of `bench.py` (train + inference legs only); prints chips/s, ms/step, encoder forward ms per run.
R=$1; A=$2; B=$3; shift 3
for r in $(seq $R); do
  for v in "$A" "$B"; do
    printf "%-40s " "$v"
    env $v python bench.py --steps 100 --warmup 10 --no-parity-leg --no-tile --no-yaml-legs --no-cpu-baseline --no-profile --detail-file /tmp/ab_detail.json "$@" 2>&1 \
      | grep -o "\"value\": [0-9.]*, \|ms_per_step\": [0-9.]*\|encoder_fwd_ms\": [0-9.]*\|inference_chips_per_s\": [0-9.]*" | tr "\n" " "
    echo
  done
done
