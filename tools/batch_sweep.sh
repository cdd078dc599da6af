#!/bin/bash
# Training / inference throughput against the per-GPU batch on one box: tools/batch_sweep.sh OUTDIR B1 B2 ...
OUT=${1:-gpurun_out/sweep}; shift; mkdir -p $OUT
for b in "$@"; do
  python bench.py --batch $b --steps 30 --warmup 6 --no-tile --no-cpu-baseline --no-parity-leg --no-profile --detail-file $OUT/detail_b$b.json 2>&1 | grep '^{' > $OUT/bench_b$b.json
  python - <<PY
import json; d=json.load(open("$OUT/bench_b$b.json")); c=d["config"]
print(f"B=$b: train {d['value']:.0f} chips/s ({d['ms_per_step']:.2f} ms, {d['ms_per_step']/$b*108:.2f} ms per 108), whole-step frac {c['whole_step_mfma_frac']:.3f}, inference {c['inference_chips_per_s']:.0f}, encoder fwd {c['encoder_fwd_ms']:.2f} ms ({c['encoder_fwd_mfma_frac']:.3f})")
PY
done
