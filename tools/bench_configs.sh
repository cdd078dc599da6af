#!/bin/bash
# The other BASELINE configs / batch sizes on one box: short bench.py runs (train + inference + encoder legs), one JSON line each.
# usage: tools/bench_configs.sh OUTDIR
OUT=${1:-gpurun_out/configs}; mkdir -p $OUT
run() { name=$1; shift; python bench.py --steps 30 --warmup 6 --no-tile --no-yaml-legs --no-cpu-baseline --no-parity-leg --detail-file $OUT/detail_$name.json "$@" 2>&1 | grep '^{' > $OUT/bench_$name.json; python - <<PY
import json; d=json.load(open("$OUT/bench_$name.json")); c=d["config"]
print(f"$name: train {d['value']:.0f} chips/s ({d['ms_per_step']:.2f} ms, whole-step frac {c['whole_step_mfma_frac']:.3f}), inference {c['inference_chips_per_s']:.0f}, encoder fwd {c['encoder_fwd_ms']:.2f} ms ({c['encoder_fwd_mfma_frac']:.3f}); roofline {d.get('roofline',{}).get('kernel')} {d.get('roofline',{}).get('frac')}")
PY
}
run b16 --batch 16
run b32 --batch 32
run b108 --batch 108
run b112 --batch 112
run b216 --batch 216
run b432 --batch 432
run t3_c13_b36 --temporal 3 --classes 13 --batch 36
run t3_c13_b72 --temporal 3 --classes 13 --batch 72
run t3_c13_b8 --temporal 3 --classes 13 --batch 8
run 300m_b32 --model prithvi_eo_v2_300 --batch 32
run 300m_b54 --model prithvi_eo_v2_300 --batch 54
run 300m_b80 --model prithvi_eo_v2_300 --batch 80
run 300m_b160 --model prithvi_eo_v2_300 --batch 160
run 600m_b32 --model prithvi_eo_v2_600 --batch 32 --steps 10 --warmup 3
