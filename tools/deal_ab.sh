#!/bin/bash
# Tile dealing of the grouped weight gradients (IG_G8W_DEAL=0: round-3 order, 1: per-XCD rectangles of one token split): tests, whole-step A/B
# and HBM / L2-miss traffic per launch of both arms.  usage: tools/deal_ab.sh OUTDIR
OUT=${1:-gpurun_out/r06_deal}; R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/$OUT
python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad" > $R/$OUT/tests.log 2>&1
bash tools/ab_step.sh 2 "IG_G8W_DEAL=0" "IG_G8W_DEAL=1" > $R/$OUT/ab_b216.log 2>&1
bash tools/ab_step.sh 1 "IG_G8W_DEAL=0" "IG_G8W_DEAL=1" --temporal 3 --classes 13 --batch 72 > $R/$OUT/ab_t3_b72.log 2>&1
for v in 0 1; do
  IG_G8W_DEAL=$v bash tools/pmc_bench.sh $OUT/pmc$v > /dev/null 2>&1
  python3 tools/pmc_summarize.py $R/$OUT/pmc$v > $R/$OUT/pmc_deal$v.json
  python3 - <<PY
import json
d=json.load(open("$R/$OUT/pmc_deal$v.json"))
for k,x in d.items():
    if "gemm8w" in k: print("IG_G8W_DEAL=$v", k, "launches", x["launches"], "traffic per launch %.1f MB" % ((2*x["fetch_kb_raw"]+x["write_kb"])*1024/1e6))
PY
  rm -rf $R/$OUT/pmc$v
done > $R/$OUT/traffic.txt 2>&1
