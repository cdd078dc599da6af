#!/bin/bash
# Rebuilds the 4-wave weight-gradient kernel (gemm4w, gemm8w.hip) with different generator configurations (csrc/gen_gemm4.py w_* keys) ON THE GPU
# BOX and times each against the 8-wave kernel (tools/wgrad_bench.py).  usage: tools/gemm4w_sweep.sh M "cfg1" "cfg2" ...
cd "$(dirname "$0")/../instageo-e2e-geospatial-ml_amd/csrc" || exit 1
M=$1; shift
for cfg in "$@"; do
  echo "=== G4_CFG = $cfg"
  rm -f gemm4_gen.inc gemm8w.o gemm4.o
  make -s G4_CFG="$cfg" >/dev/null 2>&1 || { echo "build failed"; continue; }
  python ../../tools/wgrad_bench.py $M 2>&1 | grep "grouped"
done
rm -f gemm4_gen.inc gemm8w.o gemm4.o
make -s >/dev/null 2>&1
