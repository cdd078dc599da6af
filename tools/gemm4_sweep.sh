#!/bin/bash
# Rebuilds the library's 4-wave GEMM (gemm4.hip) with different generator configurations (csrc/gen_gemm4.py key=value ...) ON THE GPU BOX and
# times each against the 8-phase engine in one process (tools/gemm8_bench.py --v4).  usage: tools/gemm4_sweep.sh M "cfg1" "cfg2" ...
cd "$(dirname "$0")/../instageo-e2e-geospatial-ml_amd/csrc" || exit 1
M=$1; shift
for cfg in "$@"; do
  echo "=== G4_CFG = $cfg"
  rm -f gemm4_gen.inc gemm4.o
  make -s G4_CFG="$cfg" >/dev/null 2>&1 || { echo "build failed"; continue; }
  python ../../tools/gemm8_bench.py $M --v4 2>&1 | grep -E "qkv|proj|fc2 \(|d_fc1|fc1 \(gelu \+|d_fc2"
done
rm -f gemm4_gen.inc gemm4.o
make -s >/dev/null 2>&1
