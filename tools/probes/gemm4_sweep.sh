#!/bin/bash
# Builds and times variants of the generated 4-wave GEMM K-loop on the GPU box.  usage: gemm4_sweep.sh "cfg1" "cfg2" ...   (cfg = "k=v k=v"; a
# leading "@a" runs the timing with alias mode a: see gemm4_probe.hip)
cd "$(dirname "$0")"
GEN=../../instageo-e2e-geospatial-ml_amd/csrc/gen_gemm4.py
i=0
for cfg in "$@"; do
  i=$((i+1))
  alias=0
  if [[ "$cfg" == @* ]]; then alias=${cfg:1:1}; cfg=${cfg:2}; fi
  d=/tmp/g4_$i; mkdir -p $d; cp gemm4_probe.hip $d/
  python3 $GEN $d/gemm4_gen.inc $cfg || exit 1
  (cd $d && hipcc -O3 --offload-arch=gfx950 gemm4_probe.hip -o gemm4_probe) || exit 1
  echo "=== variant $i: alias=$alias $cfg"
  if [ $i -eq 1 ]; then
    timeout 120 $d/gemm4_probe 1000 512 256 3
    timeout 120 $d/gemm4_probe 777 768 768 3
  fi
  timeout 120 $d/gemm4_probe 42552 2304 768 60 $alias
  timeout 120 $d/gemm4_probe 42552 768 3072 60 $alias
done
