#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_builds.sh OTHER.so ROUNDS -- command...   (IG_HIP_LIB selects the build)
OTHER=$1; ROUNDS=$2; shift 3
for r in $(seq $ROUNDS); do
  echo "== round $r: in-tree build"; "$@" 2>&1 | grep -v amdgpu.ids
  echo "== round $r: $OTHER"; IG_HIP_LIB=$OTHER "$@" 2>&1 | grep -v amdgpu.ids
done
