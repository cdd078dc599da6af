#!/bin/bash
# tools/prof_mode.sh BATCH TAG : kernel stats of a short training bench in the mode the environment selects
R=${GRAFT_REPO_ROOT:-/root/repo}; B=$1; TAG=$2; OUT=$R/gpurun_out/det; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o b -- python3 $R/bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-parity-leg --no-tile --no-profile --detail-file /tmp/d.json > $OUT/prof_$TAG.log 2>&1)
cp $OUT/prof_$TAG/b_kernel_stats.csv $OUT/kernel_stats_$TAG.csv; rm -rf $OUT/prof_$TAG
