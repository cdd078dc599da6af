#!/bin/bash
# HBM-traffic PMC passes for bench.py (separate --pmc runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes).
# Usage: tools/pmc_bench.sh OUTDIR [bench.py args...]   -> OUTDIR/pmc_FETCH_SIZE/, OUTDIR/pmc_WRITE_SIZE/, then
#        python3 tools/pmc_summarize.py OUTDIR > profiles/<name>.json
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_$P -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-profile --no-cpu-baseline --no-parity-leg --no-tile --no-yaml-legs "$@" > $OUT/pmc_$P.log 2>&1
done
