#!/bin/bash
# Everything the round's evidence files come from, on ONE box: tools/round_profiles.sh TAG   (outputs under gpurun_out/TAG/)
#   1. default bench.py (all legs)                       -> bench.log, bench_detail.json
#   2. rocprofv3 --kernel-trace --stats of a short bench -> kernel_stats.csv, kernel_trace.csv (tools/step_timeline.py)
#   3. HBM-traffic PMC passes of the same command        -> pmc_bench.json
#   4. the other BASELINE configs / batch sizes          -> configs/*.json
# every profiler invocation is bounded by `timeout`.
TAG=${1:-r04}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 600 python bench.py --detail-file $OUT/bench_detail.json > $OUT/bench.log 2>&1; tail -c 2200 $OUT/bench.log
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-leg --no-tile --no-yaml-legs --no-profile --detail-file /tmp/d.json > $OUT/prof.log 2>&1)
cp $OUT/prof/b_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; python3 tools/step_timeline.py $OUT/prof/b_kernel_trace.csv 4 > $OUT/step_timeline.txt 2>&1; cat $OUT/step_timeline.txt | head -3
timeout 700 tools/pmc_bench.sh $OUT/pmc > $OUT/pmc.log 2>&1; python3 tools/pmc_summarize.py $OUT/pmc > $OUT/pmc_bench.json 2>$OUT/pmc_sum.log; wc -c $OUT/pmc_bench.json
timeout 600 tools/bench_configs.sh $OUT/configs 2>&1 | grep -v amdgpu
rm -rf $OUT/prof/b_kernel_trace.csv $OUT/pmc/pmc_*/  # large raw files stay on the box
