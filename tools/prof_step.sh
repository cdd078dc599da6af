#!/bin/bash
# tools/prof_step.sh BATCH TAG [bench args]: rocprofv3 kernel trace of a short training bench -> per-step busy / idle split and the
# per-kernel time of ONE training step (inference legs excluded) under gpurun_out/step_TAG/
R=${GRAFT_REPO_ROOT:-/root/repo}; B=$1; TAG=$2; shift 2; OUT=$R/gpurun_out/step_$TAG; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o b -- python3 $R/bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-parity-leg --no-tile --no-yaml-legs --no-profile --detail-file /tmp/d.json "$@" > $OUT/prof.log 2>&1)
python3 $R/tools/step_timeline.py $OUT/prof/b_kernel_trace.csv 4 > $OUT/timeline.txt 2>&1; head -1 $OUT/timeline.txt
python3 - <<PY > $OUT/train_step_kernels.txt
import csv, collections
rows = list(csv.DictReader(open("$OUT/prof/b_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "normalize_kernel" in r["Kernel_Name"]]
steps = [rows[a:b] for a, b in zip(starts, starts[1:]) if any("adamw_kernel" in r["Kernel_Name"] for r in rows[a:b])][4:]
acc = collections.defaultdict(lambda: [0, 0.0])
for ks in steps:
    for r in ks:
        k = acc[r["Kernel_Name"][:110]]
        k[0] += 1; k[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
n = len(steps)
tot = sum(v[1] for v in acc.values()) / n
print(f"{n} training steps, {sum(v[0] for v in acc.values()) / n:.0f} launches, {tot:.1f} us of kernels per step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1] / n:9.1f} us  x{v[0] / n:5.1f}  avg {v[1] / v[0]:8.1f}  {k}")
PY
head -40 $OUT/train_step_kernels.txt; rm -rf $OUT/prof
