#!/usr/bin/env python3
"""GPU busy / idle inside the training step from a rocprofv3 --kernel-trace CSV: per step = from one `normalize_kernel` launch (K0,
the first kernel of a bench step) to the next.  usage: step_timeline.py kernel_trace.csv [steps to skip]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "normalize_kernel" in r["Kernel_Name"]]
steps = []
for a, b in zip(starts, starts[1:]):
    ks = rows[a:b]
    if not any("adamw" in r["Kernel_Name"] for r in ks):
        continue  # not a training step
    t0, t1 = int(ks[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks)
    gaps = sorted(((int(n["Start_Timestamp"]) - int(p["End_Timestamp"])), p["Kernel_Name"][:60], n["Kernel_Name"][:60]) for p, n in zip(ks, ks[1:] + [rows[b]]))
    steps.append((t1 - t0, busy, len(ks), gaps))
steps = steps[skip:]
if not steps:
    sys.exit("no training steps found")
n = len(steps)
wall = sum(s[0] for s in steps) / n / 1e6
busy = sum(s[1] for s in steps) / n / 1e6
print(f"{n} steps: wall {wall:.3f} ms, kernels {busy:.3f} ms ({100 * busy / wall:.1f} %), idle {wall - busy:.3f} ms, {steps[0][2]} launches/step, mean gap {1e3 * (wall - busy) / steps[0][2]:.2f} us")
g = steps[len(steps) // 2][3]
print("largest gaps of one step (us):")
for d, p, q in g[-8:]:
    print(f"  {d / 1e3:8.2f}  after {p}  before {q}")
