"""Per-step kernel-time difference of two rocprofv3 kernel_stats.csv files (normalised by their AdamW launch counts):
python tools/diff_kernel_stats.py A.csv B.csv [rows]"""
import csv
import sys


def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}


a, b = load(sys.argv[1]), load(sys.argv[2])
sa = a[[k for k in a if "adamw_kernel" in k][0]][0]
sb = b[[k for k in b if "adamw_kernel" in k][0]][0]
rows = []
for k in set(a) | set(b):
    ta, tb = a.get(k, (0, 0))[1] / sa / 1e3, b.get(k, (0, 0))[1] / sb / 1e3
    rows.append((tb - ta, k, ta, tb, b.get(k, (0, 0))[0] / sb))
rows.sort(reverse=True)
for d, k, ta, tb, n in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    print(f"{d:+8.1f} us/step  {ta:8.1f} -> {tb:8.1f}  x{n:5.1f}  {k[:90]}")
print("total us/step", round(sum(r[2] for r in rows), 1), "->", round(sum(r[3] for r in rows), 1))
