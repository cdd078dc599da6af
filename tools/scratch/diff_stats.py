import csv, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
sa = a[[k for k in a if 'adamw_kernel' in k][0]][0]; sb = b[[k for k in b if 'adamw_kernel' in k][0]][0]
rows = []
for k in set(a) | set(b):
    ta = a.get(k, (0, 0))[1] / sa / 1e3; tb = b.get(k, (0, 0))[1] / sb / 1e3
    rows.append((tb - ta, k, ta, tb, b.get(k, (0, 0))[0] / sb))
rows.sort(reverse=True)
for d, k, ta, tb, n in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    print(f"{d:+8.1f} us/step  {ta:8.1f} -> {tb:8.1f}  x{n:5.1f}  {k[:90]}")
print("total", sum(r[2] for r in rows), sum(r[3] for r in rows))
