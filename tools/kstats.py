"""Print rocprofv3 kernel_stats.csv compactly: python tools/kstats.py FILE [N]"""
import csv, re, sys
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= limit: break
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])
    n = re.sub(r"\(.*", "", n)
    print(f"{n[:70]:70s} x{int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
