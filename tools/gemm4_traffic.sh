#!/bin/bash
# L2-miss traffic of gemm4_kernel<0,0,false> per GEMM shape (separate --pmc passes, FETCH_SIZE doubled as the guide prescribes).
# usage: tools/gemm4_traffic.sh OUTDIR [M]
OUT=$1; M=${2:-85104}; R=${GRAFT_REPO_ROOT:-/root/repo}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_$P -o pmc -- python3 $R/tools/gemm4_traffic.py $M > $OUT/pmc_$P.log 2>&1
done
python3 - <<PY
import csv, glob
M, D = $M, 768
shapes = [("qkv fwd", 3 * D, D), ("d_qkv", D, 3 * D), ("d_proj", D, D), ("d_fc1", D, 4 * D)]
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % ctr, recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == ctr and "gemm4_kernel<0, 0, false" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals[ctr] = [float(r["Counter_Value"]) for r in rows]
n = len(vals["FETCH_SIZE"]) // len(shapes)
print(f"M = {M}: {n} launches per shape; the first of each group excluded (cold weights)")
for i, (name, N, K) in enumerate(shapes):
    f = vals["FETCH_SIZE"][i * n + 1:(i + 1) * n]; w = vals["WRITE_SIZE"][i * n + 1:(i + 1) * n]
    fetch = 2 * sum(f) / len(f) * 1024 / 1e6; write = sum(w) / len(w) * 1024 / 1e6
    alg_r, alg_w = (M * K + N * K) * 2 / 1e6, M * N * 2 / 1e6
    print(f"{name:8s} N={N:5d} K={K:5d}: fetch {fetch:7.1f} MB (algorithmic {alg_r:6.1f}, x 8 XCD weight copies {alg_r + 7 * N * K * 2 / 1e6:6.1f})  write {write:6.1f} MB ({alg_w:6.1f})  total {fetch + write:7.1f} / {alg_r + alg_w:6.1f} = {(fetch + write) / (alg_r + alg_w):.2f} x")
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
