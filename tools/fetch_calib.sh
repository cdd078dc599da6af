#!/bin/bash
# tools/fetch_calib.sh OUTDIR : FETCH_SIZE / WRITE_SIZE per known byte count (separate --pmc passes, --kernel-trace only)
OUT=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib.hip -o /tmp/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_$P -o pmc -- /tmp/fetch_calib > $OUT/pmc_$P.log 2>&1
done
python3 - <<PY
import csv,glob,collections
BYTES=160*1024*6144
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$OUT/pmc_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"known bytes per kernel: {BYTES} ({BYTES/2**20:.0f} MiB); counters are in KiB")
for k,v in sorted(acc.items()):
    f=v.get("FETCH_SIZE",[0])[-1]*1024; w=v.get("WRITE_SIZE",[0])[-1]*1024
    print(f"{k:24s} FETCH_SIZE {f/2**20:9.1f} MiB = {f/BYTES:5.3f} x bytes (correction factor {BYTES/f if f else 0:5.3f})   WRITE_SIZE {w/2**20:9.1f} MiB = {w/BYTES:5.3f} x bytes")
PY
