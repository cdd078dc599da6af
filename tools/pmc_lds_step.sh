#!/bin/bash
# LDS bank-conflict share of every kernel of a short training bench: tools/pmc_lds_step.sh OUTDIR [bench args]
# (one --pmc pass with --kernel-trace only; prints kernels by LDS-active cycles with conflict cycles / active cycles)
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/lds -o pmc -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity-leg --no-tile --no-yaml-legs --no-profile --detail-file /tmp/d.json "$@" > $OUT/lds.log 2>&1
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in glob.glob("$OUT/lds/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="SQ_LDS_IDX_ACTIVE": n[k]+=1
rows=sorted(agg.items(), key=lambda kv: -kv[1]["SQ_LDS_IDX_ACTIVE"])
print(f"{'kernel':70s} {'launches':>8s} {'LDS active Mcyc':>16s} {'conflict share':>15s}")
for k,v in rows[:40]:
    a=v["SQ_LDS_IDX_ACTIVE"]; c=v["SQ_LDS_BANK_CONFLICT"]
    if a<=0: continue
    print(f"{k[:70]:70s} {n[k]:8d} {a/1e6:16.1f} {c/a:15.3f}")
PY
