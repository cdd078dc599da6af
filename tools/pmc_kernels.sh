#!/bin/bash
# SQ stall-breakdown PMC pass for a micro-benchmark script: tools/pmc_kernels.sh OUTDIR script.py [args]
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $R/"$@" > $OUT/pmc_sq.log 2>&1
python3 - <<PY
import csv,glob,collections,os
f=glob.glob("$OUT/pmc_sq/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:150]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        if r["Counter_Name"]=="SQ_WAVE_CYCLES": n[k]+=1
for k,v in agg.items():
    if os.environ.get("PMC_FILTER", "direct") not in k and os.environ.get("PMC_FILTER", "direct") != "*": continue
    wc=v["SQ_WAVE_CYCLES"]
    print(k, "launches",n[k])
    # units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles (4 shader cycles), so their ratio
    # to SQ_WAVE_CYCLES is a fraction of wave time; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count shader CYCLES: MFMA-pipe
    # utilisation = MFMA busy cycles / SQ_BUSY_CYCLES (both in cycles), and against wave time it is divided by 4 x quad-cycles
    cyc = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES")
    for c in v:
        if c in cyc:
            extra = f"{v[c]/max(v['SQ_BUSY_CYCLES'],1):6.3f} of SQ busy cycles, {v[c]/(4*wc):6.3f} of wave cycles (cycles / 4 x quad-cycles)"
        else:
            extra = f"{v[c]/wc:6.3f} of wave quad-cycles"
        print(f"   {c:28s} {v[c]/max(n[k],1):14.0f}  {extra}")
PY
