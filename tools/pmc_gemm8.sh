#!/bin/bash
# SQ / LDS counter passes for a GEMM microbench: tools/pmc_gemm8.sh OUTDIR [bench script + args]
# default bench: tools/gemm8_bench.py --v8-only; e.g. tools/pmc_gemm8.sh gpurun_out/pmc_w tools/wgrad_bench.py --group-only
# (separate --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes; at most 5 SQ counters per pass, and every
# pass is checked and bounded by `timeout 300`: a rejected counter set must not produce a silently empty table or a hung box).  Prints per-kernel averages with
#   mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs x GRBM_GUI_ACTIVE / 8 XCDs)      (cycles / cycles)
#   wait fractions = SQ_WAIT_* / SQ_WAVE_CYCLES                                           (quad-cycles / quad-cycles)
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p $OUT
if [ $# -eq 0 ]; then set -- tools/gemm8_bench.py --v8-only; fi
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
pass() {  # name counters...
    local name=$1; shift
    timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $SCRIPT "${ARGS[@]}" > $OUT/pmc_$name.log 2>&1
    local rc=$?
    local n=$(find $OUT/pmc_$name -name '*counter_collection.csv' -size +0 | wc -l)
    if [ $rc -ne 0 ] || [ "$n" -eq 0 ]; then echo "PMC pass $name FAILED (rc=$rc, csv files=$n): see $OUT/pmc_$name.log"; tail -5 $OUT/pmc_$name.log; fi
}
ARGS=("$@")
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass lds1 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
pass lds2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16
pass fetch FETCH_SIZE   # FETCH_SIZE and WRITE_SIZE in SEPARATE passes: together rocprofv3 aborts (signal 6) and hangs on this pool
pass write WRITE_SIZE
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for fn in glob.glob("$OUT/pmc_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","").replace(" ","")
        if not any(t in k for t in ("gemm", "wgrad", "cls_", "classifier", "bn_", "attn")): continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k,v in agg.items():
    print(k)
    a={c: v[c]/max(n[k][c],1) for c in v}
    for c in sorted(a): print(f"   {c:32s} {a[c]:16.0f}")
    if a.get("GRBM_GUI_ACTIVE"):
        cyc=a["GRBM_GUI_ACTIVE"]/8
        print(f"   -> kernel cycles (GRBM_GUI_ACTIVE/8) {cyc:.0f}; mfma_util = {a.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(4*256*cyc):.3f}")
    wc=a.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VALU"):
            if c in a: print(f"   -> {c}/SQ_WAVE_CYCLES = {a[c]/wc:.3f}")
    if a.get("SQ_LDS_IDX_ACTIVE"): print(f"   -> LDS bank-conflict cycles / LDS active cycles = {a.get('SQ_LDS_BANK_CONFLICT',0)/a['SQ_LDS_IDX_ACTIVE']:.3f}")
    if "FETCH_SIZE" in a: print(f"   -> HBM/L2-miss traffic per launch: (2 x FETCH_SIZE + WRITE_SIZE) KiB = {(2*a['FETCH_SIZE']+a.get('WRITE_SIZE',0))/1024:.1f} MiB")
PY
