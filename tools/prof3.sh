#!/bin/bash
# PMC passes, each under its own timeout (a bad counter set can wedge rocprofv3)
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
ARGS="$*"
run() { name=$1; shift; (cd /tmp && timeout -k 5 75 rocprofv3 "$@" --output-format csv -d $ROOT/$OUT/$name -- python3 $ROOT/tools/prof_run.py $ARGS > $ROOT/$OUT/$name.log 2>&1; echo "pass $name rc=$?"); }
run a --kernel-trace --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE
run b --kernel-trace --pmc TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum
run c --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run d --kernel-trace --pmc TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run e --kernel-trace --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run f --kernel-trace --pmc TD_TD_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
run g --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
grep -h "per render" $OUT/*.log | head -2
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "megakernel" not in r.get("Kernel_Name", "") and "poolkernel" not in r.get("Kernel_Name", ""): continue
            k = r.get("Counter_Name")
            agg[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
        for k in sorted(agg): print("   %-40s sum=%.6g n=%d" % (k, agg[k], cnt[k]))
PY
