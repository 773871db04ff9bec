#!/bin/bash
# Vector-memory front end of the trace kernel (TA = address unit, TCP = the CU's L1): is the kernel waiting for the
# rate at which the L1 takes per-lane addresses?  Separate rocprofv3 --pmc passes over tools/prof_run.py.
#   tools/prof_ta.sh <tag> [prof_run.py args]
set -u
TAG=$1; shift
OUT=gpurun_out/ta_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
ARGS="$*"
run() { name=$1; shift; (cd /tmp && timeout -k 5 200 rocprofv3 "$@" --output-format csv -d $ROOT/$OUT/$name -- python3 $ROOT/tools/prof_run.py $ARGS > $ROOT/$OUT/$name.log 2>&1; echo "pass $name rc=$?"); }
run ta1 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
run ta2 --kernel-trace --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
run ta3 --kernel-trace --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
run ta4 --kernel-trace --pmc TA_BUFFER_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum
run tcp1 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp2 --kernel-trace --pmc TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tcp3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
run tcp4 --kernel-trace --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run tcp5 --kernel-trace --pmc TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum
run sq1 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM
grep -h "per render" $OUT/*.log | head -3
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if "queuekernel" not in n and "packetkernel" not in n and "k_gather" not in n: continue
            k = (n[:40], r.get("Counter_Name"))
            agg[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
        for k in sorted(agg): print("   %-40s %-40s per_launch=%.6g (n=%d)" % (k[0], k[1], agg[k] / cnt[k], cnt[k]))
PY
