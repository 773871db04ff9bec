#!/bin/bash
# Instruction fetch of the trace kernel: requests and their accumulated outstanding level (average latency = LEVEL / IFETCH).
# The kernel's code is 55-57 KB against a 64 KB instruction cache shared by two CUs.   tools/prof_ifetch.sh <tag> [prof_run.py args]
set -u
TAG=$1; shift
OUT=gpurun_out/ifetch_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
ARGS="$*"
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $ROOT/$OUT/p -- python3 $ROOT/tools/prof_run.py $ARGS > $ROOT/$OUT/p.log 2>&1)
grep -h "per render" $OUT/p.log
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "packetkernel" not in n and "queuekernel" not in n: continue
        k = (n[n.find("pt_"):][:44], r.get("Counter_Name"))
        agg[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
    for k in sorted(agg): print("   %-44s %-20s per_launch=%.6g (n=%d)" % (k[0], k[1], agg[k] / cnt[k], cnt[k]))
PY
