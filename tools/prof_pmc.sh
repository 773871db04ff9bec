#!/bin/bash
# PMC characterisation of the trace kernel (separate rocprofv3 passes, kernel-trace only): $1 tag, rest = prof_run.py args
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
ARGS="$*"
run() { name=$1; shift; (cd /tmp && timeout -k 5 200 rocprofv3 "$@" --output-format csv -d $ROOT/$OUT/$name -- python3 $ROOT/tools/prof_run.py $ARGS > $ROOT/$OUT/$name.log 2>&1); }
run kt --kernel-trace --stats
run pmc1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM
run pmc2 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
run pmc3 --kernel-trace --pmc FETCH_SIZE
run pmc4 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run pmc5 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run pmc6 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM
grep -h "per render" $OUT/*.log | head -3
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; grep -E "Name|kernel|reduce" $f | cut -c1-200 | head -6; done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if "queuekernel" not in n and "packetkernel" not in n: continue
            k = (n[n.find("pt_"):][:40], r.get("Counter_Name"))
            agg[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
        for k in sorted(agg): print("   %-40s %-28s per_launch=%.6g (n=%d)" % (k[0], k[1], agg[k] / cnt[k], cnt[k]))
PY
