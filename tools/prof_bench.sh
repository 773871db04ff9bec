#!/bin/bash
# rocprofv3 evidence for the bench.py command (run on the GPU box via gpurun):
#   kernel-trace --stats of `python3 bench.py` and separate PMC passes for HBM traffic.
set -u
TAG=${1:?round tag, e.g. r06}
OUT=gpurun_out/prof_bench_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run() { name=$1; shift; args=$1; shift; (cd /tmp && timeout -k 5 240 rocprofv3 "$@" --output-format csv -d $ROOT/$OUT/$name -- python3 $ROOT/bench.py $args > $ROOT/$OUT/$name.log 2>&1; echo "pass $name rc=$?"); }
run kt "--steps 3 --warmup 1 --no-cpu-baseline" --kernel-trace --stats
run fetch "--steps 1 --warmup 0 --no-cpu-baseline --no-fast-leg" --kernel-trace --pmc FETCH_SIZE
run write "--steps 1 --warmup 0 --no-cpu-baseline --no-fast-leg" --kernel-trace --pmc WRITE_SIZE
run l2 "--steps 1 --warmup 0 --no-cpu-baseline --no-fast-leg" --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run ta "--steps 1 --warmup 0 --no-cpu-baseline --no-fast-leg" --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
run sq "--steps 1 --warmup 0 --no-cpu-baseline --no-fast-leg" --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
for f in $OUT/*.log; do echo "== $f"; grep -h '"metric"' $f | cut -c1-400; done
for f in $(find $OUT/kt -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = (r.get("Kernel_Name", "")[:48], r.get("Counter_Name"))
            if "queuekernel" not in k[0] and "packetkernel" not in k[0] and "drainkernel" not in k[0] and "reduce" not in k[0]: continue
            agg[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
        for k in sorted(agg): print("   %-48s %-24s sum=%.6g launches=%d per_launch=%.6g" % (k[0], k[1], agg[k], cnt[k], agg[k] / cnt[k]))
PY
python3 tools/make_traffic_json.py $OUT $OUT/traffic.json
