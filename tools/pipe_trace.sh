#!/bin/bash
# timeline of two frames in flight on one GPU (whole frame): start / end of every trace and reduce kernel under rocprofv3 --kernel-trace
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/r05f; mkdir -p $O
export MOPTIX_BENCH_PIPELINE=1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/pipe -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast-leg > $O/pipe.log 2>&1)
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/pipe/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "pt_packetkernel<false" in n or "k_reduce" in n or "vectorized_elementwise" in n or "fill" in n.lower():
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:60], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
t0 = rows[0][0]
for s, e, n, q in rows[-40:]:
    print("%10.3f -> %10.3f ms  (%8.3f)  q %s  %s" % ((s - t0) * 1e-6, (e - t0) * 1e-6, (e - s) * 1e-6, q, n))
PY
