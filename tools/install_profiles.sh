#!/bin/bash
# Copies the evidence of the last tools/r04_prof.sh run (gpurun_out/) into profiles/ (run here, after gpurun has merged its output).
set -eu
TAG=${1:-r04}
f=$(ls -t gpurun_out/prof_bench_$TAG/kt/runc/*_kernel_stats.csv | head -1)
cp "$f" profiles/${TAG}_bench_coffee256_kernel_stats.csv
cp gpurun_out/r04c/valu_ceiling.txt profiles/${TAG}_valu_ceiling.txt
python3 tools/prof_summary.py gpurun_out/prof_bench_$TAG > profiles/${TAG}_bench_coffee256_rocprofv3_summary.txt
cp gpurun_out/prof_bench_$TAG/traffic.json profiles/traffic.json
python3 - <<PY
import json, bench
t = json.load(open("profiles/traffic.json"))
assert t["source_hash"] == bench.source_hash("."), ("traffic.json is for other sources", t["source_hash"], bench.source_hash("."))
print("traffic.json matches the sources:", t["source_hash"], "| VALU instructions per launch %.4g, fabric %.1f GB" % (t["SQ"]["SQ_INSTS_VALU"], t["traffic_GB_per_launch"]))
PY
head -3 profiles/${TAG}_bench_coffee256_kernel_stats.csv | cut -c1-150
