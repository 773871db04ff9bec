#!/bin/bash
# Copies the evidence of the last tools/prof_round.sh <tag> run (gpurun_out/, merged back by gpurun) into profiles/.  Run here, after the call.
set -eu
TAG=${1:?round tag, e.g. r06}
O=gpurun_out/${TAG}c
f=$(ls -t gpurun_out/prof_bench_$TAG/kt/runc/*_kernel_stats.csv | head -1)
cp "$f" profiles/${TAG}_bench_coffee256_kernel_stats.csv
cp $O/valu_ceiling.txt profiles/${TAG}_valu_ceiling.txt
cp $O/gather_ceiling.txt profiles/${TAG}_gather_ceiling.txt
cp $O/bench_line.json profiles/${TAG}_bench_line.json
python3 tools/prof_summary.py gpurun_out/prof_bench_$TAG > profiles/${TAG}_bench_coffee256_rocprofv3_summary.txt
cp gpurun_out/prof_bench_$TAG/traffic.json profiles/traffic.json
P=gpurun_out/prof_configs_$TAG
cp $P/configs.json profiles/${TAG}_configs.json
for c in c2 c4 c5; do cp $P/${TAG}_traffic_$c.json profiles/${TAG}_traffic_$c.json; done
python3 tools/prof_configs_summary.py $P > profiles/${TAG}_configs_pmc.txt
cp $O/scaling_emulation.txt profiles/${TAG}_scaling_emulation.txt
{ echo "# round $TAG: lane census and phase clocks of the counting build (MOPTIX_DEBUG=1): coffee 1920x1080 at 64 spp (tools/gpu_quick.py), then the glass knot at 16 spp (tools/gpu_census.py)"
  grep -v "depth history" $O/census_coffee.txt; echo "== million_standin 1920x1080 16 spp"; grep -v "depth history" $O/census_c5.txt; } > profiles/${TAG}_lane_census.txt
python3 tools/fuzz_summary.py $O/fuzz_*.log > profiles/${TAG}_fuzz.txt
python3 tools/oracle_fuzz_summary.py $O/ofuzz_*.log > profiles/${TAG}_oracle_fuzz.txt
[ -f gpurun_out/${TAG}d/ab.log ] && cp gpurun_out/${TAG}d/ab.log profiles/${TAG}_experiments_ab.txt
python3 tools/node_step_isa.py > profiles/${TAG}_node_step_isa.txt
{ echo "# registers, spills and LDS of the shipped trace kernels (tools/kernel_resources.py on minimaloptix_amd/lib/libmoptix.so; pinned by tests/test_capi_symbols.py)"
  echo "# round 3: timed instantiation 11 vector / 111 scalar spills; round 4 and 5: 2 / 57; round 6: 6 / 53"
  python3 tools/kernel_resources.py minimaloptix_amd/lib/libmoptix.so packetkernel; python3 tools/kernel_resources.py minimaloptix_amd/lib/libmoptix.so queuekernel; python3 tools/kernel_resources.py minimaloptix_amd/lib/libmoptix.so drainkernel; python3 tools/kernel_resources.py minimaloptix_amd/lib/libmoptix.so megakernel; } > profiles/${TAG}_kernel_resources.txt
python3 - <<PY
import json, bench
t = json.load(open("profiles/traffic.json"))
assert t["source_hash"] == bench.source_hash("."), ("traffic.json is for other sources", t["source_hash"], bench.source_hash("."))
print("traffic.json matches the sources:", t["source_hash"], "| VALU instructions per launch %.4g, fabric %.1f GB" % (t["SQ"]["SQ_INSTS_VALU"], t["traffic_GB_per_launch"]))
for c in ("c2", "c4", "c5"):
    t = json.load(open("profiles/${TAG}_traffic_%s.json" % c)); assert t["source_hash"] == bench.source_hash("."), c
PY
head -3 profiles/${TAG}_bench_coffee256_kernel_stats.csv | cut -c1-150
