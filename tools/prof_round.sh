#!/bin/bash
# A round's evidence run (gpurun):  tools/prof_round.sh <tag>   [STEP=micro|bench|configs|scaling|census|fuzz]
# VALU and gather ceilings, rocprofv3 passes of the bench command, the un-profiled bench line with those passes replayed, BASELINE configs 2 / 4 / 5,
# scaling emulation with one and two frames in flight, lane census, fuzz campaigns.  tools/install_profiles.sh <tag> copies the results into profiles/.
set -u
export TMPDIR=/tmp
TAG=${1:?round tag, e.g. r06}
O=gpurun_out/${TAG}c
mkdir -p $O
STEP=${STEP:-all}
if [ $STEP = all ] || [ $STEP = micro ]; then
  timeout 600 ./tools/micro/valu_issue 20000 > $O/valu_ceiling.txt 2>&1; echo "valu rc=$?"; tail -2 $O/valu_ceiling.txt
  timeout 600 ./tools/micro/gather > $O/gather_ceiling.txt 2>&1; echo "gather rc=$?"
fi
if [ $STEP = all ] || [ $STEP = bench ]; then
  bash tools/prof_bench.sh $TAG > $O/prof_bench.log 2>&1; tail -5 $O/prof_bench.log
  cp gpurun_out/prof_bench_$TAG/traffic.json profiles/traffic.json          # (on the box) so that the line below replays this device code's counters
  timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench_line.json
fi
if [ $STEP = all ] || [ $STEP = configs ]; then
  bash tools/prof_configs.sh $TAG > $O/configs.log 2>&1; tail -12 $O/configs.log | cut -c1-300
fi
if [ $STEP = all ] || [ $STEP = scaling ]; then
  { echo "# tools/gpu_scaling_emulation.py on one MI355X (compute only: the ranks' shares rendered one after the other, no RCCL), round $TAG"
    echo "# -- tile split, one frame at a time"; SPLIT=tile timeout 900 python3 tools/gpu_scaling_emulation.py
    echo "# -- sample split, one frame at a time"; SPLIT=sample timeout 900 python3 tools/gpu_scaling_emulation.py
    echo "# -- tile split, two frames in flight (PIPE=1: wall ms per frame over 12 frames, two contexts; bench.py times this mode as well from 8 ranks on)"; PIPE=1 SPLIT=tile timeout 900 python3 tools/gpu_scaling_emulation.py
    echo "# -- sample split, two frames in flight"; PIPE=1 SPLIT=sample timeout 900 python3 tools/gpu_scaling_emulation.py
  } > $O/scaling_emulation.txt 2>&1; cat $O/scaling_emulation.txt
fi
if [ $STEP = all ] || [ $STEP = census ]; then
  MOPTIX_DEBUG=1 SPP=64 timeout 300 python3 tools/gpu_quick.py > $O/census_coffee.txt 2>&1; tail -3 $O/census_coffee.txt
  MOPTIX_DEBUG=1 SCENE=million_standin IARG=1000000 SPP=16 timeout 300 python3 tools/gpu_census.py > $O/census_c5.txt 2>&1; tail -2 $O/census_c5.txt
fi
if [ $STEP = all ] || [ $STEP = fuzz ]; then
  for seed in 101 102 103; do CASES=300 SEED=$seed timeout 1500 python3 tools/gpu_fuzz.py > $O/fuzz_$seed.log 2>&1; tail -1 $O/fuzz_$seed.log; done
  CASES=300 SEED=55 timeout 1500 python3 tools/gpu_oracle_fuzz.py > $O/ofuzz_55.log 2>&1; tail -1 $O/ofuzz_55.log
fi
